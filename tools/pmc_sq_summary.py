#!/usr/bin/env python3
"""Per-kernel means of an SQ counter pass (rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-plonk).
usage: pmc_sq_summary.py <dir> <out.json>.  Wave time splits into three disjoint parts (MI355X_MICROARCH.md, counter table):
ACTIVE_INST_ANY (issuing), WAIT_INST_ANY (ready but the pipe is taken: issue-bound), WAIT_ANY (parked on s_waitcnt / barrier: latency-bound)."""
import collections, csv, glob, json, os, sys

src, out = sys.argv[1], sys.argv[2]
path = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"].split("(")[0].replace("bbgpu::", "").replace("void ", "")
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in sorted(agg.items()):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    res[k] = {"launches": len(next(iter(v.values()))), "mean_per_launch": {c: round(x) for c, x in m.items()},
              "wave_time_fraction": {"issuing": round(m["SQ_ACTIVE_INST_ANY"] / wc, 3), "issue_stalled": round(m["SQ_WAIT_INST_ANY"] / wc, 3),
                                     "parked_on_waitcnt_or_barrier": round(m["SQ_WAIT_ANY"] / wc, 3)},
              "valu_share_of_issued": round(m["SQ_ACTIVE_INST_VALU"] / max(m["SQ_ACTIVE_INST_ANY"], 1.0), 3)}
json.dump({"command": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE "
                      "--output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-plonk", "kernels": res}, open(out, "w"), indent=1)
for k in res:
    if any(w in k for w in ("msm_accumulate", "ntt_pass", "sortB", "sortA_scatter", "msm_rowcol", "msm_merge_kernel", "msm_digits")):
        print(k, res[k]["wave_time_fraction"], "VALU wave-instructions/launch %.3g" % res[k]["mean_per_launch"].get("SQ_INSTS_VALU", 0))
