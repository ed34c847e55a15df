#!/usr/bin/env python3
"""Condense two rocprofv3 counter passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each on its own, no tracing domains) of
`bench.py` into profiles/r01_pmc_traffic.json: mean bytes per launch and kernel.

    python tools/pmc_summary.py <fetch_dir> <write_dir> [out.json]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM): Counter_Value is KiB; on gfx950 FETCH_SIZE
reads 1/2 for WIDE COALESCED 16-B/lane streams and other access widths must be calibrated on a known byte count in the
code's own access pattern.  Calibration used here: ntt_pass_kernel moves exactly 2 x 32 B per element (one read, one write of
the whole vector per pass, 32 B per lane as 2 x dwordx4) -- raw FETCH + WRITE is compared with that known figure and the
resulting factor is recorded; both the raw and the guide-doubled fetch figures are kept."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]) * 1024.0)
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r01_pmc_traffic.json")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [])))
        w = sum(write.get(k, [0])) / max(1, len(write.get(k, [])))
        kernels[k] = {"launches": len(fetch.get(k, [])), "fetch_bytes_raw": f, "write_bytes": w, "raw_total": f + w, "guide_doubled_fetch_total": 2 * f + w}
    ntt = [v for k, v in kernels.items() if "ntt_pass_kernel" in k]
    n = 1 << 20
    calib = None
    if ntt:
        calib = {"known_bytes_per_launch": 2 * 32 * n, "raw_total_mean": sum(v["raw_total"] for v in ntt) / len(ntt),
                 "raw_over_known": sum(v["raw_total"] for v in ntt) / len(ntt) / (2 * 32 * n),
                 "note": "2^20-point passes of the bench's NTT leg; a ratio ~1.0 means the raw counters are exact for this 32-B-per-lane pattern"}
    acc = kernels.get("bbgpu::msm_accumulate_kernel", {})
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing domains) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-plonk",
           "units": "Counter_Value is KiB; bytes = value * 1024; means per launch",
           "calibration": calib,
           "msm_accumulate_kernel_bytes_per_launch": acc.get("raw_total"),
           "msm_accumulate_kernel_bytes_per_launch_guide_doubled_fetch": acc.get("guide_doubled_fetch_total"),
           "kernels": kernels}
    json.dump(out, open(out_path, "w"), indent=1)
    print("wrote", out_path, "accumulate raw", acc.get("raw_total"))


if __name__ == "__main__":
    main()
