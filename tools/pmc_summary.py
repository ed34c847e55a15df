#!/usr/bin/env python3
"""Condense rocprofv3 counter passes into profiles/r02_pmc_traffic.json: HBM bytes per launch and kernel, corrected as
/opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes -- separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes (no tracing
domains), Counter_Value in KiB, and the gfx950 under-count of FETCH_SIZE calibrated ON KNOWN BYTE COUNTS IN THE KERNELS' OWN ACCESS SHAPES:
tools/ubench/ubench_traffic runs under the same two passes and gives one factor per shape (known bytes / raw counter):

    calib_gather64  one random 64-byte row per lane (4 x dwordx4) from a 2 GiB table  -> msm_accumulate_kernel, srs_table_kernel
    calib_stream32  32 contiguous bytes per lane in and out                           -> ntt_pass_kernel
    calib_stream16  16 contiguous bytes per lane in and out (the guide's x2 case)     -> everything else (digits, sort, tail)

    python tools/pmc_summary.py <bench_fetch_dir> <bench_write_dir> <calib_fetch_dir> <calib_write_dir> [out.json [<ntt22_fetch_dir> <ntt22_write_dir>]]

bench passes:  rocprofv3 --pmc FETCH_SIZE -d <dir> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-plonk --no-boundary
calib passes:  rocprofv3 --pmc FETCH_SIZE -d <dir> -- tools/ubench/ubench_traffic        (and the same with WRITE_SIZE)"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KNOWN = {  # bytes per launch of tools/ubench/ubench_traffic.hip
    "calib_gather64": {"read": (1 << 24) * 64 + (1 << 24) * 4, "write": 0},
    "calib_stream32": {"read": (1 << 24) * 32, "write": (1 << 24) * 32},
    "calib_stream16": {"read": (1 << 24) * 16, "write": (1 << 24) * 16},
}


def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]) * 1024.0)
    return acc


def lib_sha16():
    """which libbbgpu.so the counter passes ran (bench.py prints the same digest of the library IT loaded: roofline.traffic_source)"""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        return hashlib.sha256(open(os.environ.get("BBGPU_LIB") or os.path.join(root, "barretenberg_amd", "libbbgpu.so"), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def mean(v):
    return sum(v) / len(v) if v else 0.0


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    cfetch, cwrite = load(sys.argv[3], "FETCH_SIZE"), load(sys.argv[4], "WRITE_SIZE")
    out_path = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_pmc_traffic.json")
    calib = {}
    for k, known in KNOWN.items():
        # the third launch of each (the first two also pay cold misses of the memset-initialised buffers)
        rf = cfetch.get(k, [0])[-1]
        rw = cwrite.get(k, [0])[-1]
        calib[k] = {"known_read_bytes": known["read"], "known_write_bytes": known["write"], "raw_fetch_bytes": rf, "raw_write_bytes": rw,
                    "fetch_factor": known["read"] / rf if rf else None, "write_factor": (known["write"] / rw) if (rw and known["write"]) else None}

    def shape_of(kernel):
        if "msm_accumulate" in kernel or "srs_table" in kernel:
            return "calib_gather64"
        if "ntt_pass" in kernel:
            return "calib_stream32"
        return "calib_stream16"

    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = mean(fetch.get(k, [])), mean(write.get(k, []))
        sh = shape_of(k)
        ff = calib[sh]["fetch_factor"] or 1.0
        wf = calib[sh]["write_factor"] or 1.0
        kernels[k] = {"launches": len(fetch.get(k, [])), "fetch_bytes_raw": f, "write_bytes_raw": w, "calibration_shape": sh,
                      "fetch_bytes": f * ff, "write_bytes": w * wf, "total_bytes": f * ff + w * wf}
    acc = kernels.get("bbgpu::msm_accumulate_kernel", {})
    ntt = [v for k, v in kernels.items() if "ntt_pass" in k]
    # bench.py --no-boundary launches 2^20-point transforms only: one transform = one pass-1 launch + one pass-2 launch
    ntt_total = sum(v["total_bytes"] * v["launches"] for v in ntt) / max(1, sum(v["launches"] for v in ntt)) * 2 if ntt else None
    # optional: the same two passes around tools/ntt_only.py 22 (every ntt_pass launch of that process is a 2^22-point pass)
    ntt22_total = None
    if len(sys.argv) > 7:
        f22, w22 = load(sys.argv[6], "FETCH_SIZE"), load(sys.argv[7], "WRITE_SIZE")
        ff, wf = calib["calib_stream32"]["fetch_factor"] or 1.0, calib["calib_stream32"]["write_factor"] or 1.0
        per = [mean(f22.get(k, [])) * ff + mean(w22.get(k, [])) * wf for k in sorted(set(f22) | set(w22)) if "ntt_pass" in k]
        ntt22_total = sum(per) if len(per) == 2 else None  # one pass-1 kernel + one pass-2 kernel
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing domains) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-plonk --no-boundary; "
                     "calibration: the same two passes -- tools/ubench/ubench_traffic",
           "library_sha256_16": lib_sha16(),
           "units": "Counter_Value is KiB; bytes = value * 1024; means per launch; *_raw as counted, the others multiplied by the factor of the kernel's access shape",
           "calibration": calib,
           "msm_accumulate_kernel_bytes_per_launch": acc.get("total_bytes"),
           "msm_accumulate_kernel_bytes_per_launch_raw": (acc.get("fetch_bytes_raw", 0) + acc.get("write_bytes_raw", 0)) if acc else None,
           "msm_algorithmic_bytes": (1 << 20) * 96 + 96,
           "ntt_2e20_bytes_per_transform": ntt_total,
           "ntt_2e20_algorithmic_bytes": 2 * 32 * (1 << 20),
           "ntt_2e22_bytes_per_transform": ntt22_total,
           "ntt_2e22_algorithmic_bytes": 2 * 32 * (1 << 22),
           "kernels": kernels}
    json.dump(out, open(out_path, "w"), indent=1)
    print("wrote", out_path)
    for k, v in calib.items():
        print("  %s fetch factor %s write factor %s" % (k, v["fetch_factor"], v["write_factor"]))
    print("  accumulate: raw %.4g corrected %.4g bytes/launch; ntt 2^20: %s bytes/transform" % (out["msm_accumulate_kernel_bytes_per_launch_raw"] or 0, acc.get("total_bytes", 0), ntt_total))


if __name__ == "__main__":
    main()
