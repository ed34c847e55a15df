#!/usr/bin/env python3
"""Per-pass durations of the resident transforms by kind, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ntt_pt -- python3 tools/ntt_pass_times.py run 22
    python3 tools/ntt_pass_times.py report gpurun_out/ntt_pt 22
`run` issues, for every kind in ORDER, 3 + 10 transforms of 2^lg elements on one stream; `report` groups the ntt_pass_fused dispatches of the trace
in that order (two per transform) and prints the mean duration of each pass over the last 10 transforms of each kind."""
import glob, os, sys
ORDER = ["fft", "coset_fft", "ifft", "fft", "coset_ifft", "ifft"]
WARM, REPS = 3, 10


def run(lg):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np, torch
    from barretenberg_amd import BbGpu
    G = BbGpu(0)
    n = 1 << lg
    x = np.random.default_rng(1).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    d = torch.from_numpy(x.view(np.int64)).cuda()
    s = torch.cuda.Stream()
    for kind in ("fft", "ifft", "coset_fft", "coset_ifft"):  # tables of every kind built before the first measured dispatch
        G.ntt_device(d.data_ptr(), n, kind, stream=s.cuda_stream)
    torch.cuda.synchronize()
    for kind in ORDER:
        for _ in range(WARM + REPS):
            G.ntt_device(d.data_ptr(), n, kind, stream=s.cuda_stream)
        torch.cuda.synchronize()
    G.shutdown()


def report(path, lg):
    import csv
    rows = []
    for f in glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "ntt_pass_fused" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[8:]  # the four table-building transforms
    per = 2 * (WARM + REPS)
    assert len(rows) == per * len(ORDER), (len(rows), per * len(ORDER))
    print("2^%d, mean over the last %d transforms of each run: pass 1 us, pass 2 us, gap between them us, start-to-end us" % (lg, REPS))
    for k, kind in enumerate(ORDER):
        g = rows[k * per + 2 * WARM:(k + 1) * per]
        p1 = [e - s for s, e, _ in g[0::2]]
        p2 = [e - s for s, e, _ in g[1::2]]
        gap = [g[i + 1][0] - g[i][1] for i in range(0, len(g), 2)]
        tot = [g[i + 1][1] - g[i][0] for i in range(0, len(g), 2)]
        name = lambda s: s.split("ntt_pass_fused_kernel")[1].split("(")[0]
        print("%-11s %s %7.1f  %s %7.1f  gap %5.1f  total %7.1f" % (kind, name(g[0][2]), sum(p1) / len(p1) / 1e3, name(g[1][2]), sum(p2) / len(p2) / 1e3,
                                                                 sum(gap) / len(gap) / 1e3, sum(tot) / len(tot) / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]))
    else:
        report(sys.argv[2], int(sys.argv[3]))
