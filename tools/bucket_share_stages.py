import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from barretenberg_amd import BbGpu
G = BbGpu(0)
n = 1 << 20
rng = np.random.default_rng(7)
x = rng.integers(0, 1 << 64, size=4, dtype=np.uint64); x[3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
srs = G.srs_generate(x, n)
sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64(0x1FFFFFFFFFFFFFFF)
d = torch.from_numpy(sc.view(np.int64)).cuda()
for _ in range(20): G.msm_device(srs, d.data_ptr(), n)
G.set_timing(True)
for N in (1, 2, 4, 8):
    for s in sorted({0, N // 2, N - 1}):
        acc = np.zeros(7)
        for _ in range(5):
            G.msm_wait(G.msm_device_buckets_async(srs, d.data_ptr(), n, s, N)); acc += np.array(G.last_timing()[:7])
        print("N=%d share %d: total %.3f digits %.3f sort %.3f acc %.3f merge %.3f folds %.3f collect %.3f" % ((N, s) + tuple(acc / 5)), flush=True)
