#!/usr/bin/env python3
"""nothing but resident ffts of 2^lg elements (for counter passes: every ntt_pass launch of the process belongs to that size)
    rocprofv3 --pmc FETCH_SIZE -d <dir> -- python3 tools/ntt_only.py 22 [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from barretenberg_amd import BbGpu
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
G = BbGpu(0)
n = 1 << lg
d = torch.from_numpy(np.random.default_rng(1).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64).view(np.int64)).cuda()
for _ in range(reps):
    G.ntt_device(d.data_ptr(), n, "fft")
torch.cuda.synchronize()
G.shutdown()
