"""one library (BBGPU_LIB or the in-tree build), resident transforms at 2^18 / 2^20 / 2^22: run it alternately with another build for a one-box A/B"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from barretenberg_amd import BbGpu
G = BbGpu(0)
s = torch.cuda.Stream()
for lg in (18, 20, 22):
    n = 1 << lg
    x = np.random.default_rng(1).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    d = torch.from_numpy(x.view(np.int64)).cuda()
    for kind in ("fft", "coset_fft", "ifft"):
        for _ in range(3): G.ntt_device(d.data_ptr(), n, kind, stream=s.cuda_stream)
        torch.cuda.synchronize()
        for _ in range(max(20, int(60.0 / (0.1 * (n >> 20 or 1))) if n >= (1 << 20) else 1500)): G.ntt_device(d.data_ptr(), n, kind, stream=s.cuda_stream)  # ~60 ms: out of the post-idle ramp
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): G.ntt_device(d.data_ptr(), n, kind, stream=s.cuda_stream)
        e1.record(s); torch.cuda.synchronize()
        print("lib=%s 2^%d %-10s %.4f ms" % (os.path.basename(os.environ.get("BBGPU_LIB", "libbbgpu.so")), lg, kind, e0.elapsed_time(e1) / 20), flush=True)
G.shutdown()
