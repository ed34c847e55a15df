import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from barretenberg_amd import BbGpu
G = BbGpu(0); n = 1 << 20
x = np.random.default_rng(1).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
d = torch.from_numpy(x.view(np.int64)).cuda()
s = torch.cuda.Stream()
for kind in ("fft", "coset_fft", "ifft"):
    for _ in range(3): G.ntt_device(d.data_ptr(), n, kind, stream=s.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(20): G.ntt_device(d.data_ptr(), n, kind, stream=s.cuda_stream)
    e1.record(s); torch.cuda.synchronize()
    print("lib=%s %-10s %.4f ms" % (os.path.basename(os.environ.get("BBGPU_LIB", "libbbgpu.so")), kind, e0.elapsed_time(e1) / 20))
