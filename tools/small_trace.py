import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from barretenberg_amd import BbGpu
from oracle.pyoracle import Oracle, aligned_copy
O = Oracle(); G = BbGpu(0); G.set_host_thresholds(0, 0)
srs = O.make_srs(O.random_scalars(7, 1)[0], 1024); table = O.point_table(srs); sc = O.random_scalars(9, 1024)
for n in (256, 1000):
    s, t = aligned_copy(sc[:n]), aligned_copy(table[:2 * n])
    for _ in range(6): G.pippenger(s, t, n)
G.shutdown()
