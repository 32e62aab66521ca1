"""Co-operative FPS (clouds above 24 576 points, csrc/sample.hip: fps_coop_kernel): time per round at the ScanNet-shape sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sph3d_gcn_amd import tf_sample, _lib
dev = torch.device("cuda:0"); _lib.lib()
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
rng = np.random.RandomState(5)
for B, N, m in ((1, 65536, 16384), (1, 65536, 2048), (2, 40000, 2048), (1, 131072, 2048), (1, 262144, 1024), (8, 65536, 1024), (3, 40000, 1000)):
    x = torch.from_numpy((rng.rand(B, N, 3) * np.array([6.0, 6.0, 3.0])).astype(np.float32)).to(dev)
    t = timeit(lambda: tf_sample.farthest_point_sample(m, x))
    print("%s B%2d N%7d -> %5d : %9.1f us  %.3f us/round" % (os.environ.get("TAG", ""), B, N, m, t * 1e3, t * 1e3 / (m - 1)), flush=True)
