import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from sph3d_gcn_amd import tf_sample
dev = torch.device("cuda:0")
for B, n, m in [(3, 40000, 1000), (3, 32768, 1000), (2, 40000, 1000), (4, 40000, 1000), (1, 40000, 1000), (8, 65536, 1000), (3, 49152, 1000)]:
    x = torch.rand(B, n, 3, device=dev)
    tf_sample.farthest_point_sample(m, x); torch.cuda.synchronize()
    t0 = time.perf_counter(); tf_sample.farthest_point_sample(m, x); torch.cuda.synchronize()
    print(B, n, m, "%.2f ms  %.2f us/round" % ((time.perf_counter() - t0) * 1e3, (time.perf_counter() - t0) * 1e6 / m))
