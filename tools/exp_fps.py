import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_sample
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0'); _lib.lib()
xyz = torch.from_numpy(synth.s3dis_batch(1000, 16, 8192)[0]).to(dev)[:, :, :3].contiguous()
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for n, m in ((8192, 2048), (2048, 768), (768, 384), (384, 128)):
    x = xyz[:, :n].contiguous()
    print("fps %d -> %d: %.3f ms (%.2f us per round)" % (n, m, timeit(lambda: tf_sample.farthest_point_sample(m, x)), timeit(lambda: tf_sample.farthest_point_sample(m, x)) * 1e3 / m))
