"""FPS chain times at the S3DIS level sizes (B=16)."""
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
tot = 0.0
for m in (2048, 768, 384, 128):
    t = timeit(lambda: tf_sample.farthest_point_sample(m, xyz))
    idx = tf_sample.farthest_point_sample(m, xyz)
    print("n=%5d -> %4d: %.3f ms  (%.3f us/round)" % (xyz.shape[1], m, t, t * 1e3 / m)); tot += t
    xyz = torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
print("chain %.3f ms" % tot)
