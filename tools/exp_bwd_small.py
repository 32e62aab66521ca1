"""One small-level conv gradient shape (default 16 x 768, C = 256, r = 2, radius 0.4) in a loop: for a kernel trace / counters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, tf_conv3d, tf_sample
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0'); _lib.lib()
N = int(os.environ.get("N", "768")); C = int(os.environ.get("C", "256")); rad = float(os.environ.get("RAD", "0.4"))
xyz = torch.from_numpy(synth.s3dis_batch(1000, 16, 8192)[0]).to(dev)[:, :, :3].contiguous()
for m in (2048, 768, 384):
    if m < N: break
    idx = tf_sample.farthest_point_sample(m, xyz)
    xyz = torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    if m == N: break
nidx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, rad, None, 64)
filt = tf_buildkernel.spherical_kernel(xyz, xyz, nidx, cnt, dst, rad, [8, 2, 2])
x = torch.randn(16, N, C, device=dev); w = torch.randn(33, C, 2, device=dev); go = torch.randn(16, N, C * 2, device=dev)
for _ in range(20):
    tf_conv3d.depthwise_conv3d_grad(x, w, go, nidx, cnt, filt)
    tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt)
torch.cuda.synchronize()
print("edges", int(cnt.sum()))
