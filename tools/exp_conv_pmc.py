"""Conv launches at S3DIS level 0 for PMC passes: MODE=fwd|bwd, C=128."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, tf_conv3d
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0'); _lib.lib()
B, K, N = 16, 64, 8192
C = int(os.environ.get("C", "128")); MODE = os.environ.get("MODE", "fwd")
xyz = torch.from_numpy(synth.s3dis_batch(1000, B, N)[0]).to(dev)[:, :, :3].contiguous()
nidx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, 0.1, None, K)
filt = tf_buildkernel.spherical_kernel(xyz, xyz, nidx, cnt, dst, 0.1, [8, 2, 2])
x = torch.randn(B, N, C, device=dev); w = torch.randn(33, C, 2, device=dev); go = torch.randn(B, N, C * 2, device=dev)
for _ in range(3):
    if MODE == "fwd":
        tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt)
    else:
        tf_conv3d.depthwise_conv3d_grad(x, w, go, nidx, cnt, filt)
torch.cuda.synchronize()
