"""Mean-interpolate gradient (gather over the transposed inter-level graph) at the decoder's last level, direct kernel timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import tf_nnquery, tf_unpool3d, tf_sample, sph3gcn_util as u
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0')
B, K = 16, 64
xyz = torch.from_numpy(synth.s3dis_batch(1000, B, 8192)[0]).to(dev)[:, :, :3].contiguous()
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
si = tf_sample.farthest_point_sample(2048, xyz)
pairs = torch.stack([torch.arange(B, device=dev).view(B, 1).expand(B, 2048).int(), si], dim=-1).contiguous()
xyz_c = u.gather_nd(xyz, pairs)
iidx, icnt, idst = tf_nnquery.build_sphere_neighbor(xyz_c, xyz, 0.1, None, K)
print("edges", int(icnt.sum()))
for C in (128, 64, 256):
    xc = torch.randn(B, 2048, C, device=dev); gz = torch.randn(B, 8192, C, device=dev)
    t = timeit(lambda: tf_unpool3d.mean_interpolate_grad(xc, gz, iidx, icnt))
    tf_ = timeit(lambda: tf_unpool3d.mean_interpolate(xc, iidx, icnt))
    print("C=%d: mean_interpolate fwd %.3f ms  grad %.3f ms" % (C, tf_, t), flush=True)
