"""What bounds the conv gradient: the level-0 kernels timed with the gathered grad_out rows folded onto fewer and fewer
distinct rows per cloud (keys modulo R: same instruction stream, same edge lists, shrinking memory footprint)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, tf_conv3d, _tgraph
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0')
B, K, N, rad = 16, 64, 8192, 0.1
xyz = torch.from_numpy(synth.s3dis_batch(1000, B, N)[0]).to(dev)[:, :, :3].contiguous()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
nidx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, rad, None, K)
filt = tf_buildkernel.spherical_kernel(xyz, xyz, nidx, cnt, dst, rad, [8, 2, 2])
for C in (128, 64):
    x = torch.randn(B, N, C, device=dev); w = torch.randn(33, C, 2, device=dev); go = torch.randn(B, N, C * 2, device=dev)
    tf_conv3d.depthwise_conv3d_grad(x, w, go, nidx, cnt, filt)
    offsets, ent_key, ent_scale, active = _tgraph.transpose(nidx, cnt, N, filt, None, 33)
    orig = ent_key.clone()
    for R in (8192, 2048, 512, 64, 1):
        ent_key.copy_(orig % R)
        t = timeit(lambda: tf_conv3d.depthwise_conv3d_grad(x, w, go, nidx, cnt, filt))
        print("C=%d distinct grad_out rows per cloud %5d: %.3f ms" % (C, R, t), flush=True)
    ent_key.copy_(orig)
