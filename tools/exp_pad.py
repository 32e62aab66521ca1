import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as Fn
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, tf_conv3d
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0'); _lib.lib()
B, K, N = 32, 64, 10000
xyz = torch.from_numpy(synth.modelnet_batch(0, B, N)).to(dev).contiguous()
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20000000); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
nidx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, 0.1, None, K)
filt = tf_buildkernel.spherical_kernel(xyz, xyz, nidx, cnt, dst, 0.1, [8, 2, 2])
print("nnz", int(cnt.sum()))
for C, r in ((35, 2), (67, 1)):
    C4 = (C + 3) // 4 * 4
    x = torch.randn(B, N, C, device=dev); w = torch.randn(33, C, r, device=dev); go = torch.randn(B, N, C * r, device=dev)
    xp = Fn.pad(x, (0, C4 - C)).contiguous(); wp = Fn.pad(w, (0, 0, 0, C4 - C)).contiguous(); gop = Fn.pad(go, (0, (C4 - C) * r)).contiguous()
    tf = timeit(lambda: tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt)); tfp = timeit(lambda: tf_conv3d.depthwise_conv3d(xp, wp, nidx, cnt, filt))
    tb = timeit(lambda: tf_conv3d.depthwise_conv3d_grad(x, w, go, nidx, cnt, filt)); tbp = timeit(lambda: tf_conv3d.depthwise_conv3d_grad(xp, wp, gop, nidx, cnt, filt))
    o = tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt); op = tf_conv3d.depthwise_conv3d(xp, wp, nidx, cnt, filt)[:, :, :C * r]
    print("C=%d r=%d  fwd generic %.3f ms padded-vec %.3f ms | bwd generic %.3f padded-vec %.3f | max diff %.1e" % (C, r, tf, tfp, tb, tbp, (o - op).abs().max().item()))
