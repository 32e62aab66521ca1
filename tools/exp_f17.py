import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, tf_conv3d
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0'); _lib.lib()
B, K, N = 16, 64, 8192
xyz = torch.from_numpy(synth.s3dis_batch(1000, B, N)[0]).to(dev)[:, :, :3].contiguous()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20000000); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
nidx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, 0.1, None, K)
filt = tf_buildkernel.spherical_kernel(xyz, xyz, nidx, cnt, dst, 0.1, [8, 2, 2])
print("bins used", torch.unique(filt).tolist())
filt17 = torch.where(filt == 0, filt, filt - 16).contiguous()
for C in (64, 128):
    x = torch.randn(B, N, C, device=dev)
    w33 = torch.randn(33, C, 2, device=dev); w17 = torch.cat([w33[:1], w33[17:]]).contiguous()
    t33 = timeit(lambda: tf_conv3d.depthwise_conv3d(x, w33, nidx, cnt, filt))
    t17 = timeit(lambda: tf_conv3d.depthwise_conv3d(x, w17, nidx, cnt, filt17))
    err = (tf_conv3d.depthwise_conv3d(x, w33, nidx, cnt, filt) - tf_conv3d.depthwise_conv3d(x, w17, nidx, cnt, filt17)).abs().max().item()
    print("C=%d fwd F=33 %.3f ms   F=17 %.3f ms  diff %.1e" % (C, t33, t17, err))
