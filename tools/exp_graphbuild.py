"""Level-0 graph construction, piece by piece (B=16 x 8192, K=64, kernel [8,2,2]): separate kernels against the fused one."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, _tgraph
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0')
B, K = 16, 64
xyz0 = torch.from_numpy(synth.s3dis_batch(1000, B, 8192)[0]).to(dev)[:, :, :3].contiguous()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
from sph3d_gcn_amd import tf_sample
for N, rad in ((8192, 0.1), (2048, 0.2)):
    xyz = xyz0 if N == 8192 else torch.gather(xyz0, 1, tf_sample.farthest_point_sample(N, xyz0).long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    a = timeit(lambda: tf_nnquery.build_sphere_neighbor(xyz, xyz, rad, None, K))
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, rad, None, K)
    b = timeit(lambda: tf_buildkernel.spherical_kernel(xyz, xyz, idx, cnt, dst, rad, [8, 2, 2]))
    filt = tf_buildkernel.spherical_kernel(xyz, xyz, idx, cnt, dst, rad, [8, 2, 2])
    def tr():
        _tgraph._cache.clear()
        _tgraph.transpose(idx, cnt, N, bin_index=filt, num_bins=33)
    c = timeit(tr)
    d = timeit(lambda: tf_nnquery.build_sphere_graph(xyz, rad, K, [8, 2, 2], with_transpose=False))
    def fused():
        _tgraph._cache.clear()
        tf_nnquery.build_sphere_graph(xyz, rad, K, [8, 2, 2], with_transpose=True)
    e = timeit(fused)
    print("N=%d: search %.3f + bins %.3f + transpose %.3f = %.3f ms | fused search+bins %.3f | fused all (incl. finish) %.3f ms" % (N, a, b, c, a + b + c, d, e), flush=True)
