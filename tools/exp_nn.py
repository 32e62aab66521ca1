import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery, tf_sample
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0'); _lib.lib()
B, K = 16, 64
xyz = torch.from_numpy(synth.s3dis_batch(1000, B, 8192)[0]).to(dev)[:, :, :3].contiguous()
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20000000); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
idx = tf_sample.farthest_point_sample(2048, xyz)
xyz1 = torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
print("intra 8192x8192 r=.1  %.3f ms" % timeit(lambda: tf_nnquery.build_sphere_neighbor(xyz, xyz, 0.1, None, K)))
print("inter db2048 q8192 r=.1 %.3f ms" % timeit(lambda: tf_nnquery.build_sphere_neighbor(xyz1, xyz, 0.1, None, K)))
print("intra 2048 r=.2  %.3f ms" % timeit(lambda: tf_nnquery.build_sphere_neighbor(xyz1, xyz1, 0.2, None, K)))
print("intra 2048 r=.1  %.3f ms" % timeit(lambda: tf_nnquery.build_sphere_neighbor(xyz1, xyz1, 0.1, None, K)))
print("fused graph 8192 r=.1 no transpose %.3f ms" % timeit(lambda: tf_nnquery.build_sphere_graph(xyz, 0.1, K, [8, 2, 2], with_transpose=False)))
print("fused graph 8192 r=.1 + transpose  %.3f ms" % timeit(lambda: tf_nnquery.build_sphere_graph(xyz, 0.1, K, [8, 2, 2], with_transpose=True)))
# how long are the scans?  hits per query class (j // 1024) and the index at which slot K fills
i_, c_, d_ = tf_nnquery.build_sphere_neighbor(xyz, xyz, 0.1, None, K)
last = torch.gather(i_, 2, (c_.long() - 1).clamp(min=0).unsqueeze(-1)).squeeze(-1).float()
for cls in range(8):
    sl = slice(cls * 1024, (cls + 1) * 1024)
    full = (c_[:, sl] == K).float().mean().item()
    scan = torch.where(c_[:, sl] == K, last[:, sl], torch.full_like(last[:, sl], 8192.0)).mean().item()
    print("class %d (r = %.2f): %.0f %% of the queries fill K; mean scan length %.0f points" % (cls, 0.1 + 0.05 * cls, 100 * full, scan))
