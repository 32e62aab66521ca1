"""gather forward at the level shapes of the S3DIS plan (isolated timings) — for A/B of library builds via SPH3D_LIB"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import tf_conv3d, tf_nnquery, tf_sample
from sph3d_gcn_amd.harness import synth
dev = torch.device("cuda:0")
B = 16
xyz = torch.from_numpy(synth.s3dis_batch(1000, B, 8192)[0]).to(dev)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
out = []
levels = [(8192, 0.1, (64, 128)), (2048, 0.2, (128, 256, 512)), (768, 0.4, (256, 512)), (384, 0.8, (256, 512, 1024))]
cur = xyz
for n, rad, cs in levels:
    while cur.shape[1] > n:
        nxt = {8192: 2048, 2048: 768, 768: 384}[cur.shape[1]]
        idx = tf_sample.farthest_point_sample(nxt, cur)
        cur = torch.gather(cur, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    nidx, cnt, dst, filt = tf_nnquery.build_sphere_graph(cur, rad, 64, [8, 2, 2], with_transpose=False)
    for C in cs:
        x = torch.randn(B, n, C, device=dev); w = torch.randn(33, C, 2, device=dev)
        out.append("%dx%d %.1f" % (n, C, timeit(lambda: tf_conv3d.depthwise_conv3d(x, w, nidx, cnt, filt))))
print(os.environ.get("SPH3D_LIB", "default")[-14:], " ".join(out))
