"""The fused level-0 graph kernel (256 workgroups of 1024 threads, 112 KB of LDS: one per CU) alone and while the FPS
chain (16 workgroups of 1024 threads, resident for 2.2 ms) runs on another stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import tf_nnquery, tf_sample, _tgraph
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0')
xyz = torch.from_numpy(synth.s3dis_batch(1000, 16, 8192)[0]).to(dev)[:, :, :3].contiguous()
side = torch.cuda.Stream()
def go(): tf_nnquery.build_sphere_graph(xyz, 0.1, 64, [8, 2, 2], with_transpose=False)
def timed(n, with_fps):
    go(); torch.cuda.synchronize()
    if with_fps:
        with torch.cuda.stream(side):
            for _ in range(3): tf_sample.farthest_point_sample(2048, xyz)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("fused graph kernel alone %.3f ms, beside FPS %.3f ms" % (timed(5, False), timed(5, True)))
f0 = torch.cuda.Event(enable_timing=True); f1 = torch.cuda.Event(enable_timing=True)
f0.record(); tf_sample.farthest_point_sample(2048, xyz); f1.record(); torch.cuda.synchronize()
a = f0.elapsed_time(f1)
with torch.cuda.stream(side):
    for _ in range(6): go()
f0.record(); tf_sample.farthest_point_sample(2048, xyz); f1.record(); torch.cuda.synchronize()
print("FPS 8192->2048 alone %.3f ms, beside graph kernels %.3f ms" % (a, f0.elapsed_time(f1)))
