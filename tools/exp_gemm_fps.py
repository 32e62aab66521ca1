"""Do the 16 workgroups of the FPS chain (one per cloud, 1024 threads, resident for 2 ms) disturb a GEMM that is sized
for exactly four workgroups on every CU?  The level-0 products timed alone and with an FPS launch running on another stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import tf_gemm, tf_sample
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0')
xyz = torch.from_numpy(synth.s3dis_batch(1000, 16, 8192)[0]).to(dev)[:, :, :3].contiguous()
side = torch.cuda.Stream()
def timed(fn, n, with_fps):
    fn(); torch.cuda.synchronize()
    if with_fps:
        with torch.cuda.stream(side):
            for _ in range(3): tf_sample.farthest_point_sample(2048, xyz)      # ~6.6 ms of FPS
        e = torch.cuda.Event(); e.record(side)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for R, Ci, Co in ((131072, 256, 128), (131072, 128, 128), (32768, 512, 256), (32768, 1024, 128)):
    x = torch.randn(R, Ci, device=dev); w = torch.randn(Ci, Co, device=dev); dy = torch.randn(R, Co, device=dev)
    for name, fn in (("NN", lambda: tf_gemm._pointwise_gemm(x, w, False)), ("NT", lambda: tf_gemm._pointwise_gemm(dy, w, True)),
                     ("TN", lambda: tf_gemm._pointwise_gemm_tn(x, dy))):
        a = timed(fn, 30, False); b = timed(fn, 30, True)
        print("R%6d %4d->%4d %s alone %.3f ms   with FPS running %.3f ms  (x%.2f)" % (R, Ci, Co, name, a, b, b / a), flush=True)
