"""Would one launch for a layer's two backward products pay?  Input gradient (NT) and weight gradient (TN + slab sum) of the step's shapes:
one after the other on one stream vs side by side on two streams (what a horizontally fused launch could reach at best)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_gemm
dev = torch.device('cuda:0'); _lib.lib()
shapes = [(131072, 128, 128), (131072, 256, 128), (32768, 256, 256), (32768, 512, 256), (12288, 512, 256), (6144, 512, 512), (6144, 1024, 512),
          (2048, 1024, 512), (6144, 2048, 256), (12288, 1024, 256), (32768, 1024, 128)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ts = tp = 0
for R, Ci, Co in shapes:
    x = torch.randn(R, Ci, device=dev); w = torch.randn(Ci, Co, device=dev); dy = torch.randn(R, Co, device=dev)
    def serial():
        tf_gemm._pointwise_gemm_impl(dy, w, True); tf_gemm._pointwise_gemm_tn_impl(x, dy)
    def par():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1): tf_gemm._pointwise_gemm_impl(dy, w, True)
        with torch.cuda.stream(s2): tf_gemm._pointwise_gemm_tn_impl(x, dy)
        cur.wait_stream(s1); cur.wait_stream(s2)
    a = timeit(lambda: tf_gemm._pointwise_gemm_impl(dy, w, True)); b = timeit(lambda: tf_gemm._pointwise_gemm_tn_impl(x, dy))
    c = timeit(serial); d = timeit(par)
    ts += c; tp += d
    print("R%6d %4d->%4d  NT %5.1f  TN %5.1f  serial %5.1f  two streams %5.1f us" % (R, Ci, Co, a, b, c, d))
print("sum serial %.0f us, two streams %.0f us" % (ts, tp))
