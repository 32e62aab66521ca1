"""Weight-gradient (TN, split-K) product at the step's shapes; variants selected by SPH3D_TN_* environment variables."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import tf_gemm
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(131072,128,128),(131072,256,128),(32768,256,256),(32768,512,256),(12288,512,256),(6144,512,512),(6144,1024,512),(2048,1024,512),(6144,2048,256),(12288,1024,256),(32768,1024,128)]
tot = 0; out = []
for R, Ci, Co in shapes:
    x = torch.randn(R, Ci, device=dev); dy = torch.randn(R, Co, device=dev)
    t = timeit(lambda: tf_gemm._pointwise_gemm_tn(x, dy)); tot += t
    ref = x.t() @ dy
    err = ((tf_gemm._pointwise_gemm_tn(x, dy) - ref).abs().max() / ref.abs().max()).item()
    out.append("%.3f(%.0f)" % (t, 2 * R * Ci * Co / 1e9 / t))
    assert err < 1e-4, err
print(os.environ.get("SPH3D_TN_BK16", "-"), os.environ.get("SPH3D_TN_WANT", "-"), "total %.3f ms :" % tot, " ".join(out))
