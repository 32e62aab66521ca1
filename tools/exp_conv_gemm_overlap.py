"""Can the level-0 depthwise gather and the pointwise product overlap when they run as two KERNELS on two streams (different
register allocations, co-resident on a CU) — what the one-kernel layer could not do (csrc/sepring.hip)?  Independent operands:
conv on stream A, GEMM (with the statistics epilogue) on stream B, alone / one after the other / side by side."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph3d_gcn_amd import _lib, tf_nnquery, tf_conv3d, tf_norm
from sph3d_gcn_amd.harness import synth
dev = torch.device('cuda:0'); _lib.lib()
B, n = 16, 8192
xyz = torch.from_numpy(synth.s3dis_batch(1000, B, n)[0]).to(dev)
nidx, cnt, dst, filt = tf_nnquery.build_sphere_graph(xyz, 0.1, 64, [8, 2, 2], with_transpose=False)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
with torch.no_grad():
    for C, Cout in ((128, 128), (64, 128)):
        x = torch.randn(B, n, C, device=dev); dw = torch.randn(33, C, 2, device=dev)
        d0 = torch.randn(B * n, C * 2, device=dev); w = torch.randn(C * 2, Cout, device=dev) / 16
        conv = lambda: tf_conv3d._depthwise_conv3d_impl(x, dw, nidx, cnt, filt)
        gemm = lambda: tf_norm._gemm_bnstats_impl(d0, w, None)
        def both():
            cur = torch.cuda.current_stream()
            sa.wait_stream(cur); sb.wait_stream(cur)
            with torch.cuda.stream(sa): conv()
            with torch.cuda.stream(sb): gemm()
            cur.wait_stream(sa); cur.wait_stream(sb)
        def chunked(parts=4):
            # the layer as it would run: conv of cloud group i on A, then its product on B beside the conv of group i + 1
            cur = torch.cuda.current_stream()
            sa.wait_stream(cur); sb.wait_stream(cur)
            step = B // parts
            for i in range(parts):
                with torch.cuda.stream(sa):
                    d = tf_conv3d._depthwise_conv3d_impl(x[i * step:(i + 1) * step], dw, nidx[i * step:(i + 1) * step], cnt[i * step:(i + 1) * step], filt[i * step:(i + 1) * step])
                    ev = torch.cuda.Event(); ev.record(sa)
                sb.wait_event(ev)
                with torch.cuda.stream(sb):
                    tf_norm._gemm_bnstats_impl(d.view(-1, C * 2), w, None)
                d.record_stream(sb)
            cur.wait_stream(sa); cur.wait_stream(sb)
        tc, tg = timeit(conv), timeit(gemm)
        ts = timeit(lambda: (conv(), gemm()))
        tb = timeit(both)
        t4, t2 = timeit(lambda: chunked(4)), timeit(lambda: chunked(2))
        print("C=%3d -> %3d: conv %6.1f us, gemm(stats) %6.1f us, one after the other %6.1f, side by side on two streams %6.1f, "
              "layer in 4 / 2 cloud groups pipelined over two streams %6.1f / %6.1f" % (C, Cout, tc, tg, ts, tb, t4, t2))
