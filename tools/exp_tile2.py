"""tile2 experiment: plan + tiled forward vs the gather kernel at the north-star shape (correctness + timing)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sph3d_gcn_amd import _lib, _plan, tf_conv3d, tf_nnquery
from sph3d_gcn_amd.harness import synth

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 16))
xyz = torch.from_numpy(synth.s3dis_batch(1000, B, 8192)[0]).to(dev)
idx, cnt, dst, filt = tf_nnquery.build_sphere_graph(xyz, 0.1, 64, [8, 2, 2], with_transpose=False)
F = 33


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for ucap in [int(u) for u in os.environ.get("UCAPS", "144").split(",")]:
    for use_order in (True, False):
        _plan.clear()
        _plan.register_geometry(filt, xyz, xyz)
        t_plan = timeit(lambda: (_plan._fwd2.clear(), _plan.forward_plan2(idx, cnt, filt, F, ucap, use_order)), 10)
        chdr, rec, ulist, _ = _plan.forward_plan2(idx, cnt, filt, F, ucap, use_order)
        nt = chdr.view(-1, 136)[:, 0].float()
        a = chdr.view(-1, 136)[:, 1:129:2]
        U = (a >> 16).float()
        T = ((a >> 8) & 0xff).float()
        msk = T > 0
        print("ucap %d order %d: plan %.1f us  tiles/chunk %.2f  T mean %.2f  U mean %.1f  reuse %.2f" % (
            ucap, use_order, t_plan, nt.mean().item(), T[msk].mean().item(), U[msk].mean().item(),
            cnt.sum().item() / U[msk].sum().item()))
        for C in (128, 64, 256):
            x = torch.randn(B, 8192, C, device=dev)
            w = torch.randn(F, C, 2, device=dev)
            _plan.set_mode("gather")
            ref = tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt)
            t_g = timeit(lambda: tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt))
            out = torch.empty_like(ref)
            l = _lib.lib()
            fn = lambda: _lib.check(l.sph3d_depthwise_conv3d_tiled2(B, 8192, 8192, F, C, 2, ucap, _lib.ptr(chdr), _lib.ptr(rec),
                                                                    _lib.ptr(ulist), _lib.ptr(x), _lib.ptr(w), _lib.ptr(out),
                                                                    _lib.stream_ptr()))
            fn()
            torch.cuda.synchronize()
            err = (out - ref).abs().max().item()
            scale = ref.abs().max().item()
            t_t = timeit(fn)
            print("   C=%d: gather %.1f us  tiled2 %.1f us   max|diff| %.3g (scale %.3g)" % (C, t_g, t_t, err, scale))
t_o = timeit(lambda: (_plan._orders.clear(), _plan.spatial_order(xyz)), 10)
print("spatial_order %.1f us" % t_o)
