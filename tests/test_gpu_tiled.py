"""LDS-tiled depthwise convolution (tile.hip, convtile.hip) vs the gather kernels and the CPU oracle.

The tile plan must not change results beyond fp32 summation order: every case is run three ways — tiled, gather
kernels (``_plan.set_mode('direct')``) and oracle — at the 1e-5 bar of the convolution tests.  Small row capacities
force the plan's split (16 -> 8 -> ... -> 1 targets) and "direct sub-tile" paths.
"""
import numpy as np
import pytest
import torch

import oracle
from sph3d_gcn_amd import tf_nnquery, tf_buildkernel, tf_conv3d, _plan, _lib
from sph3d_gcn_amd.harness import synth

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _n(t):
    return t.detach().cpu().numpy()


def _cloud(kind, B, N, seed):
    if kind == "s3dis":
        return synth.s3dis_batch(seed, B, N)[0]
    if kind == "modelnet":
        return synth.modelnet_batch(seed, B, N)
    return synth.uniform_cloud(seed, B, N, 1.0)


def _hip_graph(dev, kind, B, N, radius, K, kernel, seed=3):
    xyz = _t(_cloud(kind, B, N, seed), dev)
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, radius, None, K)
    filt = tf_buildkernel.spherical_kernel(xyz, xyz, idx, cnt, dst, radius, kernel)
    return xyz, idx, cnt, filt


@pytest.fixture(autouse=True)
def _reset_plan_state():
    yield
    _plan.set_mode("auto")
    _plan.UCAP = 236
    _plan.clear()


def test_plan_structure(dev):
    """order is a permutation; every staged-slot byte of every (target, bin) group names the row its edge points to;
    pads name the zero row; split / direct sub-tiles appear when the capacity is small."""
    B, N, K, F = 2, 700, 32, 33
    xyz, idx, cnt, filt = _hip_graph(dev, "s3dis", B, N, 0.15, K, [8, 2, 2])
    for ucap in (236, 40, 8):
        _plan.clear()
        _plan.register_geometry(filt, xyz, xyz)
        order, desc, rows, pbounds, slotw, bounds, key, _ = _plan.forward_plan(idx, cnt, filt, F, ucap=ucap)
        torch.cuda.synchronize()
        o, d, rw = _n(order), _n(desc).reshape(B, -1, 33), _n(rows)
        pb = _n(pbounds).reshape(B, N, F + 1)
        sb = _n(slotw).view(np.uint8)
        bd, ky = _n(bounds).reshape(B, N, F + 1), _n(key)
        idx_n, cnt_n, filt_n = _n(idx), _n(cnt), _n(filt)
        for b in range(B):
            assert sorted(o[b].tolist()) == list(range(N))
        n_direct = n_split = 0
        for b in range(B):
            for c in range(d.shape[1]):
                g = d[b, c, 0]
                npts = min(16, N - 16 * c)
                assert g in (1, 2, 4, 8, 16)
                n_split += g < 16
                for s in range((npts + g - 1) // g):
                    U, uoff = d[b, c, 1 + 2 * s], d[b, c, 2 + 2 * s]
                    tg = o[b, 16 * c + s * g: 16 * c + min((s + 1) * g, npts)]
                    want = set()
                    for m in tg:
                        want.update(idx_n[b, m, :cnt_n[b, m]].tolist())
                    if U < 0:
                        n_direct += 1
                        assert len(want) > ucap
                        continue
                    assert U == len(want) <= ucap
                    ul = rw[uoff:uoff + U]
                    assert set(ul.tolist()) == want
                    for m in tg:
                        for f in range(F):
                            e0, e1 = bd[b, m, f], bd[b, m, f + 1]
                            p0, p1 = pb[b, m, f], pb[b, m, f + 1]
                            assert p1 - p0 == (e1 - e0 + 3) // 4
                            sl = sb[4 * p0:4 * p1]
                            assert (ul[sl[:e1 - e0]] == ky[e0:e1]).all()
                            assert (sl[e1 - e0:] == ucap).all()
                        # the CSR itself: edges of bin f, in slot order
                        c_m = cnt_n[b, m]
                        for f in np.unique(filt_n[b, m, :c_m]):
                            e0, e1 = bd[b, m, f], bd[b, m, f + 1]
                            assert (ky[e0:e1] == idx_n[b, m, :c_m][filt_n[b, m, :c_m] == f]).all()
        if ucap == 8:
            assert n_direct > 0
        if ucap == 40:
            assert n_split > 0


# (kind, B, N, radius, K, C, r, kernel)
TILED_CASES = [
    ("s3dis", 2, 2048, 0.1, 64, 128, 2, [8, 2, 2]),
    ("s3dis", 8, 1024, 0.15, 64, 64, 2, [8, 2, 2]),       # B % 8 == 0: XCD-affine item decode
    ("s3dis", 3, 700, 0.15, 32, 64, 1, [8, 2, 2]),
    ("modelnet", 2, 1500, 0.1, 48, 256, 2, [8, 2, 2]),
    ("uniform", 2, 900, 0.12, 40, 36, 2, [8, 2, 2]),      # partial channel slice (C = 36)
    ("uniform", 1, 333, 0.2, 64, 512, 1, [8, 2, 3]),      # 49 bins
    ("s3dis", 1, 4096, 0.1, 64, 32, 2, [8, 2, 1]),        # 17 bins
    ("uniform", 2, 500, 0.15, 24, 67, 1, [8, 2, 2]),      # odd channel count: zero-padded to 68
]


@pytest.mark.parametrize("ucap", [236, 24])
@pytest.mark.parametrize("case", TILED_CASES, ids=lambda c: "%s-B%d-N%d-r%g-K%d-C%d-r%d" % c[:7])
def test_tiled_conv_matches_gather_kernels_and_oracle(dev, case, ucap):
    kind, B, N, radius, K, C, r, kernel = case
    F = kernel[0] * kernel[1] * kernel[2] + 1
    _plan.UCAP = ucap
    xyz, idx, cnt, filt = _hip_graph(dev, kind, B, N, radius, K, kernel)
    rng = np.random.RandomState(C + r)
    x = rng.randn(B, N, C).astype(np.float32)
    w = rng.randn(F, C, r).astype(np.float32)
    go = rng.randn(B, N, C * r).astype(np.float32)
    out_o = oracle.depthwise_conv3d(x, w, _n(idx), _n(cnt), _n(filt))
    gi_o, gf_o = oracle.depthwise_conv3d_grad(x, w, go, _n(idx), _n(cnt), _n(filt))
    res = {}
    for mode in ("auto", "direct"):
        _plan.set_mode(mode)
        xt, wt = _t(x, dev).requires_grad_(True), _t(w, dev).requires_grad_(True)
        out = tf_conv3d.depthwise_conv3d(xt, wt, idx, cnt, filt)
        out.backward(_t(go, dev))
        res[mode] = (_n(out), _n(xt.grad), _n(wt.grad))
    assert _plan._fwd and _plan._bwd, "the tiled path did not run"
    si = max(1.0, float(np.abs(gi_o).max()))
    sf = max(1.0, float(np.abs(gf_o).max()))
    for mode in ("auto", "direct"):
        out, gi, gf = res[mode]
        np.testing.assert_allclose(out, out_o, **TOL)
        np.testing.assert_allclose(gi / si, gi_o / si, **TOL)
        np.testing.assert_allclose(gf / sf, gf_o / sf, **TOL)


def test_tiled_variants(dev):
    """every (channels per lane, waves per workgroup) build of the tiled kernels gives the same numbers"""
    kind, B, N, radius, K, C, r, kernel = TILED_CASES[0]
    xyz, idx, cnt, filt = _hip_graph(dev, kind, B, N, radius, K, kernel)
    rng = np.random.RandomState(1)
    x, w, go = (_t(rng.randn(*s).astype(np.float32), dev) for s in ((B, N, C), (33, C, r), (B, N, C * r)))
    _plan.set_mode("direct")
    ref = tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt)
    rgi, rgf = tf_conv3d.depthwise_conv3d_grad(x, w, go, idx, cnt, filt)
    for v in (208, 216, 408, 416):
        _plan.set_mode("auto", v)
        out = tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt)
        np.testing.assert_allclose(_n(out), _n(ref), **TOL)
        if v in (208, 216, 408):
            gi, gf = tf_conv3d.depthwise_conv3d_grad(x, w, go, idx, cnt, filt)
            s = float(rgf.abs().max())
            np.testing.assert_allclose(_n(gi), _n(rgi), rtol=1e-5, atol=1e-5 * float(rgi.abs().max()))
            np.testing.assert_allclose(_n(gf) / s, _n(rgf) / s, **TOL)
