"""LDS-tiled depthwise convolution forward (tile.hip, convtile.hip) vs the gather kernel and the CPU oracle.

The tile plan must not change results beyond fp32 summation order: every case is run three ways — tiled
(``_plan.set_mode('tiled')``), gather kernel and oracle — at the 1e-5 bar of the convolution tests.  Small row
capacities force the plan's split (16 -> 8 -> ... -> 1 targets) path.
"""
import numpy as np
import pytest
import torch

import oracle
from sph3d_gcn_amd import tf_nnquery, tf_buildkernel, tf_conv3d, _plan
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
    _plan.set_mode("tiled")               # the binning op remembers the graph's coordinates only while a tiled mode is on
    try:
        filt = tf_buildkernel.spherical_kernel(xyz, xyz, idx, cnt, dst, radius, kernel)
    finally:
        _plan.set_mode("gather")
    return xyz, idx, cnt, filt


@pytest.fixture(autouse=True)
def _reset_plan_state():
    yield
    _plan.set_mode("gather")
    _plan.UCAP = 236
    _plan.clear()


def _check_plan(B, T, F, ucap, plan, lists, order):
    """plan = (hdr, tgt, rows, pb, slotw, xsteps, counters); lists(b, t) -> {bin: [source rows in CSR order]} of target t"""
    hdr, tgt, rows, pb, slotw, xsteps, counters = (_n(x) for x in plan)
    cands = (T + 15) // 16
    hdr = hdr.reshape(B, cands, 2)
    tg = tgt.reshape(B, cands, 16)
    pb = pb.reshape(B, cands, 16, F + 2)
    sb = slotw.view(np.uint8).reshape(B, cands, 16, 256)
    xs = xsteps.reshape(B, cands * 16, 8)
    o = _n(order)
    n_direct = n_split = 0

    def check_subtile(b, c, q0, cnt, U, ul):
        tl = tg[b, c, q0:q0 + cnt]
        per = [lists(b, t) for t in tl]
        want = set()
        for l in per:
            for v in l.values():
                want.update(v)
        assert U == len(want) <= ucap and set(ul.tolist()) == want
        for i, l in enumerate(per):
            assert pb[b, c, q0 + i, F + 1] == sum(len(v) for v in l.values())
            for f in range(F):
                src = l.get(f, [])
                p0, p1 = pb[b, c, q0 + i, f], pb[b, c, q0 + i, f + 1]
                assert p1 - p0 == (len(src) + 3) // 4 and p1 <= 64
                sl = sb[b, c, q0 + i, 4 * p0:4 * p1]
                assert (ul[sl[:len(src)]] == np.asarray(src, dtype=np.int64)).all()
                assert (sl[len(src):] == ucap).all()

    for b in range(B):
        assert sorted(o[b].tolist()) == list(range(T))
        covered = np.zeros((cands, 16), dtype=int)
        for c in range(cands):
            npts = min(16, T - 16 * c)
            assert sorted(tg[b, c, :npts].tolist()) == sorted(o[b, 16 * c:16 * c + npts].tolist())
            cnt0, U0 = hdr[b, c]
            if cnt0:
                check_subtile(b, c, 0, cnt0, U0, rows[(b * cands + c) * ucap:(b * cands + c) * ucap + U0])
                covered[c, :cnt0] += 1
        for i in range(counters[1 + b]):
            c, q0, cnt, U, uoff = xs[b, i, :5]
            covered[c, q0:q0 + cnt] += 1
            if U < 0:
                n_direct += 1
                l = lists(b, tg[b, c, q0])
                assert cnt == 1 and sum(len(v) for v in l.values()) > ucap
            else:
                n_split += 1
                assert uoff >= B * cands * ucap
                check_subtile(b, c, q0, cnt, U, rows[uoff:uoff + U])
        for c in range(cands):
            npts = min(16, T - 16 * c)
            assert (covered[c, :npts] == 1).all() and (covered[c, npts:] == 0).all()
    return n_direct, n_split


def test_plan_structure(dev):
    """order is a permutation; the tile lists cover it exactly once (first sub-tiles + extra steps); every staged-slot byte
    of every (target, bin) group names the row its edge points to; pads name the zero row; the binned CSR keeps the slot
    order inside a bin; split / direct steps appear when the capacity is small."""
    B, N, K, F = 2, 700, 32, 33
    xyz, idx, cnt, filt = _hip_graph(dev, "s3dis", B, N, 0.15, K, [8, 2, 2])
    idx_n, cnt_n, filt_n = _n(idx), _n(cnt), _n(filt)

    def fwd_lists(b, m):
        c = cnt_n[b, m]
        return {int(f): idx_n[b, m, :c][filt_n[b, m, :c] == f].tolist() for f in np.unique(filt_n[b, m, :c])}

    for ucap in (236, 40, 8):
        _plan.clear()
        _plan.register_geometry(filt, xyz, xyz)
        hdr, tgt, rows, pb, slotw, xsteps, counters, bounds, key, _ = _plan.forward_plan(idx, cnt, filt, F, ucap=ucap)
        order = _plan.spatial_order(xyz)
        torch.cuda.synchronize()
        bd, ky = _n(bounds).reshape(B, N, F + 1), _n(key)
        for b in range(B):
            for m in range(0, N, 13):
                for f, src in fwd_lists(b, m).items():
                    assert ky[bd[b, m, f]:bd[b, m, f + 1]].tolist() == src
        nd, ns = _check_plan(B, N, F, ucap, (hdr, tgt, rows, pb, slotw, xsteps, counters), fwd_lists, order)
        assert (nd > 0) == (ucap == 8) and (ns > 0 or ucap == 236)


# (kind, B, N, radius, K, C, r, kernel)
TILED_CASES = [
    ("s3dis", 2, 2048, 0.1, 64, 128, 2, [8, 2, 2]),
    ("s3dis", 8, 1024, 0.15, 64, 256, 2, [8, 2, 2]),      # B % 8 == 0: XCD-affine item decode; two channel slices
    ("s3dis", 3, 700, 0.15, 32, 128, 1, [8, 2, 2]),
    ("modelnet", 2, 1500, 0.1, 48, 512, 2, [8, 2, 2]),
    ("uniform", 2, 900, 0.12, 40, 132, 2, [8, 2, 2]),     # partial last channel slice (C = 132)
    ("uniform", 1, 333, 0.2, 64, 512, 1, [8, 2, 3]),      # 49 bins
    ("s3dis", 1, 4096, 0.1, 64, 128, 2, [8, 2, 1]),       # 17 bins
    ("uniform", 2, 500, 0.15, 24, 131, 1, [8, 2, 2]),     # odd channel count: zero-padded to 132
]


@pytest.mark.parametrize("ucap", [236, 24])
@pytest.mark.parametrize("case", TILED_CASES, ids=lambda c: "%s-B%d-N%d-r%g-K%d-C%d-r%d" % c[:7])
def test_tiled_conv_matches_gather_kernel_and_oracle(dev, case, ucap):
    kind, B, N, radius, K, C, r, kernel = case
    F = kernel[0] * kernel[1] * kernel[2] + 1
    if ucap < K:
        ucap = (K + 3) // 4 * 4          # the tiled kernel needs a capacity of at least one point's neighbours
    _plan.UCAP = ucap
    xyz, idx, cnt, filt = _hip_graph(dev, kind, B, N, radius, K, kernel)
    rng = np.random.RandomState(C + r)
    x = rng.randn(B, N, C).astype(np.float32)
    w = rng.randn(F, C, r).astype(np.float32)
    out_o = oracle.depthwise_conv3d(x, w, _n(idx), _n(cnt), _n(filt))
    res = {}
    for mode in ("tiled", "gather"):
        _plan.set_mode(mode)
        res[mode] = _n(tf_conv3d.depthwise_conv3d(_t(x, dev), _t(w, dev), idx, cnt, filt))
    assert _plan._fwd, "the tiled path did not run"
    for mode in ("tiled", "gather"):
        np.testing.assert_allclose(res[mode], out_o, **TOL)


def test_tiled_mode_falls_back_where_it_does_not_apply(dev):
    """narrow layers (C < 128) and graphs nobody registered coordinates for run the gather kernel in 'tiled' mode too"""
    xyz, idx, cnt, filt = _hip_graph(dev, "uniform", 2, 400, 0.15, 32, [8, 2, 2])
    _plan.set_mode("tiled")
    x, w = torch.randn(2, 400, 64, device=dev), torch.randn(33, 64, 2, device=dev)
    out = tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt)
    assert not _plan._fwd
    filt2 = filt.clone()                                  # same values, unknown tensor: no geometry
    x, w = torch.randn(2, 400, 128, device=dev), torch.randn(33, 128, 2, device=dev)
    a = tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt2)
    assert not _plan._fwd
    b = tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt)
    assert _plan._fwd
    np.testing.assert_allclose(_n(a), _n(b), **TOL)
    assert out.shape == (2, 400, 128)
