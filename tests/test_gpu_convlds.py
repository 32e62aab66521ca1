"""Depthwise convolution gathered from LDS tiles (csrc/convlds.hip) vs the gather kernel and the CPU oracle.

The LDS kernel sums a point's neighbours bin group by bin group ((x_a + x_b) * w per pair), the gather kernels and the
oracle edge by edge: all three must agree within the 1e-5 bar of the convolution tests (north_star: "within 1e-5 fp32 on
conv activations").  The plan itself (tiles, row unions, bin-sorted pair records) is checked against a numpy restatement
of what it must contain.
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


@pytest.fixture(autouse=True)
def _reset_plan_state():
    yield
    _plan.set_mode("gather")
    _plan.clear()


def _graph(dev, kind, B, N, M, radius, K, kernel, seed=3, register=True):
    """intra graph (M == N) or an inter-level graph whose queries are the first M database points"""
    xyz = _t(_cloud(kind, B, N, seed), dev)
    q = xyz if M == N else xyz[:, :M].contiguous()
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, q, radius, None, K)
    _plan.set_mode("lds" if register else "gather")   # the binning op remembers the coordinates only while the LDS mode is on
    try:
        filt = tf_buildkernel.spherical_kernel(xyz, q, idx, cnt, dst, radius, kernel)
    finally:
        _plan.set_mode("gather")
    return xyz, q, idx, cnt, filt


def _check_plan(plan, idx, cnt, filt, N, F, order):
    """numpy restatement of the plan contract (include/sph3d.h: sph3d_conv_plan)"""
    hdr, rec, meta, rows = (_n(x) for x in plan)
    B, M, K = idx.shape
    CH, NP = 128, 48
    nch = (M + CH - 1) // CH
    ucap = _lib.lib().sph3d_conv_plan_ucap(F)
    hdr = hdr.reshape(B, nch, 132)
    rec = rec.view(np.uint32)[:B * nch * CH * 2 * NP].reshape(B, nch, CH, NP, 2)
    meta = meta.reshape(B, nch, CH, 2)
    rows = rows.view(np.uint16)[:B * nch * CH * 64].reshape(B, nch, CH * 64)
    idx = np.clip(idx, 0, N - 1)
    filt = np.clip(filt, 0, F - 1)
    cnt = np.clip(cnt, 0, min(K, 64))
    ntiles = 0
    sizes = []
    for b in range(B):
        seen = np.zeros(M, np.int32)
        for c in range(nch):
            npts = min(CH, M - CH * c)
            nt = hdr[b, c, 0]
            assert 1 <= nt <= min(npts, 64)
            cover = np.zeros(CH, np.int32)
            for t in range(nt):
                a, uoff = hdr[b, c, 1 + 2 * t], hdr[b, c, 2 + 2 * t]
                first, T, U = a & 0xff, (a >> 8) & 0xff, a >> 16
                assert 1 <= T <= 64 and 0 <= U <= ucap and first + T <= npts
                cover[first:first + T] += 1
                ms = meta[b, c, first:first + T, 0]
                cs = meta[b, c, first:first + T, 1] & 0xff
                ps = meta[b, c, first:first + T, 1] >> 8
                assert (np.diff(ps) <= 0).all(), "targets of a tile: most pairs first"
                # the tile's targets are the positions [128c + first, +T) of the order
                assert sorted(ms.tolist()) == sorted(order[b, CH * c + first:CH * c + first + T].tolist())
                want = set()
                for m, cm in zip(ms, cs):
                    assert cm == cnt[b, m]
                    want.update(idx[b, m, :cm].tolist())
                ul = rows[b, c, uoff:uoff + U]
                assert U == len(want) and ul.tolist() == sorted(want)
                for i, (m, cm, pm) in enumerate(zip(ms, cs, ps)):
                    e = rec[b, c, first + i]
                    sa, sb, bn = e[:, 0] & 0xffff, e[:, 0] >> 16, e[:, 1]
                    # the pairs restate the target's edges: per bin the multiset of its rows, zero-row partners for odd groups
                    got = {}
                    for p in range(pm):
                        assert sa[p] < U and 0 <= bn[p] < F
                        got.setdefault(int(bn[p]), []).append(int(ul[sa[p]]))
                        if sb[p] != ucap:
                            got[int(bn[p])].append(int(ul[sb[p]]))
                    ref = {}
                    for k in range(cm):
                        ref.setdefault(int(filt[b, m, k]), []).append(int(idx[b, m, k]))
                    assert {f: sorted(v) for f, v in got.items()} == {f: sorted(v) for f, v in ref.items()}
                    assert pm == sum((len(v) + 1) // 2 for v in ref.values()) <= NP
                    assert (sa[pm:] == ucap).all() and (sb[pm:] == ucap).all() and (bn[pm:] == F).all()
                    seen[m] += 1
                ntiles += 1
                sizes.append((T, U))
            assert (cover[:npts] == 1).all() and (cover[npts:] == 0).all()
        assert (seen == 1).all()
    return ntiles, sizes


@pytest.mark.parametrize("case", [
    ("s3dis", 2, 1500, 1500, 0.15, 64, [8, 2, 2]),
    ("uniform", 3, 700, 333, 0.2, 32, [8, 2, 2]),         # inter-level graph (M != N), ragged last chunk
    ("modelnet", 1, 1024, 1024, 0.3, 64, [8, 4, 2]),      # 65 bins: the wide filter planes, smaller tiles
], ids=lambda c: "%s-B%d-N%d-M%d" % c[:4])
def test_plan_structure(dev, case):
    kind, B, N, M, radius, K, kernel = case
    F = kernel[0] * kernel[1] * kernel[2] + 1
    xyz, q, idx, cnt, filt = _graph(dev, kind, B, N, M, radius, K, kernel)
    plan = _plan.conv_plan(idx, cnt, filt, F, N)
    order = _n(_plan.spatial_order(q))                    # the binning op registered the query coordinates
    for b in range(B):
        assert sorted(order[b].tolist()) == list(range(M))
    torch.cuda.synchronize()
    nt, sizes = _check_plan(plan, _n(idx), _n(cnt), _n(filt), N, F, order)
    assert nt >= B * ((M + 127) // 128)
    if kind == "s3dis":
        # spatial tiles: many targets per tile and real row reuse (an index-order plan of this cloud has ~3 targets per tile)
        T = np.array([s[0] for s in sizes])
        assert T.mean() > 24


# (kind, B, N, M, radius, K, C, r, kernel)
LDS_CASES = [
    ("s3dis", 2, 2048, 2048, 0.1, 64, 128, 2, [8, 2, 2]),
    ("s3dis", 8, 1024, 1024, 0.15, 64, 256, 2, [8, 2, 2]),     # B % 8 == 0: XCD-affine chunk ranges; four channel slices
    ("s3dis", 3, 700, 700, 0.15, 32, 64, 1, [8, 2, 2]),        # r = 1, one slice, K < 64
    ("modelnet", 2, 1500, 1500, 0.1, 48, 512, 2, [8, 2, 2]),
    ("uniform", 2, 900, 300, 0.12, 40, 192, 2, [8, 2, 2]),     # inter-level graph
    ("uniform", 1, 333, 333, 0.2, 64, 128, 1, [8, 2, 3]),      # 49 bins -> wide filter planes
    ("s3dis", 1, 4096, 4096, 0.1, 64, 64, 2, [8, 2, 1]),       # 17 bins
    ("uniform", 16, 128, 128, 0.3, 64, 1024, 2, [8, 2, 2]),    # a deep level: 16 slices, two chunks per cloud
    ("uniform", 5, 100, 70, 0.5, 64, 128, 2, [8, 4, 2]),       # 65 bins, dense neighbourhoods, B % 8 != 0
]


@pytest.mark.parametrize("case", LDS_CASES, ids=lambda c: "%s-B%d-N%d-M%d-r%g-K%d-C%d-r%d" % c[:8])
def test_lds_conv_equals_gather_kernel_and_oracle(dev, case):
    kind, B, N, M, radius, K, C, r, kernel = case
    F = kernel[0] * kernel[1] * kernel[2] + 1
    xyz, q, idx, cnt, filt = _graph(dev, kind, B, N, M, radius, K, kernel)
    rng = np.random.RandomState(C + r)
    x = rng.randn(B, N, C).astype(np.float32)
    w = rng.randn(F, C, r).astype(np.float32)
    out_o = oracle.depthwise_conv3d(x, w, _n(idx), _n(cnt), _n(filt))
    res = {}
    for mode in ("lds", "gather"):
        _plan.set_mode(mode)
        res[mode] = _n(tf_conv3d.depthwise_conv3d(_t(x, dev), _t(w, dev), idx, cnt, filt))
    assert _plan._plans, "the LDS path did not run"
    np.testing.assert_allclose(res["lds"], res["gather"], **TOL)
    np.testing.assert_allclose(res["lds"], out_o, **TOL)


def test_lds_conv_empty_rows_and_out_of_range_ids(dev):
    """rows with no neighbours give 0; out-of-range bins are clamped like the gather kernel does"""
    B, N, K, C, r, F = 2, 300, 64, 64, 2, 33
    xyz, q, idx, cnt, filt = _graph(dev, "uniform", B, N, N, 0.2, K, [8, 2, 2])
    cnt = cnt.clone()
    cnt[0, 5] = 0
    cnt[1, 17] = 0
    filt = filt.clone()
    filt[0, 7, 0] = 99
    filt[1, 3, 1] = -4
    x, w = torch.randn(B, N, C, device=dev), torch.randn(F, C, r, device=dev)
    _plan.set_mode("lds")
    a = tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt)
    assert _plan._plans
    _plan.set_mode("gather")
    b = tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt)
    np.testing.assert_allclose(_n(a), _n(b), **TOL)
    assert float(a[0, 5].abs().max()) == 0.0 and float(a[1, 17].abs().max()) == 0.0


def test_lds_conv_over_two_inputs_equals_concatenated_input(dev):
    B, N, K, r, F = 2, 640, 64, 2, 33
    xyz, q, idx, cnt, filt = _graph(dev, "s3dis", B, N, N, 0.2, K, [8, 2, 2])
    a, b = torch.randn(B, N, 128, device=dev), torch.randn(B, N, 64, device=dev)
    w = torch.randn(F, 192, r, device=dev)
    _plan.set_mode("lds")
    o1 = tf_conv3d.depthwise_conv3d_concat(a, b, w, idx, cnt, filt)
    assert _plan._plans
    _plan.set_mode("gather")
    o2 = tf_conv3d.depthwise_conv3d(torch.cat((a, b), 2), w, idx, cnt, filt)
    np.testing.assert_allclose(_n(o1), _n(o2), **TOL)


def test_lds_mode_falls_back_where_it_does_not_apply(dev):
    """odd channel counts and K > 64 run the gather kernels in 'lds' mode too; an unregistered graph gets an index-order plan"""
    xyz, q, idx, cnt, filt = _graph(dev, "uniform", 2, 400, 400, 0.15, 32, [8, 2, 2], register=False)
    _plan.set_mode("lds")
    x, w = torch.randn(2, 400, 36, device=dev), torch.randn(33, 36, 2, device=dev)
    out = tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt)
    assert not _plan._plans and out.shape == (2, 400, 72)
    x, w = torch.randn(2, 400, 128, device=dev), torch.randn(33, 128, 2, device=dev)
    a = tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt)           # nobody registered coordinates: index order
    assert _plan._plans and not _plan._orders
    _plan.set_mode("gather")
    b = tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt)
    np.testing.assert_allclose(_n(a), _n(b), **TOL)


def test_lds_conv_bench_shape(dev):
    """the headline shape (16 x 8192 points, K = 64, C = 128 and 64, r = 2) through the fused graph builder with the plan
    prebuilt on the graph path: equal to the gather kernel and, on one cloud's slice, to the oracle within 1e-5"""
    B, N, K, F = 16, 8192, 64, 33
    xyz = _t(synth.s3dis_batch(1000, B, N)[0], dev)
    _plan.set_mode("lds")
    idx, cnt, dst, filt = tf_nnquery.build_sphere_graph(xyz, 0.1, K, [8, 2, 2], with_transpose=False)
    assert _plan._plans, "build_sphere_graph prebuilds the plan in LDS mode"
    for C in (128, 64):
        x, w = torch.randn(B, N, C, device=dev), torch.randn(F, C, 2, device=dev)
        _plan.set_mode("lds")
        a = tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt)
        _plan.set_mode("gather")
        b = tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt)
        np.testing.assert_allclose(_n(a), _n(b), **TOL)
        c = 11
        o = oracle.depthwise_conv3d(_n(x[c:c + 1]), _n(w), _n(idx[c:c + 1]), _n(cnt[c:c + 1]), _n(filt[c:c + 1]))
        np.testing.assert_allclose(_n(a[c:c + 1]), o, **TOL)
