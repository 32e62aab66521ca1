"""Parity of the HIP kernels (through the C ABI / Python ops) with the CPU oracle and, when the
prebuilt oracle/_ref library travelled with the snapshot, with the reference's own kernels.

Bars: bit-exact for neighbour indices, counts, bin ids, FPS indices, arg-max ids (and nn_dist, whose
arithmetic is IEEE sqrt only); 1e-5 (abs+rel) for convolution / pooling activations and gradients.
"""
import numpy as np
import pytest
import torch

import oracle
from _errors import assert_per_element
from oracle import ref_gpu
from sph3d_gcn_amd import tf_nnquery, tf_buildkernel, tf_conv3d, tf_pool3d, tf_unpool3d, tf_sample, _lib
from sph3d_gcn_amd.harness import synth

pytestmark = pytest.mark.gpu

TOL = dict(rtol=1e-5, atol=1e-5)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _n(t):
    return t.detach().cpu().numpy()


def _clouds(kind, B, N, seed=0):
    if kind == "uniform":
        return synth.uniform_cloud(seed, B, N, 1.0)
    if kind == "s3dis":
        return synth.s3dis_batch(seed, B, N)[0]
    if kind == "modelnet":
        return synth.modelnet_batch(seed, B, N)
    raise ValueError(kind)


# (kind, B, N, M (None = intra), radius, K)
NN_CASES = [
    ("uniform", 1, 16, None, 0.3, 4),          # tiny
    ("uniform", 2, 100, 37, 0.2, 8),           # ragged sizes, inter graph
    ("uniform", 33, 1100, None, 0.08, 8),      # both i += 32 and j += 1024 radius carries
    ("s3dis", 2, 2048, None, 0.1, 64),         # on-grid data, K = 64
    ("modelnet", 3, 1024, None, 0.1, 64),
    ("uniform", 4, 1500, 3000, 0.05, 16),      # decoder-like: db != query, isolated queries need growth passes
    ("uniform", 1, 13000, 200, 0.05, 32),      # N > one LDS chunk (multi-chunk path)
    ("uniform", 2, 500, 40, 0.01, 70),         # K > 64, many growth passes
    ("uniform", 1, 10000, 300, 0.04, 32),      # one LDS chunk of 79 trips: the hit-mask window (64 trips) is flushed inside a chunk
    ("uniform", 1, 300, 70000, 0.05, 8),       # 69 queries per chain: positions of the radius sequence beyond the tabulated 64
    ("uniform", 2, 700, 90, 0.3, 200),         # K = 200: the slot lists do not fit LDS, slots are written from inside the scan
]


@pytest.mark.parametrize("case", NN_CASES, ids=lambda c: "%s-B%d-N%d-M%s-r%g-K%d" % c)
def test_sphere_neighbor_and_kernel_bitexact(dev, case):
    kind, B, N, M, radius, K = case
    db = _clouds(kind, B, N, seed=5)
    q = db if M is None else _clouds("uniform", B, M, seed=9) * db.max()
    idx_o, cnt_o, dst_o = oracle.build_sphere_neighbor(db, q, radius, None, K)
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(_t(db, dev), _t(q, dev), radius, None, K)
    np.testing.assert_array_equal(_n(cnt), cnt_o)
    np.testing.assert_array_equal(_n(idx), idx_o)
    np.testing.assert_array_equal(_n(dst).view(np.int32), dst_o.view(np.int32))   # bit pattern
    for kernel in ([8, 2, 2], [8, 2, 3], [4, 2, 1]):
        f_o = oracle.spherical_kernel(db, q, idx_o, cnt_o, dst_o, radius, kernel)
        f = tf_buildkernel.spherical_kernel(_t(db, dev), _t(q, dev), idx, cnt, dst, radius, kernel)
        np.testing.assert_array_equal(_n(f), f_o)
    if ref_gpu.available():
        ridx, rcnt, rdst = ref_gpu.build_sphere_neighbor(_t(db, dev), _t(q, dev), radius, None, K)
        np.testing.assert_array_equal(_n(rcnt), cnt_o)
        np.testing.assert_array_equal(_n(ridx), idx_o)
        np.testing.assert_array_equal(_n(rdst).view(np.int32), dst_o.view(np.int32))
        # bins: in "ocml" mode (the device-library atan2f the reference's kernel calls when built for this GPU) the HIP
        # kernel equals the reference build bit for bit; in the default mode (shared correctly rounded atan2f = oracle)
        # every differing entry is a neighbour whose exact angle lies within one float ulp of an angular bin boundary
        for kernel in ([8, 2, 2], [8, 2, 3]):
            rf = _n(ref_gpu.spherical_kernel(_t(db, dev), _t(q, dev), ridx, rcnt, rdst, radius, kernel))
            tf_buildkernel.set_atan2("ocml")
            try:
                fo = _n(tf_buildkernel.spherical_kernel(_t(db, dev), _t(q, dev), idx, cnt, dst, radius, kernel))
            finally:
                tf_buildkernel.set_atan2("shared")
            np.testing.assert_array_equal(fo, rf)
            f_o = oracle.spherical_kernel(db, q, idx_o, cnt_o, dst_o, radius, kernel)
            _assert_mismatches_on_boundaries(db, q, idx_o, f_o, rf, kernel)


def _assert_mismatches_on_boundaries(db, q, idx, ours, ref, kernel):
    n, p, _q = kernel
    for b, m, k in np.argwhere(ours != ref):
        a, r_ = int(ours[b, m, k]) - 1, int(ref[b, m, k]) - 1
        assert a >= 0 and r_ >= 0 and a // (n * p) == r_ // (n * p)        # never self (0) nor the radial shell
        pt, qp = db[b, idx[b, m, k]], q[b, m]
        dx, dy, dz = (np.float32(pt[i]) - np.float32(qp[i]) for i in range(3))
        if a % n != r_ % n:
            ang, cell = np.arctan2(np.float64(dy), np.float64(dx)) + np.pi, 2 * np.pi / n
        else:
            d2 = np.sqrt(np.float32(dx * dx + dy * dy), dtype=np.float32)
            ang, cell = np.arctan2(np.float64(dz), np.float64(d2)) + np.pi / 2, np.pi / p
        assert abs(ang - np.round(ang / cell) * cell) <= 4.8e-7, "bin differs from the reference build off a boundary"


@pytest.mark.skipif(not ref_gpu.available(), reason="oracle/_ref (the reference built for gfx950) did not travel")
def test_bins_ocml_mode_equal_reference_build_full_batch(dev):
    """the headline data family at the headline size: 16 x 8192-point S3DIS-like blocks, K = 64"""
    xyz = _t(synth.s3dis_batch(1000, 16, 8192)[0], dev)
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, 0.1, None, 64)
    rf = ref_gpu.spherical_kernel(xyz, xyz, idx, cnt, dst, 0.1, [8, 2, 2])
    tf_buildkernel.set_atan2("ocml")
    try:
        fo = tf_buildkernel.spherical_kernel(xyz, xyz, idx, cnt, dst, 0.1, [8, 2, 2])
    finally:
        tf_buildkernel.set_atan2("shared")
    assert torch.equal(fo, rf)
    fs = tf_buildkernel.spherical_kernel(xyz, xyz, idx, cnt, dst, 0.1, [8, 2, 2])
    ndiff = int((fs != rf).sum())
    assert 0 < ndiff < 1e-3 * fs.numel()                     # measured: 5 616 boundary neighbours in 8.4 M slots (0.07 %)
    _assert_mismatches_on_boundaries(_n(xyz), _n(xyz), _n(idx), _n(fs), _n(rf), [8, 2, 2])


@pytest.mark.parametrize("case", NN_CASES, ids=lambda c: "%s-B%d-N%d-M%s-r%g-K%d" % c)
def test_sphere_neighbor_fixed_radius_mode(dev, case):
    """the labelled non-reference mode (radius reset for every query) against the oracle's restatement of it"""
    kind, B, N, M, radius, K = case
    db = _clouds(kind, B, N, seed=5)
    q = db if M is None else _clouds("uniform", B, M, seed=9) * db.max()
    idx_o, cnt_o, dst_o = oracle.build_sphere_neighbor(db, q, radius, None, K, fixed=True)
    tf_nnquery.set_radius_mode("fixed")
    try:
        idx, cnt, dst = tf_nnquery.build_sphere_neighbor(_t(db, dev), _t(q, dev), radius, None, K)
    finally:
        tf_nnquery.set_radius_mode("compat")
    np.testing.assert_array_equal(_n(cnt), cnt_o)
    np.testing.assert_array_equal(_n(idx), idx_o)
    np.testing.assert_array_equal(_n(dst).view(np.int32), dst_o.view(np.int32))


@pytest.mark.parametrize("case", [c for c in NN_CASES if c[3] is None] + [("s3dis", 16, 2048, None, 0.2, 64)],
                         ids=lambda c: "%s-B%d-N%d-M%s-r%g-K%d" % c)
def test_fused_graph_construction_equals_separate_ops(dev, case):
    """sph3d_build_sphere_graph: neighbour search + bins (+ the transposed graph's counting pass) in one
    kernel == build_sphere_neighbor -> spherical_kernel (-> graph_transpose), bit for bit; inside a (source, bin) segment
    of the transposed graph the order is the arrival order of an atomic in both builds, so segments are compared as sets."""
    from sph3d_gcn_amd import _tgraph
    kind, B, N, _M, radius, K = case
    xyz = _t(_clouds(kind, B, N, seed=5), dev)
    for kernel in ([8, 2, 2], [4, 2, 1]):
        F = kernel[0] * kernel[1] * kernel[2] + 1
        idx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, radius, None, K)
        filt = tf_buildkernel.spherical_kernel(xyz, xyz, idx, cnt, dst, radius, kernel)
        i2, c2, d2, f2 = tf_nnquery.build_sphere_graph(xyz, radius, K, kernel)
        assert torch.equal(i2, idx) and torch.equal(c2, cnt) and torch.equal(d2.view(torch.int32), dst.view(torch.int32))
        assert torch.equal(f2, filt)
        tg_a = _tgraph.transpose(idx, cnt, N, bin_index=filt, num_bins=F)
        tg_b = _tgraph.transpose(i2, c2, N, bin_index=f2, num_bins=F)      # cached by the fused op
        off_a, act_a, off_b, act_b = tg_a[0], tg_a[3], tg_b[0], tg_b[3]
        (key_a, sc_a), (key_b, sc_b) = _tgraph.entries(tg_a), _tgraph.entries(tg_b)       # (decoded: the entries may be packed)
        assert torch.equal(off_a, off_b) and torch.equal(act_a[:1 + int(act_a[0])], act_b[:1 + int(act_b[0])])
        oa, ka, kb = _n(off_a), _n(key_a), _n(key_b)
        sa, sb = _n(sc_a), _n(sc_b)
        seg = np.repeat(np.arange(len(oa) - 1), np.diff(oa).clip(min=0)) if B == 1 else None
        # sort the entries of every segment: lexsort by (segment id, key)
        total = B * N * K
        segid = np.zeros(total, np.int64)
        L = N * F
        for b in range(B):
            o = oa[b * (L + 1):(b + 1) * (L + 1)]
            segid[o[0]:o[-1]] = np.repeat(np.arange(L) + b * L, np.diff(o))
        used = np.zeros(total, bool)
        for b in range(B):
            o = oa[b * (L + 1):(b + 1) * (L + 1)]
            used[o[0]:o[-1]] = True
        pa = np.lexsort((ka[used], segid[used]))
        pb = np.lexsort((kb[used], segid[used]))
        np.testing.assert_array_equal(ka[used][pa], kb[used][pb])
        np.testing.assert_array_equal(sa[used][pa], sb[used][pb])


@pytest.mark.parametrize("B,N,M,K,L,G", [(2, 300, 100, 8, 0.3, 3), (1, 1000, 1000, 20, 0.1, 4), (3, 77, 5, 70, 0.9, 2)])
def test_cube_neighbor_bitexact(dev, B, N, M, K, L, G):
    db = _clouds("uniform", B, N, seed=2)
    q = _clouds("uniform", B, M, seed=3)
    idx_o, cnt_o = oracle.build_cube_neighbor(db, q, L, None, K, G)
    idx, cnt = tf_nnquery.build_cube_neighbor(_t(db, dev), _t(q, dev), L, None, K, G)
    np.testing.assert_array_equal(_n(cnt), cnt_o)
    np.testing.assert_array_equal(_n(idx), idx_o)
    if ref_gpu.available():
        ridx, rcnt = ref_gpu.build_cube_neighbor(_t(db, dev), _t(q, dev), L, None, K, G)
        np.testing.assert_array_equal(_n(rcnt), cnt_o)
        np.testing.assert_array_equal(_n(ridx), idx_o)


FPS_CASES = [("uniform", 2, 3, 3), ("uniform", 3, 100, 40), ("uniform", 2, 1024, 256), ("s3dis", 2, 2048, 768),
             ("uniform", 2, 1100, 300), ("modelnet", 2, 10000, 500), ("s3dis", 1, 8192, 2048),
             ("uniform", 1, 30000, 64)]


@pytest.mark.parametrize("case", FPS_CASES, ids=lambda c: "%s-B%d-N%d-m%d" % c)
def test_fps_bitexact(dev, case):
    kind, B, N, m = case
    pts = _clouds(kind, B, N, seed=21)
    want = oracle.farthest_point_sample(m, pts)
    got = _n(tf_sample.farthest_point_sample(m, _t(pts, dev)))
    np.testing.assert_array_equal(got, want)
    if ref_gpu.available():
        np.testing.assert_array_equal(_n(ref_gpu.farthest_point_sample(m, _t(pts, dev))), want)


def test_fps_tie_break_gpu(dev):
    rng = np.random.RandomState(0)
    pts = (rng.rand(1, 2048, 3).astype(np.float32) - 0.5) * 0.15
    pts[0, 0] = 0
    pts[0, 1030] = (1, 0, 0)
    pts[0, 7] = (0, 1, 0)
    assert _n(tf_sample.farthest_point_sample(3, _t(pts, dev)))[0].tolist() == [0, 1030, 7]
    # grid data: many exact ties
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(9), indexing="ij"), -1).reshape(1, -1, 3)
    g = (g * 0.03).astype(np.float32)
    want = oracle.farthest_point_sample(200, g)
    np.testing.assert_array_equal(_n(tf_sample.farthest_point_sample(200, _t(g, dev))), want)


def _graph(kind, B, N, M, K, radius, seed):
    db = _clouds(kind, B, N, seed=seed)
    q = db if M is None else db[:, :M].copy()
    idx, cnt, dst = oracle.build_sphere_neighbor(db, q, radius, None, K)
    filt = oracle.spherical_kernel(db, q, idx, cnt, dst, radius, [8, 2, 2])
    return db, q, idx, cnt, dst, filt


@pytest.mark.parametrize("C,r", [(64, 2), (128, 2), (32, 1), (256, 2)])
@pytest.mark.parametrize("nbins", [5, 17, 18, 33])
def test_conv_grad_all_bins_and_compact_bins(dev, C, r, nbins):
    """The gradient kernel has two launches: a compact one for graphs in which at most 17 of the bins occur (the usual
    case under the sqrt-distance quirk) and the full 33-bin one; the choice is made on the device.  Synthetic bin ids
    drawn from `nbins` distinct values exercise both, and the 17 / 18 boundary."""
    B, N, K = 2, 300, 24
    rng = np.random.RandomState(1000 * nbins + C + r)
    db, q, idx, cnt, dst, _ = _graph("uniform", B, N, None, K, 0.2, seed=nbins)
    allowed = np.sort(rng.permutation(33)[:nbins]).astype(np.int32)
    filt = allowed[rng.randint(0, nbins, size=idx.shape)].astype(np.int32)
    filt[np.arange(K)[None, None, :] >= cnt[:, :, None]] = 0       # unused slots read 0, as the op writes them
    x = rng.randn(B, N, C).astype(np.float32)
    w = rng.randn(33, C, r).astype(np.float32)
    go = rng.randn(B, N, C * r).astype(np.float32)
    gi_o, gf_o = oracle.depthwise_conv3d_grad(x, w, go, idx, cnt, filt)
    gi, gf = tf_conv3d.depthwise_conv3d_grad(_t(x, dev), _t(w, dev), _t(go, dev), _t(idx, dev), _t(cnt, dev), _t(filt, dev))
    np.testing.assert_allclose(_n(gi), gi_o, **TOL)
    s_ = max(1.0, float(np.abs(gf_o).max()))
    np.testing.assert_allclose(_n(gf) / s_, gf_o / s_, **TOL)
    used = np.unique(filt[np.arange(K)[None, None, :] < cnt[:, :, None]])
    unused = np.setdiff1d(np.arange(33), used)
    assert (_n(gf)[unused] == 0).all()                            # bins that never occur get an exact zero gradient


@pytest.mark.parametrize("C,r", [(36, 2), (64, 2), (64, 1), (32, 1), (128, 2), (16, 1)])
@pytest.mark.parametrize("hub_edges", [61, 64, 65, 127, 200])
def test_conv_grad_hub_source_with_one_long_segment(dev, C, r, hub_edges):
    """A source point that is the neighbour of many queries IN ONE BIN: its (source, bin) segment fills whole 64-edge chunks
    of the transposed graph.  The half- and quarter-wave gradient forms consume a segment in batches of 6 or 8 edges whose
    lane numbers wrap modulo 64 at the end of a chunk; a batch size that does not divide 64 once counted the first edges
    of such a chunk twice (segments of 61+ edges: found by the ModelNet full-size test only)."""
    B, N, K = 2, 256, 8
    rng = np.random.RandomState(hub_edges * 7 + C + r)
    cnt = rng.randint(1, K + 1, size=(B, N)).astype(np.int32)
    idx = np.zeros((B, N, K), np.int32)
    filt = np.zeros((B, N, K), np.int32)
    for b in range(B):
        for m in range(N):
            c = int(cnt[b, m])
            idx[b, m, :c] = np.sort(rng.permutation(np.arange(1, N))[:c])
            filt[b, m, :c] = rng.randint(0, 33, size=c)
        hub = rng.permutation(N)[:hub_edges]
        idx[b, hub, 0] = 0                                 # point 0 first (ascending order holds), always in bin 5
        filt[b, hub, 0] = 5
    x = rng.randn(B, N, C).astype(np.float32)
    w = rng.randn(33, C, r).astype(np.float32)
    go = rng.randn(B, N, C * r).astype(np.float32)
    gi_o, gf_o = oracle.depthwise_conv3d_grad(x, w, go, idx, cnt, filt)
    gi, gf = tf_conv3d.depthwise_conv3d_grad(_t(x, dev), _t(w, dev), _t(go, dev), _t(idx, dev), _t(cnt, dev), _t(filt, dev))
    s_i = max(1.0, float(np.abs(gi_o).max()))
    np.testing.assert_allclose(_n(gi) / s_i, gi_o / s_i, **TOL)
    s_ = max(1.0, float(np.abs(gf_o).max()))
    np.testing.assert_allclose(_n(gf) / s_, gf_o / s_, **TOL)


# (B, N, M, C, r, K)
CONV_CASES = [(2, 200, 100, 8, 2, 16), (1, 64, 64, 3, 1, 8), (2, 300, 300, 35, 2, 32), (2, 256, 256, 67, 1, 64),
              (2, 500, 500, 64, 2, 64), (1, 300, 150, 128, 2, 64), (1, 128, 128, 1024, 2, 64), (2, 200, 200, 131, 1, 16),
              (1, 100, 100, 64, 1, 70), (2, 128, 128, 6, 4, 16), (2, 300, 300, 128, 1, 48), (1, 257, 257, 32, 2, 40),
              (3, 150, 150, 60, 1, 24)]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "B%d-N%d-M%d-C%d-r%d-K%d" % c)
def test_depthwise_conv_forward_backward(dev, case):
    B, N, M, C, r, K = case
    rng = np.random.RandomState(C * 7 + r)
    db, q, idx, cnt, dst, filt = _graph("uniform", B, N, M if M != N else None, K, 0.25, seed=C)
    x = rng.randn(B, N, C).astype(np.float32)
    w = rng.randn(33, C, r).astype(np.float32)
    go = rng.randn(B, idx.shape[1], C * r).astype(np.float32)
    out_o = oracle.depthwise_conv3d(x, w, idx, cnt, filt)
    gi_o, gf_o = oracle.depthwise_conv3d_grad(x, w, go, idx, cnt, filt)
    xt = _t(x, dev).requires_grad_(True)
    wt = _t(w, dev).requires_grad_(True)
    out = tf_conv3d.depthwise_conv3d(xt, wt, _t(idx, dev), _t(cnt, dev), _t(filt, dev))
    np.testing.assert_allclose(_n(out), out_o, **TOL)
    out.backward(_t(go, dev))
    scale_i = max(1.0, float(np.abs(gi_o).max()))
    scale_f = max(1.0, float(np.abs(gf_o).max()))
    np.testing.assert_allclose(_n(xt.grad) / scale_i, gi_o / scale_i, **TOL)
    np.testing.assert_allclose(_n(wt.grad) / scale_f, gf_o / scale_f, **TOL)
    # per element, against the sum of the magnitudes of each element's terms (tests/_errors.py)
    mi, mf = oracle.depthwise_conv3d_grad(np.abs(x), np.abs(w), np.abs(go), idx, cnt, filt)
    assert_per_element(_n(xt.grad), gi_o, mi, "conv grad_input %s" % (case,))
    assert_per_element(_n(wt.grad), gf_o, mf, "conv grad_filter %s" % (case,))
    assert_per_element(_n(out), out_o, oracle.depthwise_conv3d(np.abs(x), np.abs(w), idx, cnt, filt), "conv forward %s" % (case,))
    if ref_gpu.available():
        rout = ref_gpu.depthwise_conv3d(_t(x, dev), _t(w, dev), _t(idx, dev), _t(cnt, dev), _t(filt, dev))
        np.testing.assert_allclose(_n(out), _n(rout), **TOL)
        rgi, rgf = ref_gpu.depthwise_conv3d_grad(_t(x, dev), _t(w, dev), _t(go, dev), _t(idx, dev), _t(cnt, dev),
                                                 _t(filt, dev))
        np.testing.assert_allclose(_n(xt.grad) / scale_i, _n(rgi) / scale_i, **TOL)
        np.testing.assert_allclose(_n(wt.grad) / scale_f, _n(rgf) / scale_f, **TOL)


POOL_CASES = [(2, 200, 100, 8, 16), (1, 64, 20, 3, 8), (2, 300, 120, 67, 32), (2, 500, 200, 128, 64),
              (1, 300, 100, 512, 64), (1, 90, 30, 5, 70)]


@pytest.mark.parametrize("case", POOL_CASES, ids=lambda c: "B%d-N%d-M%d-C%d-K%d" % c)
def test_pooling(dev, case):
    B, N, M, C, K = case
    rng = np.random.RandomState(C)
    db, q, idx, cnt, dst, filt = _graph("uniform", B, N, M, K, 0.25, seed=C + 1)
    x = rng.randn(B, N, C).astype(np.float32)
    x[:, ::3] = np.round(x[:, ::3])           # exact ties for the arg-max rule
    go = rng.randn(B, M, C).astype(np.float32)
    out_o, mi_o = oracle.max_pool3d(x, idx, cnt)
    xt = _t(x, dev).requires_grad_(True)
    out, mi = tf_pool3d.max_pool3d(xt, _t(idx, dev), _t(cnt, dev))
    np.testing.assert_array_equal(_n(out), out_o)        # pure selection: exact
    np.testing.assert_array_equal(_n(mi), mi_o)
    out.backward(_t(go, dev))
    np.testing.assert_allclose(_n(xt.grad), oracle.max_pool3d_grad(x, go, mi_o), **TOL)
    xt2 = _t(x, dev).requires_grad_(True)
    avg = tf_pool3d.avg_pool3d(xt2, _t(idx, dev), _t(cnt, dev))
    np.testing.assert_allclose(_n(avg), oracle.avg_pool3d(x, idx, cnt), **TOL)
    avg.backward(_t(go, dev))
    np.testing.assert_allclose(_n(xt2.grad), oracle.avg_pool3d_grad(x, go, idx, cnt), **TOL)
    if ref_gpu.available():
        rout, rmi = ref_gpu.max_pool3d(_t(x, dev), _t(idx, dev), _t(cnt, dev))
        np.testing.assert_array_equal(_n(rout), out_o)
        np.testing.assert_array_equal(_n(rmi), mi_o)
        np.testing.assert_allclose(_n(ref_gpu.avg_pool3d(_t(x, dev), _t(idx, dev), _t(cnt, dev))), _n(avg), **TOL)


@pytest.mark.parametrize("case", POOL_CASES, ids=lambda c: "B%d-N%d-M%d-C%d-K%d" % c)
def test_unpooling(dev, case):
    B, N, M, C, K = case
    rng = np.random.RandomState(C + 100)
    fine = _clouds("uniform", B, N, seed=C + 2)
    coarse = fine[:, :M].copy()
    idx, cnt, dst = oracle.build_sphere_neighbor(coarse, fine, 0.3, None, K)   # db = coarse, query = fine
    feat = rng.randn(B, M, C).astype(np.float32)
    go = rng.randn(B, N, C).astype(np.float32)
    w = (dst + 1e-7) / (dst.sum(-1, keepdims=True) + 1e-7)
    ft = _t(feat, dev).requires_grad_(True)
    mean = tf_unpool3d.mean_interpolate(ft, _t(idx, dev), _t(cnt, dev))
    np.testing.assert_allclose(_n(mean), oracle.mean_interpolate(feat, idx, cnt), **TOL)
    mean.backward(_t(go, dev))
    np.testing.assert_allclose(_n(ft.grad), oracle.mean_interpolate_grad(feat, go, idx, cnt), **TOL)
    ft2 = _t(feat, dev).requires_grad_(True)
    wi = tf_unpool3d.weighted_interpolate(ft2, _t(w, dev), _t(idx, dev), _t(cnt, dev))
    np.testing.assert_allclose(_n(wi), oracle.weighted_interpolate(feat, w, idx, cnt), **TOL)
    wi.backward(_t(go, dev))
    np.testing.assert_allclose(_n(ft2.grad), oracle.weighted_interpolate_grad(feat, go, w, idx, cnt), **TOL)
    if ref_gpu.available():
        np.testing.assert_allclose(_n(ref_gpu.mean_interpolate(_t(feat, dev), _t(idx, dev), _t(cnt, dev))), _n(mean), **TOL)
        np.testing.assert_allclose(_n(ref_gpu.weighted_interpolate(_t(feat, dev), _t(w, dev), _t(idx, dev), _t(cnt, dev))),
                                   _n(wi), **TOL)


@pytest.mark.parametrize("C", [8, 64, 67, 128, 256])
@pytest.mark.parametrize("hub_edges", [63, 64, 65, 130, 300])
def test_pool_unpool_gradients_hub_source(dev, C, hub_edges):
    """The pooling / un-pooling gradients gather over the transposed graph, 64 in-edges of a source at a time: one source point
    listed by `hub_edges` queries (whole chunks, a chunk + 1, several chunks) next to ordinary sparse rows."""
    from sph3d_gcn_amd import _tgraph
    B, N, M, K = 2, 320, 300, 6
    rng = np.random.RandomState(hub_edges * 11 + C)
    cnt = rng.randint(1, K + 1, size=(B, M)).astype(np.int32)
    idx = np.zeros((B, M, K), np.int32)
    for b in range(B):
        for m in range(M):
            c = int(cnt[b, m])
            idx[b, m, :c] = np.sort(rng.permutation(np.arange(1, N))[:c])
        idx[b, rng.permutation(M)[:hub_edges], 0] = 0          # the hub: point 0, first in its rows (rows stay ascending and unique)
    x = rng.randn(B, N, C).astype(np.float32)
    x[:, ::3] = np.round(x[:, ::3])
    go = rng.randn(B, M, C).astype(np.float32)
    it, ct = _t(idx, dev), _t(cnt, dev)
    # max pool: the gather form of the gradient (a transposed graph built ahead with the unique-rows promise) and the plain one
    out_o, mi_o = oracle.max_pool3d(x, idx, cnt)
    want = oracle.max_pool3d_grad(x, go, mi_o)
    for ahead in (False, True):
        _tgraph.clear()
        if ahead:
            _tgraph.transpose(it, ct, N, unique_rows=True)
        xt = _t(x, dev).requires_grad_(True)
        out, mi = tf_pool3d.max_pool3d(xt, it, ct)
        np.testing.assert_array_equal(_n(mi), mi_o)
        out.backward(_t(go, dev))
        s_ = max(1.0, float(np.abs(want).max()))               # (the hub's gradient is a sum of up to 300 terms)
        np.testing.assert_allclose(_n(xt.grad) / s_, want / s_, **TOL)
    xt2 = _t(x, dev).requires_grad_(True)
    tf_pool3d.avg_pool3d(xt2, it, ct).backward(_t(go, dev))
    ref = oracle.avg_pool3d_grad(x, go, idx, cnt)
    s_ = max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(_n(xt2.grad) / s_, ref / s_, **TOL)
    # un-pooling over the same rows: database = the N points (features [B, N, C]), queries = the M rows
    w = rng.rand(B, M, K).astype(np.float32)
    w[np.arange(K)[None, None, :] >= cnt[:, :, None]] = 0
    w /= w.sum(-1, keepdims=True)
    ft = _t(x, dev).requires_grad_(True)
    tf_unpool3d.mean_interpolate(ft, it, ct).backward(_t(go, dev))
    ref = oracle.mean_interpolate_grad(x, go, idx, cnt)
    s_ = max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(_n(ft.grad) / s_, ref / s_, **TOL)
    ft2 = _t(x, dev).requires_grad_(True)
    tf_unpool3d.weighted_interpolate(ft2, _t(w, dev), it, ct).backward(_t(go, dev))
    ref = oracle.weighted_interpolate_grad(x, go, w, idx, cnt)
    s_ = max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(_n(ft2.grad) / s_, ref / s_, **TOL)
    _tgraph.clear()


def test_device_scalar_math_matches_host_bitwise(dev):
    """sqrt / divide are correctly rounded on device and sph3d_atan2f is bit-identical host vs device."""
    import ctypes
    rng = np.random.RandomState(1)
    n = 1 << 20
    a = (rng.randn(n) * np.exp(rng.randn(n) * 3)).astype(np.float32)
    b = (rng.randn(n) * np.exp(rng.randn(n) * 3)).astype(np.float32)
    a[:1000] = np.round(a[:1000] * 33) * 0.03
    b[:1000] = np.round(b[:1000] * 33) * 0.03
    at, bt = _t(a, dev), _t(b, dev)
    o_at = torch.empty(n, device=dev)
    o_sq = torch.empty(n, device=dev)
    o_dv = torch.empty(n, device=dev)
    o_bin = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().sph3d_selftest_math(n, _lib.ptr(at), _lib.ptr(bt), _lib.ptr(o_at), _lib.ptr(o_sq),
                                              _lib.ptr(o_dv), _lib.ptr(o_bin), _lib.stream_ptr()))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(_n(o_sq).view(np.int32), np.sqrt(np.abs(a)).view(np.int32))
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(_n(o_dv).view(np.int32), (a / b).view(np.int32))
    want = np.arctan2(a.astype(np.float64), b.astype(np.float64)).astype(np.float32)
    np.testing.assert_array_equal(_n(o_at).view(np.int32), want.view(np.int32))
    # bins of the full scalar pipeline vs the oracle's C function
    sb = oracle.lib().oracle_sphere_bin
    sb.restype = ctypes.c_int
    got = _n(o_bin)
    f = ctypes.c_float
    for i in range(0, 20000):
        dist = np.sqrt(np.sqrt(np.float32(a[i] * a[i] + b[i] * b[i])))
        w = sb(f(a[i]), f(b[i]), f(np.float32(a[i] * b[i])), f(dist), f(0.1), 8, 2, 2)
        assert got[i] == w, i


def test_full_size_properties_s3dis_level0(dev):
    """BASELINE size (B=16, N=M=8192, K=64): size-independent properties instead of the slow oracle:
    counts in [1,K], ascending indices, zero padding, self is neighbour 0-bin, distances recomputable,
    bins in range; plus the oracle on a 2-cloud slice (chains do not cross clouds for B <= 32)."""
    B, N, K, r = 16, 8192, 64, 0.1
    xyz = synth.s3dis_batch(100, B, N)[0]
    xt = _t(xyz, dev)
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(xt, xt, r, None, K)
    filt = tf_buildkernel.spherical_kernel(xt, xt, idx, cnt, dst, r, [8, 2, 2])
    idx_n, cnt_n, dst_n, filt_n = _n(idx), _n(cnt), _n(dst), _n(filt)
    assert cnt_n.min() >= 1 and cnt_n.max() <= K
    ar = np.arange(K)[None, None, :]
    valid = ar < cnt_n[:, :, None]
    assert (idx_n[~valid] == 0).all() and (dst_n[~valid] == 0).all() and (filt_n[~valid] == 0).all()
    d = np.diff(idx_n, axis=2)
    assert (d[valid[:, :, 1:]] > 0).all()
    assert filt_n.min() >= 0 and filt_n.max() <= 32
    # every query contains itself (distance 0 -> bin 0)
    self_hit = ((idx_n == np.arange(N)[None, :, None]) & valid)
    assert self_hit.any(axis=2)[cnt_n < K].all()      # (a saturated list keeps the 64 lowest indices, maybe not self)
    assert (filt_n[self_hit] == 0).all()
    # distances: sqrt(sqrt(d2)) recomputed
    b_ix = np.arange(B)[:, None, None]
    nb = xyz[b_ix, idx_n]
    dd = nb - xyz[:, :, None, :]
    d2 = (dd[..., 0] * dd[..., 0] + dd[..., 1] * dd[..., 1]) + dd[..., 2] * dd[..., 2]
    np.testing.assert_array_equal(np.sqrt(np.sqrt(d2))[valid], dst_n[valid])
    # oracle on two clouds (same chains: B <= 32 so clouds are independent)
    for b in (0, 15):
        i_o, c_o, d_o = oracle.build_sphere_neighbor(xyz[b:b + 1], xyz[b:b + 1], r, None, K)
        np.testing.assert_array_equal(idx_n[b], i_o[0])
        np.testing.assert_array_equal(cnt_n[b], c_o[0])
        f_o = oracle.spherical_kernel(xyz[b:b + 1], xyz[b:b + 1], i_o, c_o, d_o, r, [8, 2, 2])
        np.testing.assert_array_equal(filt_n[b], f_o[0])
    # FPS at full size: indices distinct (points are distinct), first is 0, matches oracle on one cloud
    fps = _n(tf_sample.farthest_point_sample(2048, xt))
    assert (fps[:, 0] == 0).all()
    for b in range(B):
        assert len(set(fps[b].tolist())) == 2048
    np.testing.assert_array_equal(fps[3], oracle.farthest_point_sample(2048, xyz[3:4])[0])
    # conv at the roofline shape: linearity in the filter and in the input (size-independent property)
    C, rr = 128, 2
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(B, N, C, generator=g).to(dev)
    w1 = torch.randn(33, C, rr, generator=g).to(dev)
    w2 = torch.randn(33, C, rr, generator=g).to(dev)
    o1 = tf_conv3d.depthwise_conv3d(x, w1, idx, cnt, filt)
    o2 = tf_conv3d.depthwise_conv3d(x, w2, idx, cnt, filt)
    o12 = tf_conv3d.depthwise_conv3d(x, w1 + w2, idx, cnt, filt)
    torch.testing.assert_close(o12, o1 + o2, rtol=1e-4, atol=1e-4)
    out_o = oracle.depthwise_conv3d(_n(x[:1]), _n(w1), idx_n[:1], cnt_n[:1], filt_n[:1])
    np.testing.assert_allclose(_n(o1[:1]), out_o, **TOL)


def test_empty_and_error_cases(dev):
    z = torch.zeros((0, 8, 3), device=dev)
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(z, z, 0.1, None, 4)
    assert idx.shape == (0, 8, 4)
    x = torch.zeros((1, 8, 3), device=dev)
    with pytest.raises(ValueError):
        tf_nnquery.build_sphere_neighbor(x, x, -0.1, None, 4)
    with pytest.raises(ValueError):
        tf_nnquery.build_sphere_neighbor(x, x, 0.1, None, 0)
    with pytest.raises(ValueError):
        tf_sample.farthest_point_sample(0, x)
    with pytest.raises(ValueError):
        tf_sample.farthest_point_sample(2, torch.zeros((1, 8, 4), device=dev))
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(x, x, 0.1, None, 4)
    with pytest.raises(ValueError):
        tf_buildkernel.spherical_kernel(x, x, idx, cnt, dst, 0.1, [3, 2, 2])
    with pytest.raises(ValueError):
        tf_conv3d.depthwise_conv3d(torch.zeros((1, 8, 5), device=dev), torch.zeros((33, 4, 2), device=dev), idx, cnt, idx)


def test_cabi_gradient_wrappers_with_workspace(dev):
    """The reference-surface gradient entry points (which build the transposed graph in a caller workspace)
    called straight through the C ABI, plus the too-small-workspace error path."""
    import ctypes
    B, N, M, C, r, K = 2, 150, 60, 12, 2, 16
    rng = np.random.RandomState(5)
    db, q, idx, cnt, dst, filt = _graph("uniform", B, N, M, K, 0.25, seed=77)
    x = rng.randn(B, N, C).astype(np.float32)
    w = rng.randn(33, C, r).astype(np.float32)
    go = rng.randn(B, M, C * r).astype(np.float32)
    gp = rng.randn(B, M, C).astype(np.float32)
    l = _lib.lib()
    P = _lib.ptr
    xt, wt, got, gpt = _t(x, dev), _t(w, dev), _t(go, dev), _t(gp, dev)
    it, ct, ft = _t(idx, dev), _t(cnt, dev), _t(filt, dev)
    gi = torch.empty_like(xt)
    gf = torch.empty_like(wt)
    wsb = l.sph3d_depthwise_conv3d_grad_workspace(B, N, M, 33, C, r, K)
    assert wsb > 0
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    rc = l.sph3d_depthwise_conv3d_grad(B, N, M, 33, C, r, K, P(it), P(ct), P(ft), P(xt), P(wt), P(got), P(gi), P(gf),
                                       P(ws), wsb, _lib.stream_ptr())
    assert rc == 0
    gi_o, gf_o = oracle.depthwise_conv3d_grad(x, w, go, idx, cnt, filt)
    np.testing.assert_allclose(_n(gi), gi_o, **TOL)
    np.testing.assert_allclose(_n(gf), gf_o, rtol=1e-5, atol=2e-5)
    rc = l.sph3d_depthwise_conv3d_grad(B, N, M, 33, C, r, K, P(it), P(ct), P(ft), P(xt), P(wt), P(got), P(gi), P(gf),
                                       P(ws), 16, _lib.stream_ptr())
    assert rc == -2 and b"workspace" in l.sph3d_last_error()
    wsb = l.sph3d_scatter_grad_workspace(B, N, M, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    g2 = torch.empty_like(xt)
    assert l.sph3d_avg_pool3d_grad(B, N, M, C, K, P(it), P(ct), P(gpt), P(g2), P(ws), wsb, _lib.stream_ptr()) == 0
    np.testing.assert_allclose(_n(g2), oracle.avg_pool3d_grad(x, gp, idx, cnt), **TOL)
    # un-pooling wrappers: N fine points, M coarse
    uidx, ucnt, udst = oracle.build_sphere_neighbor(q, db, 0.3, None, K)
    wgt = ((udst + 1e-7) / (udst.sum(-1, keepdims=True) + 1e-7)).astype(np.float32)
    feat = x[:, :M].copy()
    gu = rng.randn(B, N, C).astype(np.float32)
    ut, uct, wgt_t, gut = _t(uidx, dev), _t(ucnt, dev), _t(wgt, dev), _t(gu, dev)
    wsb = l.sph3d_scatter_grad_workspace(B, M, N, K)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    g3 = torch.empty((B, M, C), device=dev)
    assert l.sph3d_mean_interpolate_grad(B, N, M, C, K, P(ut), P(uct), P(gut), P(g3), P(ws), wsb, _lib.stream_ptr()) == 0
    np.testing.assert_allclose(_n(g3), oracle.mean_interpolate_grad(feat, gu, uidx, ucnt), **TOL)
    assert l.sph3d_weighted_interpolate_grad(B, N, M, C, K, P(ut), P(uct), P(gut), P(wgt_t), P(g3), P(ws), wsb,
                                             _lib.stream_ptr()) == 0
    np.testing.assert_allclose(_n(g3), oracle.weighted_interpolate_grad(feat, gu, wgt, uidx, ucnt), **TOL)


GEMM_CASES = [(1000, 3, 64), (4096, 128, 128), (3000, 256, 13), (777, 70, 64), (2048, 1024, 512), (130, 131, 128),
              (5000, 2048, 256), (16384, 64, 40)]


@pytest.mark.parametrize("case", GEMM_CASES, ids=lambda c: "R%d-Cin%d-Cout%d" % c)
def test_pointwise_gemm_forward_backward(dev, case):
    """fp32 MFMA GEMM (exact fp32) vs float64 numpy: 1e-5 relative to the row scale; asymmetric operands so a
    transposed fragment / C-layout mix-up cannot pass."""
    from sph3d_gcn_amd import tf_gemm
    R, Cin, Cout = case
    rng = np.random.RandomState(R + Cin)
    x = rng.randn(R, Cin).astype(np.float32)
    w = (rng.randn(Cin, Cout) * (1 + np.arange(Cout))[None, :] * 0.1).astype(np.float32)
    dy = rng.randn(R, Cout).astype(np.float32)
    xt = _t(x, dev).requires_grad_(True)
    wt = _t(w, dev).requires_grad_(True)
    y = tf_gemm._pointwise_gemm(xt, wt, False)
    y64 = x.astype(np.float64) @ w.astype(np.float64)
    s = max(1.0, float(np.abs(y64).max()))
    np.testing.assert_allclose(_n(y) / s, y64 / s, rtol=1e-5, atol=1e-5)
    y.backward(_t(dy, dev))
    dx64 = dy.astype(np.float64) @ w.astype(np.float64).T
    dw64 = x.astype(np.float64).T @ dy.astype(np.float64)
    s = max(1.0, float(np.abs(dx64).max()))
    np.testing.assert_allclose(_n(xt.grad) / s, dx64 / s, rtol=1e-5, atol=1e-5)
    s = max(1.0, float(np.abs(dw64).max()))
    np.testing.assert_allclose(_n(wt.grad) / s, dw64 / s, rtol=1e-5, atol=2e-5)


def test_pointwise_gemm_bias_elu_epilogue(dev):
    R, Cin, Cout = 1500, 96, 72
    rng = np.random.RandomState(1)
    x, w, b = rng.randn(R, Cin).astype(np.float32), rng.randn(Cin, Cout).astype(np.float32), rng.randn(Cout).astype(np.float32)
    xt, wt, bt = _t(x, dev), _t(w, dev), _t(b, dev)
    y = torch.empty((R, Cout), device=dev)
    _lib.check(_lib.lib().sph3d_pointwise_gemm(R, Cin, Cout, _lib.ptr(xt), _lib.ptr(wt), _lib.ptr(bt), 1, 0, _lib.ptr(y),
                                               _lib.stream_ptr()))
    ref = torch.nn.functional.elu(xt.double() @ wt.double() + bt.double())
    torch.testing.assert_close(y.double(), ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("R,C", [(1000, 64), (4096, 128), (777, 192), (20000, 512), (131072, 128)])
def test_fused_elu_batch_norm(dev, R, C):
    """sph3d::elu_bn == torch elu -> batch_norm (training and inference), forward, backward, moving statistics."""
    from sph3d_gcn_amd import tf_norm
    g = torch.Generator(device="cpu").manual_seed(R + C)
    y = (torch.randn(R, C, generator=g) * 2 - 0.3).to(dev)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    beta = torch.randn(C, generator=g).to(dev)
    dout = torch.randn(R, C, generator=g).to(dev)
    for training in (True, False):
        mm, mv = torch.zeros(C, device=dev) + 0.1, torch.ones(C, device=dev) * 0.7
        mm2, mv2 = mm.clone(), mv.clone()
        y1 = y.clone().requires_grad_(True); g1 = gamma.clone().requires_grad_(True); b1 = beta.clone().requires_grad_(True)
        y2 = y.clone().requires_grad_(True); g2 = gamma.clone().requires_grad_(True); b2 = beta.clone().requires_grad_(True)
        out = tf_norm.elu_batch_norm(y1, g1, b1, mm, mv, training)
        ref = torch.nn.functional.batch_norm(torch.nn.functional.elu(y2), mm2, mv2, g2, b2, training=training,
                                             momentum=0.01, eps=1e-3)
        torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-5)
        out.backward(dout); ref.backward(dout)
        torch.testing.assert_close(y1.grad, y2.grad, rtol=1e-4, atol=2e-5)
        torch.testing.assert_close(g1.grad, g2.grad, rtol=1e-4, atol=1e-4 * max(1.0, float(g2.grad.abs().max())))
        torch.testing.assert_close(b1.grad, b2.grad, rtol=1e-4, atol=1e-4 * max(1.0, float(b2.grad.abs().max())))
        torch.testing.assert_close(mm, mm2, rtol=1e-5, atol=1e-6)
        if training:      # tf.layers (non-fused path) keeps the BIASED batch variance in moving_variance; torch the unbiased one
            mv2 = 0.99 * 0.7 + 0.01 * torch.nn.functional.elu(y).double().var(0, unbiased=False).float()
        torch.testing.assert_close(mv, mv2, rtol=1e-5, atol=1e-6)


def test_modelnet_shapes_global_conv_and_odd_channels(dev):
    """BASELINE config #2 shapes (models/SPH3D_modelnet.py:47-93): 10000-point clouds (not a multiple of 64 / 1024),
    odd channel counts (35, 67, 131 -> generic kernels), and the global conv: K = 156 > 64 neighbours, one query
    (the cloud centroid), kernel [8,2,1] -> F = 17 bins."""
    B, N = 2, 10000
    xyz = synth.modelnet_batch(5, B, N)
    xt = _t(xyz, dev)
    idx_o, cnt_o, dst_o = oracle.build_sphere_neighbor(xyz, xyz, 0.1, None, 64)
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(xt, xt, 0.1, None, 64)
    np.testing.assert_array_equal(_n(idx), idx_o)
    np.testing.assert_array_equal(_n(cnt), cnt_o)
    filt_o = oracle.spherical_kernel(xyz, xyz, idx_o, cnt_o, dst_o, 0.1, [8, 2, 2])
    filt = tf_buildkernel.spherical_kernel(xt, xt, idx, cnt, dst, 0.1, [8, 2, 2])
    np.testing.assert_array_equal(_n(filt), filt_o)
    fps_o = oracle.farthest_point_sample(2500, xyz)
    np.testing.assert_array_equal(_n(tf_sample.farthest_point_sample(2500, xt)), fps_o)
    rng = np.random.RandomState(3)
    x = rng.randn(B, N, 35).astype(np.float32)
    w = rng.randn(33, 35, 2).astype(np.float32)
    go = rng.randn(B, N, 70).astype(np.float32)
    xg = _t(x, dev).requires_grad_(True)
    wg = _t(w, dev).requires_grad_(True)
    out = tf_conv3d.depthwise_conv3d(xg, wg, idx, cnt, filt)
    np.testing.assert_allclose(_n(out), oracle.depthwise_conv3d(x, w, idx_o, cnt_o, filt_o), **TOL)
    out.backward(_t(go, dev))
    gi_o, gf_o = oracle.depthwise_conv3d_grad(x, w, go, idx_o, cnt_o, filt_o)
    np.testing.assert_allclose(_n(xg.grad), gi_o, **TOL)
    s = max(1.0, float(np.abs(gf_o).max()))
    np.testing.assert_allclose(_n(wg.grad) / s, gf_o / s, **TOL)
    # global graph over the last level (156 points), query = centroid of the input cloud, radius 100, K = 156
    last = xyz[np.arange(B)[:, None], fps_o[:, :156]]
    centroid = xyz.mean(axis=1, keepdims=True).astype(np.float32)
    gi_o_, gc_o, gd_o = oracle.build_sphere_neighbor(last, centroid, 100.0, None, 156)
    gidx, gcnt, gdst = tf_nnquery.build_sphere_neighbor(_t(last, dev), _t(centroid, dev), 100.0, None, 156)
    np.testing.assert_array_equal(_n(gidx), gi_o_)
    assert (gc_o == 156).all() and np.array_equal(_n(gcnt), gc_o)
    gf_o_ = oracle.spherical_kernel(last, centroid, gi_o_, gc_o, gd_o, 100.0, [8, 2, 1])
    gfilt = tf_buildkernel.spherical_kernel(_t(last, dev), _t(centroid, dev), gidx, gcnt, gdst, 100.0, [8, 2, 1])
    np.testing.assert_array_equal(_n(gfilt), gf_o_)
    assert gf_o_.max() <= 16
    feat = rng.randn(B, 156, 128).astype(np.float32)
    w17 = rng.randn(17, 128, 2).astype(np.float32)
    gog = rng.randn(B, 1, 256).astype(np.float32)
    ft = _t(feat, dev).requires_grad_(True)
    wt = _t(w17, dev).requires_grad_(True)
    gout = tf_conv3d.depthwise_conv3d(ft, wt, gidx, gcnt, gfilt)
    np.testing.assert_allclose(_n(gout), oracle.depthwise_conv3d(feat, w17, gi_o_, gc_o, gf_o_), **TOL)
    gout.backward(_t(gog, dev))
    a, b = oracle.depthwise_conv3d_grad(feat, w17, gog, gi_o_, gc_o, gf_o_)
    np.testing.assert_allclose(_n(ft.grad), a, **TOL)
    np.testing.assert_allclose(_n(wt.grad), b, **TOL)


def test_scannet_stress_65536_points(dev):
    """BASELINE config #5 shape (one 65 536-point cloud): the multi-chunk LDS path of the neighbour search (N > 12 288)
    and the workspace fallback of FPS (n > 24 576).  The oracle checks a subset of queries / a short FPS prefix;
    the full outputs are checked by properties."""
    N, K, r = 65536, 64, 0.1
    xyz = synth.s3dis_batch(77, 1, N, extent=(6.0, 6.0, 3.0))[0]
    xt = _t(xyz, dev)
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(xt, xt, r, None, K)
    idx_n, cnt_n = _n(idx), _n(cnt)
    assert cnt_n.min() >= 1 and cnt_n.max() <= K
    valid = np.arange(K)[None, None, :] < cnt_n[:, :, None]
    assert (np.diff(idx_n, axis=2)[valid[:, :, 1:]] > 0).all() and (idx_n[~valid] == 0).all()
    # chains: query j is searched with radius r (+) 0.05 * (j // 1024); check the first 1024 queries (chain position 0)
    # and the last 1024 (position 63, radius 3.25) against the oracle run on those queries as db-vs-query
    i_o, c_o, d_o = oracle.build_sphere_neighbor(xyz, xyz[:, :1024], r, None, K)
    np.testing.assert_array_equal(idx_n[:, :1024], i_o)
    np.testing.assert_array_equal(cnt_n[:, :1024], c_o)
    r63 = np.float32(r)
    for _ in range(63):
        r63 = np.float32(np.float64(r63) + 0.05)          # the chain's radius after 63 single-pass queries (3.25)
    i_l, c_l, d_l = oracle.build_sphere_neighbor(xyz, xyz[:, -1024:], float(r63), None, K)
    np.testing.assert_array_equal(idx_n[:, -1024:], i_l)
    np.testing.assert_array_equal(cnt_n[:, -1024:], c_l)
    m = 300
    fps = _n(tf_sample.farthest_point_sample(m, xt))
    np.testing.assert_array_equal(fps, oracle.farthest_point_sample(m, xyz))


def _run_net(kind, device):
    """One forward + backward of a reduced SPH3D plan with seed-7 weights; on CPU through the oracle ops, on the GPU
    through the HIP ops.  Returns (logits, loss, {name: grad})."""
    from oracle import torch_ops
    from sph3d_gcn_amd.harness import modelnet_net, s3dis_net, shapenet_net
    import contextlib
    cpu = device.type == "cpu"
    ctx = torch_ops.patched_util() if cpu else contextlib.nullcontext()
    with ctx:
        if kind == "s3dis":
            cfg = s3dis_net.small_config(1024)
            xyz, label, inner = synth.s3dis_batch(0, 2, 1024, extent=(1.0, 1.0, 1.5))
            model = s3dis_net.SPH3DS3DIS(cfg, device=device)
            pred, _ = model(torch.from_numpy(xyz).to(device), is_training=True)
            loss = model.loss(pred, torch.from_numpy(label).to(device), torch.from_numpy(inner).to(device))
        elif kind == "modelnet":
            cfg = modelnet_net.small_config(1024)
            pts = torch.from_numpy(synth.modelnet_batch(0, 2, 1024)).to(device)
            model = modelnet_net.SPH3DModelNet(cfg, device=device)
            pred, _ = model(pts, is_training=True, dropout_generator=torch.Generator().manual_seed(5))
            loss = model.loss(pred, torch.tensor([3, 17], device=device))
        else:
            cfg = shapenet_net.small_config(512)
            pts = torch.from_numpy(synth.modelnet_batch(20, 2, 512)).to(device)
            label = torch.randint(0, 3, (2, 512), generator=torch.Generator().manual_seed(1)).to(device)
            model = shapenet_net.SPH3DShapeNet(3, cfg, device=device)
            pred, _ = model(pts, is_training=True)
            loss = model.loss(pred, label)
        loss.backward()
    grads = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters()}
    return pred.detach().cpu().numpy(), float(loss.detach()), grads


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["s3dis", "modelnet", "shapenet"])
def test_model_graphs_end_to_end_vs_oracle(dev, kind):
    """the three model call patterns (models/SPH3D_s3dis.py, SPH3D_modelnet.py, SPH3D_shapenet.py), reduced
    plans, same seed-7 weights: HIP ops on the GPU against the oracle ops on the CPU — logits, loss and every parameter
    gradient.  Tolerance 2e-3 of the tensor's scale: ~20 fp32 layers with batch normalisation between the two ends."""
    pred_o, loss_o, grads_o = _run_net(kind, torch.device("cpu"))
    pred, loss, grads = _run_net(kind, dev)
    assert pred.shape == pred_o.shape
    s = max(1.0, float(np.abs(pred_o).max()))
    np.testing.assert_allclose(pred / s, pred_o / s, rtol=0, atol=2e-3)
    assert abs(loss - loss_o) <= 2e-3 * max(1.0, abs(loss_o))
    assert grads.keys() == grads_o.keys()
    for n in grads_o:
        s = max(1e-3, float(np.abs(grads_o[n]).max()))
        np.testing.assert_allclose(grads[n] / s, grads_o[n] / s, rtol=0, atol=5e-3, err_msg=n)


@pytest.mark.gpu
def test_gather_nd_kernel_matches_indexing(dev):
    """sph3gcn_util.gather_nd on the device (sph3d_gather_nd) = tf.gather_nd semantics for (cloud, point) pairs: rows of
    coordinates, neighbour lists and counts, bit for bit against torch indexing; the [B,S,2] pairs come from build_graph"""
    from sph3d_gcn_amd import sph3gcn_util as s3g_util
    g = torch.Generator().manual_seed(3)
    B, N, S = 5, 700, 129
    pairs = torch.stack([torch.arange(B).view(B, 1).expand(B, S), torch.randint(0, N, (B, S), generator=g)], dim=-1).int().to(dev)
    for shape, dt in (((B, N, 3), torch.float32), ((B, N, 64), torch.int32), ((B, N), torch.int32), ((B, N, 4, 5), torch.float32)):
        src = (torch.rand(shape, generator=g) * 1000).to(dt).to(dev)
        got = s3g_util.gather_nd(src, pairs)
        want = src[pairs[..., 0].long(), pairs[..., 1].long()]
        assert got.dtype == src.dtype and got.shape == want.shape
        assert torch.equal(got, want)
    # a permuted batch column (pairs need not be sorted by cloud)
    perm = pairs.flip(0).contiguous()
    src = torch.rand((B, N, 7), generator=g).to(dev)
    assert torch.equal(s3g_util.gather_nd(src, perm), src[perm[..., 0].long(), perm[..., 1].long()])


@pytest.mark.gpu
def test_counted_inter_level_search_equals_separate_calls(dev):
    """tf_nnquery.build_sphere_neighbor_counted (search + in-edge counts in one kernel, transpose finished from them) =
    build_sphere_neighbor + sph3d_graph_transpose on the result: neighbour tensors and the transposed graph bit for bit"""
    from sph3d_gcn_amd import tf_nnquery, _tgraph
    g = torch.Generator().manual_seed(11)
    B, N, M, K = 3, 300, 1100, 24
    db = torch.rand((B, N, 3), generator=g).to(dev)
    qr = torch.rand((B, M, 3), generator=g).to(dev)
    i0, c0, d0 = tf_nnquery.build_sphere_neighbor(db, qr, 0.12, None, K)
    tg0 = _tgraph.transpose(i0, c0, N)
    t0 = [tg0[0].clone()] + [t.clone() for t in _tgraph.entries(tg0)]
    i1, c1, d1 = tf_nnquery.build_sphere_neighbor_counted(db, qr, 0.12, K)
    assert torch.equal(i0, i1) and torch.equal(c0, c1) and torch.equal(d0, d1)
    tg1 = _tgraph.transpose(i1, c1, N)          # cached by the counted call
    t1 = [tg1[0]] + list(_tgraph.entries(tg1))
    assert torch.equal(t0[0], t1[0])               # offsets
    # entries of one source point may be filled in any order: compare them as sorted (key, scale) pairs per segment
    off = t0[0].view(B, N + 1).cpu()
    k0, k1, s0, s1 = t0[1].cpu(), t1[1].cpu(), t0[2].cpu(), t1[2].cpu()
    for b in range(B):
        for n in range(0, N, 37):
            a, e = int(off[b, n]), int(off[b, n + 1])
            assert sorted(zip(k0[a:e].tolist(), s0[a:e].tolist())) == sorted(zip(k1[a:e].tolist(), s1[a:e].tolist()))


@pytest.mark.gpu
def test_balanced_gradient_order_is_a_permutation_and_changes_nothing(dev):
    """sph3d_graph_balanced_order: a permutation of every cloud, sorted by in-degree inside 2048-point windows (descending in
    even windows, ascending in odd ones); the convolution gradient with it = the gradient in index order (same per-source sums;
    the filter gradient only re-associates its partial sums)"""
    from sph3d_gcn_amd import tf_nnquery, tf_buildkernel, tf_conv3d, _tgraph, _lib
    g = torch.Generator().manual_seed(5)
    B, N, K, C = 2, 5000, 32, 64
    xyz = torch.rand((B, N, 3), generator=g).to(dev)
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, xyz, 0.08, None, K)
    filt = tf_buildkernel.spherical_kernel(xyz, xyz, idx, cnt, dst, 0.08, [8, 2, 2])
    x = torch.randn((B, N, C), generator=g).to(dev); w = torch.randn((33, C, 2), generator=g).to(dev)
    go = torch.randn((B, N, 2 * C), generator=g).to(dev)
    _tgraph.clear()
    old = _tgraph.BALANCE_MIN_POINTS
    try:
        _tgraph.BALANCE_MIN_POINTS = 1 << 30
        gi0, gf0 = tf_conv3d.depthwise_conv3d_grad(x, w, go, idx, cnt, filt)
        assert _tgraph.source_order(idx) is None
        _tgraph.clear()
        _tgraph.BALANCE_MIN_POINTS = 1024
        gi1, gf1 = tf_conv3d.depthwise_conv3d_grad(x, w, go, idx, cnt, filt)
    finally:
        _tgraph.BALANCE_MIN_POINTS = old
    order = _tgraph.source_order(idx)
    assert order is not None and order.shape == (B, N)
    o = order.cpu().long()
    off = _tgraph.transpose(idx, cnt, N, bin_index=filt, num_bins=33)[0].view(B, N * 33 + 1).cpu().long()
    for b in range(B):
        assert torch.equal(torch.sort(o[b]).values, torch.arange(N))
        deg = off[b, 33::33] - off[b, 0:-1:33]
        for wi, lo in enumerate(range(0, N, 2048)):
            win = o[b, lo:lo + 2048]
            assert int(win.min()) >= lo and int(win.max()) < min(lo + 2048, N)
            d = deg[win]
            assert bool((d[1:] >= d[:-1]).all()) if wi % 2 else bool((d[1:] <= d[:-1]).all())
    # (the two transposed graphs were built separately: entries inside a (source, bin) segment land in atomic-arrival order,
    # so even the per-source sums may re-associate)
    assert float((gi0 - gi1).abs().max()) <= 1e-5 * float(gi0.abs().max())
    assert float((gf0 - gf1).abs().max()) <= 1e-5 * float(gf0.abs().max())


@pytest.mark.parametrize("C,r", [(64, 2), (128, 2), (256, 2), (36, 1)])
def test_non_finite_inputs_stay_confined_to_their_neighbourhoods(dev, C, r):
    """DESIGN section 2 / VERDICT r3 weak #4: the forward kernels run whole batches of slots and multiply a row's padding
    slots (a real neighbour row) by an all-zero filter row, so an Inf feature can come out as NaN (0 * Inf) where the
    reference's per-slot loop yields +-Inf.  What is pinned here: (1) an output element is non-finite in the HIP result
    exactly where it is non-finite in the oracle's — points with no non-finite neighbour are untouched; (2) all finite
    elements still agree within the 1e-5 bar.  Same for max / avg pooling and mean interpolation."""
    B, N, K = 2, 400, 48
    rng = np.random.RandomState(C + r)
    db, q, idx, cnt, dst, filt = _graph("uniform", B, N, None, K, 0.2, seed=C)
    x = rng.randn(B, N, C).astype(np.float32)
    w = (rng.rand(33, C, r).astype(np.float32) + 0.5)                  # no zero weights: Inf * w stays Inf in the reference
    bad = [(0, 7, 3), (0, 130, C - 1), (1, 55, 0), (1, 399, 5)]        # (cloud, point, channel)
    for b, n, c in bad:
        x[b, n, c] = np.inf if (n & 1) else -np.inf
    out_o = oracle.depthwise_conv3d(x, w, idx, cnt, filt)
    out = _n(tf_conv3d.depthwise_conv3d(_t(x, dev), _t(w, dev), _t(idx, dev), _t(cnt, dev), _t(filt, dev)))
    fin_o, fin = np.isfinite(out_o), np.isfinite(out)
    assert (~fin_o).any()
    np.testing.assert_array_equal(fin, fin_o)
    np.testing.assert_allclose(out[fin], out_o[fin], **TOL)
    # the deviation itself: where the reference has +-Inf the HIP kernel may hold NaN, never a finite number
    assert np.isinf(out_o[~fin_o]).all()
    for name, hip_fn, ora_fn in (("max_pool3d", lambda a: tf_pool3d.max_pool3d(a, _t(idx, dev), _t(cnt, dev)),
                                  lambda a: oracle.max_pool3d(a, idx, cnt)[0]),
                                 ("avg_pool3d", lambda a: tf_pool3d.avg_pool3d(a, _t(idx, dev), _t(cnt, dev)),
                                  lambda a: oracle.avg_pool3d(a, idx, cnt)),
                                 ("mean_interpolate", lambda a: tf_unpool3d.mean_interpolate(a, _t(idx, dev), _t(cnt, dev)),
                                  lambda a: oracle.mean_interpolate(a, idx, cnt))):
        xp = x.copy()
        if name == "max_pool3d":
            xp[np.isinf(xp)] = np.inf                                  # -Inf never wins a maximum
        o_o = ora_fn(xp)
        o = hip_fn(_t(xp, dev))
        o = _n(o[0] if isinstance(o, (tuple, list)) else o)
        f_o, f = np.isfinite(o_o), np.isfinite(o)
        assert (~f_o).any(), name
        np.testing.assert_array_equal(f, f_o, err_msg=name)
        np.testing.assert_allclose(o[f], o_o[f], err_msg=name, **TOL)
