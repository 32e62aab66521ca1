"""Known-answer tests pinning the CPU oracle (values probed from the reference kernel bodies).

No GPU needed.  These are the hand-derivable cases: direction bins, radius growth, the (i, j) radius chains,
the K cap, the strict-inequality edge, the FPS tie-break, plus constructed identities for conv/pool/unpool.
"""
import numpy as np
import pytest

import oracle


def _origin(m=1, b=1):
    return np.zeros((b, m, 3), np.float32)


def test_kat_dirs_neighbors_and_bins():
    db = np.array([[(0, 0, 0), (.04, 0, 0), (-.04, 0, 0), (0, .04, 0), (0, -.04, 0), (0, 0, .04), (0, 0, -.04),
                    (.0005, 0, 0), (.2, 0, 0)]], np.float32)
    idx, cnt, dst = oracle.build_sphere_neighbor(db, _origin(), 0.1, None, 8)
    assert cnt.tolist() == [[8]]
    assert idx[0, 0].tolist() == [0, 1, 2, 3, 4, 5, 6, 7]
    np.testing.assert_allclose(dst[0, 0], [0, .2, .2, .2, .2, .2, .2, .02236068], rtol=0, atol=1e-7)
    assert oracle.spherical_kernel(db, _origin(), idx, cnt, dst, 0.1, [8, 2, 2])[0, 0].tolist() == \
        [0, 29, 25, 31, 26, 29, 21, 13]
    assert oracle.spherical_kernel(db, _origin(), idx, cnt, dst, 0.1, [8, 2, 3])[0, 0].tolist() == \
        [0, 45, 41, 47, 42, 45, 37, 13]


def test_kat_grow():
    db = np.array([[(.23, 0, 0), (5, 5, 5)]], np.float32)
    idx, cnt, dst = oracle.build_sphere_neighbor(db, _origin(), 0.1, None, 4)
    assert cnt[0, 0] == 1 and idx[0, 0, 0] == 0
    assert abs(float(dst[0, 0, 0]) - 0.4795831) < 1e-7
    assert idx[0, 0, 1:].tolist() == [0, 0, 0] and dst[0, 0, 1:].tolist() == [0, 0, 0]


def test_kat_chain_j():
    db = np.array([[(.12, 0, 0), (0, 0, 0)]], np.float32)
    idx, cnt, _ = oracle.build_sphere_neighbor(db, _origin(1025), 0.1, None, 4)
    assert (cnt[0, :1024] == 1).all()
    assert cnt[0, 1024] == 2 and idx[0, 1024].tolist() == [0, 1, 0, 0]


def test_kat_chain_i():
    db = np.tile(np.array([[(.12, 0, 0), (0, 0, 0)]], np.float32), (33, 1, 1))
    _, cnt, _ = oracle.build_sphere_neighbor(db, _origin(1, 33), 0.1, None, 4)
    assert cnt[:32, 0].tolist() == [1] * 32 and cnt[32, 0] == 2


def test_radius_sequence():
    r = np.float32(0.1)
    seq = []
    for _ in range(10):
        seq.append(float(r))
        r = np.float32(np.float64(r) + 0.05)
    expect = [0.100000001, 0.150000006, 0.200000003, 0.25, 0.300000012, 0.350000024, 0.400000036, 0.450000048,
              0.50000006, 0.550000072]
    np.testing.assert_allclose(seq, expect, rtol=0, atol=5e-9)


def test_kat_cap_and_edge():
    db = np.zeros((1, 20, 3), np.float32)
    db[0, :, 0] = 0.001 * np.arange(20)
    idx, cnt, _ = oracle.build_sphere_neighbor(db, _origin(), 0.1, None, 4)
    assert cnt[0, 0] == 4 and idx[0, 0].tolist() == [0, 1, 2, 3]
    db = np.zeros((1, 3, 3), np.float32)
    db[0, :, 0] = [0.1, 0.0999995, 0.09]
    idx, cnt, _ = oracle.build_sphere_neighbor(db, _origin(), 0.1, None, 4)
    assert cnt[0, 0] == 1 and idx[0, 0, 0] == 2


def test_single_point():
    db = np.array([[(1.5, -2, 3)]], np.float32)
    idx, cnt, dst = oracle.build_sphere_neighbor(db, db, 0.1, None, 3)
    assert cnt[0, 0] == 1 and idx[0, 0].tolist() == [0, 0, 0] and dst[0, 0, 0] == 0
    assert oracle.spherical_kernel(db, db, idx, cnt, dst, 0.1, [8, 2, 2])[0, 0].tolist() == [0, 0, 0]


def test_fps_tie_break():
    rng = np.random.RandomState(0)
    pts = (rng.rand(1, 2048, 3).astype(np.float32) - 0.5) * 0.15
    pts[0, 0] = 0
    pts[0, 1030] = (1, 0, 0)
    pts[0, 7] = (0, 1, 0)
    assert oracle.farthest_point_sample(3, pts)[0].tolist() == [0, 1030, 7]


def test_fps_equals_naive_on_distinct_points():
    rng = np.random.RandomState(3)
    pts = rng.rand(2, 300, 3).astype(np.float32)
    got = oracle.farthest_point_sample(40, pts)
    for b in range(2):
        d = np.full(300, 1e38, np.float32)
        cur, out = 0, [0]
        for _ in range(39):
            diff = pts[b] - pts[b, cur]
            dd = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
            d = np.minimum(d, dd.astype(np.float32))
            cur = int(np.argmax(d))
            out.append(cur)
        assert got[b].tolist() == out
        assert len(set(out)) == 40


def test_fps_small_n_idle_threads():
    pts = np.array([[(0, 0, 0), (1, 0, 0), (0, 2, 0)]], np.float32)
    assert oracle.farthest_point_sample(3, pts)[0].tolist() == [0, 2, 1]


def _graph(rng, B, N, M, K):
    db = rng.rand(B, N, 3).astype(np.float32)
    q = db[:, :M].copy()
    idx, cnt, dst = oracle.build_sphere_neighbor(db, q, 0.25, None, K)
    filt = oracle.spherical_kernel(db, q, idx, cnt, dst, 0.25, [8, 2, 2])
    return db, q, idx, cnt, dst, filt


def test_conv_mean_identity():
    rng = np.random.RandomState(1)
    db, q, idx, cnt, dst, filt = _graph(rng, 2, 200, 100, 16)
    x = rng.randn(2, 200, 8).astype(np.float32)
    w = np.ones((33, 8, 1), np.float32)
    out = oracle.depthwise_conv3d(x, w, idx, cnt, np.zeros_like(filt))
    for b in range(2):
        for m in range(0, 100, 17):
            ref = x[b, idx[b, m, :cnt[b, m]]].astype(np.float64).mean(0)
            np.testing.assert_allclose(out[b, m], ref, rtol=1e-5, atol=1e-6)


def test_conv_matches_float64_and_gradcheck():
    rng = np.random.RandomState(2)
    B, N, M, C, r, K, F = 2, 60, 30, 5, 2, 8, 33
    db, q, idx, cnt, dst, filt = _graph(rng, B, N, M, K)
    x = rng.randn(B, N, C).astype(np.float32)
    w = rng.randn(F, C, r).astype(np.float32)
    go = rng.randn(B, M, C * r).astype(np.float32)
    out = oracle.depthwise_conv3d(x, w, idx, cnt, filt)

    def f64(x64, w64):
        o = np.zeros((B, M, C * r))
        for b in range(B):
            for m in range(M):
                for k in range(cnt[b, m]):
                    o[b, m] += (x64[b, idx[b, m, k]][:, None] * w64[filt[b, m, k]]).reshape(-1) / cnt[b, m]
        return o
    np.testing.assert_allclose(out, f64(x.astype(np.float64), w.astype(np.float64)), rtol=1e-5, atol=1e-6)
    gi, gf = oracle.depthwise_conv3d_grad(x, w, go, idx, cnt, filt)
    # analytic gradient in float64 by linearity: d<go,out>/dx and /dw
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    gi_ref = np.zeros_like(x64)
    gf_ref = np.zeros_like(w64)
    for b in range(B):
        for m in range(M):
            g = go[b, m].astype(np.float64).reshape(C, r) / cnt[b, m]
            for k in range(cnt[b, m]):
                n, f = idx[b, m, k], filt[b, m, k]
                gi_ref[b, n] += (g * w64[f]).sum(1)
                gf_ref[f] += g * x64[b, n][:, None]
    np.testing.assert_allclose(gi, gi_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gf, gf_ref, rtol=1e-5, atol=1e-5)
    # central differences on one coordinate of each
    eps = 1e-3
    for (b, n, c) in [(0, int(idx[0, 0, 0]), 1), (1, int(idx[1, 5, 0]), 4)]:
        xp, xm = x64.copy(), x64.copy()
        xp[b, n, c] += eps
        xm[b, n, c] -= eps
        fd = ((f64(xp, w64) - f64(xm, w64)) * go).sum() / (2 * eps)
        assert abs(fd - gi_ref[b, n, c]) < 1e-6 * max(1, abs(fd))


def test_pool_unpool_identities_and_grads():
    rng = np.random.RandomState(4)
    B, N, M, C, K = 2, 80, 40, 6, 8
    db, q, idx, cnt, dst, filt = _graph(rng, B, N, M, K)
    x = rng.randn(B, N, C).astype(np.float32)
    go = rng.randn(B, M, C).astype(np.float32)
    out, mi = oracle.max_pool3d(x, idx, cnt)
    avg = oracle.avg_pool3d(x, idx, cnt)
    for b in range(B):
        for m in range(M):
            rows = x[b, idx[b, m, :cnt[b, m]]]
            np.testing.assert_array_equal(out[b, m], rows.max(0))
            np.testing.assert_array_equal(mi[b, m], idx[b, m, :cnt[b, m]][rows.argmax(0)])   # first max wins
            np.testing.assert_allclose(avg[b, m], rows.astype(np.float64).mean(0), rtol=1e-5, atol=1e-6)
    g = oracle.max_pool3d_grad(x, go, mi)
    assert abs(g.sum() - go.sum()) < 1e-3
    ga = oracle.avg_pool3d_grad(x, go, idx, cnt)
    assert abs(ga.sum() - go.sum()) < 1e-3
    # un-pooling: coarse features (M) interpolated onto N fine points
    idx2, cnt2, dst2 = oracle.build_sphere_neighbor(q, db, 0.3, None, K)     # db=coarse(M), query=fine(N)
    feat = rng.randn(B, M, C).astype(np.float32)
    mean = oracle.mean_interpolate(feat, idx2, cnt2)
    w = (dst2 + 1e-7) / (dst2.sum(-1, keepdims=True) + 1e-7)
    wi = oracle.weighted_interpolate(feat, w, idx2, cnt2)
    for b in range(B):
        for n in range(0, N, 7):
            rows = feat[b, idx2[b, n, :cnt2[b, n]]].astype(np.float64)
            np.testing.assert_allclose(mean[b, n], rows.mean(0), rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(wi[b, n], (rows * w[b, n, :cnt2[b, n], None]).sum(0), rtol=1e-5, atol=1e-6)
    go2 = rng.randn(B, N, C).astype(np.float32)
    gm = oracle.mean_interpolate_grad(feat, go2, idx2, cnt2)
    assert abs(gm.sum() - go2.sum()) < 1e-3
    gw = oracle.weighted_interpolate_grad(feat, go2, w, idx2, cnt2)
    ref = np.zeros_like(feat, dtype=np.float64)
    for b in range(B):
        for n in range(N):
            for k in range(cnt2[b, n]):
                ref[b, idx2[b, n, k]] += go2[b, n].astype(np.float64) * w[b, n, k]
    np.testing.assert_allclose(gw, ref, rtol=1e-5, atol=1e-5)


def test_max_pool_tie_first_neighbor():
    x = np.ones((1, 4, 2), np.float32)
    idx = np.array([[[3, 1, 2, 0]]], np.int32)
    cnt = np.array([[4]], np.int32)
    out, mi = oracle.max_pool3d(x, idx, cnt)
    assert mi[0, 0].tolist() == [3, 3]


def test_cube_neighbor():
    db = np.array([[(0, 0, 0), (0.04, 0.04, 0.04), (-0.04, 0, 0.049), (0.06, 0, 0), (0.01, -0.02, 0.03)]], np.float32)
    idx, cnt = oracle.build_cube_neighbor(db, _origin(), 0.1, None, 4, 3)
    assert cnt[0, 0] == 4
    assert idx[0, 0, :, 0].tolist() == [0, 1, 2, 4]
    # bins: ((d + L/2) / (L/3)) per axis -> x*9 + y*3 + z
    assert idx[0, 0, 0, 1] == 1 * 9 + 1 * 3 + 1
    assert idx[0, 0, 1, 1] == 2 * 9 + 2 * 3 + 2
    assert idx[0, 0, 2, 1] == 0 * 9 + 1 * 3 + 2


def test_properties_random():
    rng = np.random.RandomState(11)
    for (B, N, M, K, r) in [(1, 16, 16, 4, 0.3), (3, 257, 130, 8, 0.15), (33, 70, 70, 5, 0.2)]:
        db = rng.rand(B, N, 3).astype(np.float32)
        q = rng.rand(B, M, 3).astype(np.float32)
        idx, cnt, dst = oracle.build_sphere_neighbor(db, q, r, None, K)
        assert cnt.min() >= 1 and cnt.max() <= K
        for b in range(B):
            for m in range(M):
                c = cnt[b, m]
                assert (np.diff(idx[b, m, :c]) > 0).all()
                assert (idx[b, m, c:] == 0).all() and (dst[b, m, c:] == 0).all()
        filt = oracle.spherical_kernel(db, q, idx, cnt, dst, r, [8, 2, 2])
        assert filt.min() >= 0 and filt.max() < 33


def test_arg_validation():
    db = np.zeros((1, 4, 3), np.float32)
    with pytest.raises(ValueError):
        oracle.build_sphere_neighbor(db, db, -1.0, None, 4)
    idx, cnt, dst = oracle.build_sphere_neighbor(db, db, 0.1, None, 4)
    for bad in ([3, 2, 2], [8, 3, 2], [8, 2, 0], [2, 2, 2]):
        with pytest.raises(ValueError):
            oracle.spherical_kernel(db, db, idx, cnt, dst, 0.1, bad)
    with pytest.raises(ValueError):
        oracle.farthest_point_sample(0, db)


def test_fixed_radius_variant_is_an_independent_search_per_query():
    """oracle_build_sphere_neighbor_fixed (the labelled non-reference mode): every query = brute force at the nominal
    radius, growing by 0.05 only while it has no neighbour; identical to compat mode at chain position 0."""
    rng = np.random.RandomState(4)
    B, N, M, K, r = 2, 400, 1500, 8, 0.08
    db = rng.rand(B, N, 3).astype(np.float32)
    q = rng.rand(B, M, 3).astype(np.float32)
    idx, cnt, dst = oracle.build_sphere_neighbor(db, q, r, None, K, fixed=True)
    ic, cc, dc = oracle.build_sphere_neighbor(db, q, r, None, K)
    np.testing.assert_array_equal(idx[:, :1024], ic[:, :1024])
    assert cc[:, 1024:].mean() > cnt[:, 1024:].mean()
    for b in range(B):
        for m in range(0, M, 37):
            rad = np.float32(r)
            while True:
                d = np.sqrt(((db[b] - q[b, m]) ** 2).astype(np.float32).sum(1, dtype=np.float32)).astype(np.float32)
                hit = np.nonzero((d < rad) & (np.abs(d - rad).astype(np.float64) > 1e-6))[0]
                if len(hit):
                    break
                rad = np.float32(np.float64(rad) + 0.05)
            assert cnt[b, m] == min(len(hit), K)
            assert idx[b, m, :cnt[b, m]].tolist() == hit[:K].tolist()
