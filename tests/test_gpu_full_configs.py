"""The BASELINE.json configurations at FULL size on the HIP path: size-independent properties over the whole outputs plus
oracle checks on slices the CPU finishes in seconds.

  config 2  ModelNet40 cls, 32 x 10 000 points, full SPH3D_modelnet plan, forward + backward
  config 3  ShapeNet part-seg, 64 x 2048 points, full encoder-decoder plan, forward + backward
  config 4  S3DIS seg, 16 x 8192-point blocks per GPU (the bench workload), forward + backward
  config 5  ScanNet stress, one 65 536-point block: neighbour graph, bins, depthwise conv forward + gradients, max-pool,
            un-pooling at level 0 (reference "compat" semantics: the radius-growth chain saturates most rows at K)
"""
import numpy as np
import pytest
import torch

import oracle
from _errors import assert_per_element
from sph3d_gcn_amd import tf_nnquery, tf_buildkernel, tf_conv3d, tf_pool3d, tf_unpool3d, tf_sample
from sph3d_gcn_amd import sph3gcn_util as s3g_util
from sph3d_gcn_amd.harness import modelnet_net, shapenet_net, s3dis_net, synth

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)


def _n(t):
    return t.detach().cpu().numpy()


def _finite_grads(model):
    n = 0
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        assert torch.isfinite(p.grad).all(), name
        n += p.numel()
    return n


def _graph_properties(idx, cnt, K, N):
    idx_n, cnt_n = _n(idx), _n(cnt)
    assert cnt_n.min() >= 1 and cnt_n.max() <= K
    valid = np.arange(K)[None, None, :] < cnt_n[:, :, None]
    assert (np.diff(idx_n, axis=2)[valid[:, :, 1:]] > 0).all()          # ascending database index
    assert (idx_n[~valid] == 0).all() and idx_n.min() >= 0 and idx_n.max() < N
    return idx_n, cnt_n


def _conv_vs_oracle(dev, xyz_db, xyz_q, idx, cnt, filt, F, C, r, what, seed=0):
    """depthwise forward and both gradients of ONE cloud's graph (numpy: xyz [1, N, 3], idx / cnt / filt of that cloud) at
    the plan's channel count, against the oracle: the 1e-5 bar on the activations and the per-element bound of
    tests/_errors.py on activations and gradients"""
    rng = np.random.RandomState(seed + C * 3 + r)
    N, M = xyz_db.shape[1], idx.shape[1]
    x = rng.randn(1, N, C).astype(np.float32)
    w = rng.randn(F, C, r).astype(np.float32)
    go = rng.randn(1, M, C * r).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    xt, wt = t(x).requires_grad_(True), t(w).requires_grad_(True)
    out = tf_conv3d.depthwise_conv3d(xt, wt, t(idx), t(cnt), t(filt))
    out.backward(t(go))
    out_o = oracle.depthwise_conv3d(x, w, idx, cnt, filt)
    gi_o, gf_o = oracle.depthwise_conv3d_grad(x, w, go, idx, cnt, filt)
    np.testing.assert_allclose(_n(out), out_o, **TOL)
    ax, aw, ago = np.abs(x), np.abs(w), np.abs(go)
    mi, mf = oracle.depthwise_conv3d_grad(ax, aw, ago, idx, cnt, filt)
    assert_per_element(_n(out), out_o, oracle.depthwise_conv3d(ax, aw, idx, cnt, filt), what + " forward")
    assert_per_element(_n(xt.grad), gi_o, mi, what + " grad_input")
    assert_per_element(_n(wt.grad), gf_o, mf, what + " grad_filter")


def test_modelnet_full_config(dev):
    B, N = 32, 10000
    cfg = modelnet_net.modelnet_config(N)
    xyz = synth.modelnet_batch(0, B, N)
    pts = torch.from_numpy(xyz).to(dev)
    label = torch.randint(0, 40, (B,), generator=torch.Generator().manual_seed(0)).to(dev)
    model = modelnet_net.SPH3DModelNet(cfg, device=dev)
    pred, _ = model(pts, is_training=True, dropout_generator=torch.Generator().manual_seed(5))
    assert pred.shape == (B, 40) and torch.isfinite(pred).all()
    loss = model.loss(pred, label)
    loss.backward()
    assert torch.isfinite(loss) and _finite_grads(model) == 788396          # parameter count of the full plan
    # level-0 integers of two clouds against the oracle (neighbour graph, bins, first FPS picks)
    sl = slice(3, 5)
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(pts[sl], pts[sl], cfg.radius[0], None, cfg.nn_uplimit[0])
    io, co, do = oracle.build_sphere_neighbor(xyz[sl], xyz[sl], cfg.radius[0], None, cfg.nn_uplimit[0])
    np.testing.assert_array_equal(_n(idx), io)
    np.testing.assert_array_equal(_n(cnt), co)
    filt = tf_buildkernel.spherical_kernel(pts[sl], pts[sl], idx, cnt, dst, cfg.radius[0], cfg.kernel)
    np.testing.assert_array_equal(_n(filt), oracle.spherical_kernel(xyz[sl], xyz[sl], io, co, do, cfg.radius[0], cfg.kernel))
    fps = tf_sample.farthest_point_sample(200, pts[sl])
    np.testing.assert_array_equal(_n(fps), oracle.farthest_point_sample(200, xyz[sl]))
    # the plan's odd channel counts on ONE cloud of the full-size plan, forward + both gradients against the oracle
    # (models/SPH3D_modelnet.py:47-93): level 0 = 32 mlp + 3 raw channels, r = 2 (10 000 points, K = 64, 33 bins);
    # level 1 = 64 + 3, r = 1 (2500 points); level 2 = 128 + 3, r = 1 (625 points); the global convolution = one query
    # (the centroid) over the 156 remaining points, kernel [8, 2, 1] -> 17 bins, 131 channels, r = 2
    one = slice(3, 4)
    cur = modelnet_net.normalize_xyz(pts[one]).contiguous()
    levels = [(35, 2), (67, 1), (131, 1)]
    for l, (C, r) in enumerate(levels):
        cn = _n(cur)
        i_l, c_l, d_l = tf_nnquery.build_sphere_neighbor(cur, cur, cfg.radius[l], None, cfg.nn_uplimit[l])
        f_l = tf_buildkernel.spherical_kernel(cur, cur, i_l, c_l, d_l, cfg.radius[l], cfg.kernel)
        _conv_vs_oracle(dev, cn, cn, _n(i_l), _n(c_l), _n(f_l), 33, C, r, "modelnet level %d C=%d r=%d" % (l, C, r))
        pick = tf_sample.farthest_point_sample(cfg.num_sample[l], cur)
        cur = torch.gather(cur, 1, pick.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    assert cur.shape[1] == 156
    query = cur.mean(dim=1, keepdim=True)
    gi, gc, gd = s3g_util.build_global_graph(cur, query, 100.0)
    gf = tf_buildkernel.spherical_kernel(cur, query, gi, gc, gd, 100.0, [8, 2, 1])
    assert gi.shape == (1, 1, 156) and int(gc.min()) == 156
    _conv_vs_oracle(dev, _n(cur), _n(query), _n(gi), _n(gc), _n(gf), 17, 131, 2, "modelnet global conv K=156 F=17 C=131 r=2")


def test_shapenet_full_config(dev):
    B, N = 64, 2048
    cfg = shapenet_net.shapenet_config(N)
    xyz = synth.modelnet_batch(100, B, N)
    pts = torch.from_numpy(xyz).to(dev)
    label = torch.randint(0, 3, (B, N), generator=torch.Generator().manual_seed(1)).to(dev)
    model = shapenet_net.SPH3DShapeNet(3, cfg, device=dev)
    pred, _ = model(pts, is_training=True)
    assert pred.shape == (B, N, 3) and torch.isfinite(pred).all()
    loss = model.loss(pred, label)
    loss.backward()
    assert torch.isfinite(loss) and _finite_grads(model) > 3_000_000
    sl = slice(10, 12)
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(pts[sl], pts[sl], cfg.radius[0], None, cfg.nn_uplimit[0])
    io, co, do = oracle.build_sphere_neighbor(xyz[sl], xyz[sl], cfg.radius[0], None, cfg.nn_uplimit[0])
    np.testing.assert_array_equal(_n(idx), io)
    np.testing.assert_array_equal(_n(cnt), co)
    fps = tf_sample.farthest_point_sample(cfg.num_sample[0], pts[sl])
    np.testing.assert_array_equal(_n(fps), oracle.farthest_point_sample(cfg.num_sample[0], xyz[sl]))
    # one cloud of the full-size plan: level-0 depthwise activation and both gradients at the plan's channel counts
    # (64 -> 128 and 128 -> 128 layers: C = 64 and 128, r = 2; models/SPH3D_shapenet.py:56-66) against the oracle
    one = slice(10, 11)
    i0, c0, d0 = tf_nnquery.build_sphere_neighbor(pts[one], pts[one], cfg.radius[0], None, cfg.nn_uplimit[0])
    f0 = tf_buildkernel.spherical_kernel(pts[one], pts[one], i0, c0, d0, cfg.radius[0], cfg.kernel)
    for C in (64, 128):
        _conv_vs_oracle(dev, xyz[one], xyz[one], _n(i0), _n(c0), _n(f0), 33, C, 2, "shapenet level 0 C=%d r=2" % C)


def test_s3dis_bench_config_full_step(dev):
    """the bench workload: 16 x 8192-point blocks, full plan (3 935 680 parameters), one forward + backward; the level-0
    graph of the whole batch is checked by properties, two blocks of it against the oracle"""
    B, N = 16, 8192
    cfg = s3dis_net.s3dis_config(N)
    xyz, label, inner = synth.s3dis_batch(1000, B, N)
    pts = torch.from_numpy(xyz).to(dev)
    model = s3dis_net.SPH3DS3DIS(cfg, device=dev)
    pred, _ = model(pts, is_training=True)
    assert pred.shape == (B, N, 13) and torch.isfinite(pred).all()
    loss = model.loss(pred, torch.from_numpy(label).to(dev), torch.from_numpy(inner).to(dev))
    loss.backward()
    assert torch.isfinite(loss) and _finite_grads(model) == 3935680
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(pts, pts, 0.1, None, 64)
    idx_n, cnt_n = _graph_properties(idx, cnt, 64, N)
    # NOTE the chain: with B = 16 (<= 32) every cloud starts its chains afresh, so cloud b alone reproduces rows of the batch
    io, co, do = oracle.build_sphere_neighbor(xyz[5:6], xyz[5:6], 0.1, None, 64)
    np.testing.assert_array_equal(idx_n[5:6], io)
    np.testing.assert_array_equal(cnt_n[5:6], co)


def test_scannet_65536_level0_feature_ops(dev):
    """config 5, compat (reference) semantics: N = 65 536, K = 64, r = 0.1.  Rows 0..1023 of the level-0 graph are also
    produced by the oracle (chain position 0); conv / pool / un-pool outputs of those rows and the gradients they induce
    are compared with the oracle run on the same rows; everything else by properties."""
    N, K, r, C = 65536, 64, 0.1, 64
    xyz = synth.s3dis_batch(77, 1, N, extent=(6.0, 6.0, 3.0))[0]
    xt = torch.from_numpy(xyz).to(dev)
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(xt, xt, r, None, K)
    idx_n, cnt_n = _graph_properties(idx, cnt, K, N)
    assert (cnt_n == K).mean() > 0.5                                 # the growing radius saturates most rows
    filt = tf_buildkernel.spherical_kernel(xt, xt, idx, cnt, dst, r, [8, 2, 2])
    filt_n = _n(filt)
    assert filt_n.min() >= 0 and filt_n.max() <= 32
    S = 1024
    io, co, do = oracle.build_sphere_neighbor(xyz, xyz[:, :S], r, None, K)
    np.testing.assert_array_equal(idx_n[:, :S], io)
    fo = oracle.spherical_kernel(xyz, xyz[:, :S], io, co, do, r, [8, 2, 2])
    np.testing.assert_array_equal(filt_n[:, :S], fo)
    rng = np.random.RandomState(0)
    x = rng.randn(1, N, C).astype(np.float32)
    w = rng.randn(33, C, 2).astype(np.float32)
    xg = torch.from_numpy(x).to(dev).requires_grad_(True)
    wg = torch.from_numpy(w).to(dev).requires_grad_(True)
    out = tf_conv3d.depthwise_conv3d(xg, wg, idx, cnt, filt)
    assert out.shape == (1, N, 2 * C) and torch.isfinite(out).all()
    np.testing.assert_allclose(_n(out)[:, :S], oracle.depthwise_conv3d(x, w, io, co, fo), **TOL)
    # gradient of a loss that only looks at the first S output rows == oracle gradient over the sliced graph
    go = np.zeros((1, N, 2 * C), np.float32)
    go[:, :S] = rng.randn(1, S, 2 * C)
    out.backward(torch.from_numpy(go).to(dev))
    gi_o, gf_o = oracle.depthwise_conv3d_grad(x, w, go[:, :S], io, co, fo)
    np.testing.assert_allclose(_n(xg.grad), gi_o, **TOL)
    s = max(1.0, float(np.abs(gf_o).max()))
    np.testing.assert_allclose(_n(wg.grad) / s, gf_o / s, **TOL)
    # max-pool to the first 16 384 FPS picks is too long a chain for a test: pool onto the first S queries' rows instead
    xp = torch.from_numpy(x).to(dev).requires_grad_(True)
    pooled, arg = tf_pool3d.max_pool3d(xp, idx[:, :S].contiguous(), cnt[:, :S].contiguous())
    po, ao = oracle.max_pool3d(x, io, co)
    np.testing.assert_array_equal(_n(pooled), po)
    np.testing.assert_array_equal(_n(arg), ao)
    gp = rng.randn(1, S, C).astype(np.float32)
    pooled.backward(torch.from_numpy(gp).to(dev))
    np.testing.assert_allclose(_n(xp.grad), oracle.max_pool3d_grad(x, gp, ao), **TOL)
    # un-pooling: coarse set = the first S points, every fine point interpolates from its neighbours among them
    coarse = xyz[:, :S]
    ui, uc, ud = tf_nnquery.build_sphere_neighbor(torch.from_numpy(coarse).to(dev), xt, 0.4, None, 16)
    cf = rng.randn(1, S, C).astype(np.float32)
    cg = torch.from_numpy(cf).to(dev).requires_grad_(True)
    up = tf_unpool3d.mean_interpolate(cg, ui, uc)
    assert up.shape == (1, N, C) and torch.isfinite(up).all()
    uio, uco, udo = oracle.build_sphere_neighbor(coarse, xyz[:, :2048], 0.4, None, 16)
    np.testing.assert_array_equal(_n(ui)[:, :2048], uio)
    np.testing.assert_allclose(_n(up)[:, :2048], oracle.mean_interpolate(cf, uio, uco), **TOL)
    gu = np.zeros((1, N, C), np.float32)
    gu[:, :2048] = rng.randn(1, 2048, C)
    up.backward(torch.from_numpy(gu).to(dev))
    np.testing.assert_allclose(_n(cg.grad), oracle.mean_interpolate_grad(cf, gu[:, :2048], uio, uco), **TOL)


def test_scannet_65536_fixed_radius_mode(dev):
    """config 5 in the labelled FIXED-radius mode (not reference semantics: every query searched at the nominal radius):
    bit-exact against the oracle's fixed-radius restatement on two slices, far fewer saturated rows than compat mode, and
    the level-0 conv / bins run on that graph."""
    N, K, r = 65536, 64, 0.1
    xyz = synth.s3dis_batch(77, 1, N, extent=(6.0, 6.0, 3.0))[0]
    xt = torch.from_numpy(xyz).to(dev)
    tf_nnquery.set_radius_mode("fixed")
    try:
        idx, cnt, dst = tf_nnquery.build_sphere_neighbor(xt, xt, r, None, K)
    finally:
        tf_nnquery.set_radius_mode("compat")
    idx_n, cnt_n = _graph_properties(idx, cnt, K, N)
    ic, cc, _dc = tf_nnquery.build_sphere_neighbor(xt, xt, r, None, K)
    assert (cnt_n == K).mean() < 0.5 * float((cc == K).float().mean())          # the chain no longer inflates the radius
    assert torch.equal(idx[:, :1024], ic[:, :1024])                              # chain position 0: same search
    for lo in (0, 40000):
        io, co, do = oracle.build_sphere_neighbor(xyz, xyz[:, lo:lo + 1024], r, None, K, fixed=True)
        np.testing.assert_array_equal(idx_n[:, lo:lo + 1024], io)
        np.testing.assert_array_equal(cnt_n[:, lo:lo + 1024], co)
        np.testing.assert_array_equal(_n(dst)[:, lo:lo + 1024].view(np.int32), do.view(np.int32))
    filt = tf_buildkernel.spherical_kernel(xt, xt, idx, cnt, dst, r, [8, 2, 2])
    x = torch.randn(1, N, 64, device=dev)
    w = torch.randn(33, 64, 2, device=dev)
    out = tf_conv3d.depthwise_conv3d(x, w, idx, cnt, filt)
    io, co, do = oracle.build_sphere_neighbor(xyz, xyz[:, :512], r, None, K, fixed=True)
    fo = oracle.spherical_kernel(xyz, xyz[:, :512], io, co, do, r, [8, 2, 2])
    np.testing.assert_allclose(_n(out)[:, :512], oracle.depthwise_conv3d(_n(x), _n(w), io, co, fo), **TOL)


@pytest.mark.gpu
def test_fixed_radius_mode_runs_the_model_graph_through_the_plain_kernels():
    """in the labelled non-reference radius mode the fused graph kernels (reference semantics only) must not be chosen: the
    model graph falls back to the separate search / binning ops and still steps"""
    import torch
    from sph3d_gcn_amd import tf_nnquery
    from sph3d_gcn_amd.harness import s3dis_net, synth
    dev = torch.device("cuda:0")
    cfg = s3dis_net.small_config(1024)
    xyz, label, inner = synth.s3dis_batch(3, 2, 1024, extent=(1.0, 1.0, 1.5))
    pts = torch.from_numpy(xyz).to(dev); label = torch.from_numpy(label).to(dev); inner = torch.from_numpy(inner).to(dev)
    tf_nnquery.set_radius_mode("fixed")
    try:
        model = s3dis_net.SPH3DS3DIS(cfg, device=dev)
        pred, _ = model(pts, True)
        loss = model.loss(pred, label, inner)
        loss.backward()
        assert torch.isfinite(loss).item()
    finally:
        tf_nnquery.set_radius_mode("compat")


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    """the driver depends on `python bench.py` printing ONE JSON line on rank 0: run it (3 steps) as a subprocess and check the
    contract fields, the roofline object and the conv_gather line"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    assert d["unit"] == "blocks/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and 0 < rf["frac"] < 1
    assert 0 < d["conv_gather"]["frac"] < 1
