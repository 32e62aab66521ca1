"""The barrier-free separable layer (csrc/sepring.hip; SURVEY 8f.3, training half): one kernel = depthwise gather + pointwise
product + batch-norm statistics partials.  Checked against the oracle's depthwise convolution with a float64 tail
(utils/sph3gcn_util.py:134-161), against the layer built from the separate ops (forward, every gradient, moving statistics),
and for the claim / ring protocol's corner cases (fewer row blocks than workgroups, ragged row blocks, batches that are not a
multiple of eight clouds, empty rows)."""
import numpy as np
import pytest
import torch

import oracle
from sph3d_gcn_amd import _lib, tf_nnquery, tf_buildkernel, tf_conv3d, tf_norm
from sph3d_gcn_amd import sph3gcn_util as s3g_util
from sph3d_gcn_amd.harness import synth

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _n(t):
    return t.detach().cpu().numpy()


def _graph(kind, B, N, M, radius, K, kernel, dev, seed=7):
    xyz = {"s3dis": lambda: synth.s3dis_batch(seed, B, N)[0], "modelnet": lambda: synth.modelnet_batch(seed, B, N),
           "uniform": lambda: synth.uniform_cloud(seed, B, N, 1.0)}[kind]()
    xyz = _t(xyz, dev)
    q = xyz[:, :M].contiguous()
    idx, cnt, dst = tf_nnquery.build_sphere_neighbor(xyz, q, radius, None, K)
    filt = tf_buildkernel.spherical_kernel(xyz, q, idx, cnt, dst, radius, kernel)
    return idx, cnt, filt


# (kind, B, N, M, radius, K, kernel, C, r, Cout)
TRAIN_CASES = [
    ("s3dis", 2, 2048, 2048, 0.1, 64, [8, 2, 2], 64, 2, 128),       # S3DIS level 0, first layer (two wave groups)
    ("s3dis", 8, 1024, 1024, 0.15, 64, [8, 2, 2], 128, 2, 128),     # level 0, second layer; B % 8 == 0: XCD-affine ranges
    ("s3dis", 2, 2048, 2048, 0.2, 64, [8, 2, 2], 128, 2, 256),      # level 1, first layer: 16 column blocks, one wave group
    ("uniform", 3, 700, 333, 0.15, 32, [8, 2, 2], 128, 1, 64),      # M < N, ragged last row block, four wave groups
    ("uniform", 2, 500, 500, 0.15, 24, [4, 2, 1], 36, 2, 16),       # C*r = 72 (k padded to 128), 9 bins, 16 wave groups > ring depth
    ("modelnet", 1, 1500, 1500, 0.1, 48, [8, 2, 3], 8, 1, 32),      # 49 bins, narrow input, eight wave groups
    ("uniform", 9, 257, 257, 0.2, 64, [8, 2, 2], 32, 2, 64),        # 17 row blocks per cloud, 9 clouds
    ("uniform", 1, 64, 40, 0.3, 16, [8, 2, 2], 128, 2, 128),        # 3 row blocks in the whole launch: 253 workgroups without work
    ("s3dis", 16, 512, 512, 0.2, 64, [8, 2, 3], 128, 2, 128),       # 49 bins: a shorter ring beside the larger filter table
]


@pytest.mark.parametrize("with_bias", [False, True], ids=["nobias", "bias"])
@pytest.mark.parametrize("case", TRAIN_CASES, ids=lambda c: "%s-B%d-N%d-M%d-C%d-r%d-Co%d" % (c[0], c[1], c[2], c[3], c[7], c[8], c[9]))
def test_training_kernel_vs_oracle(dev, case, with_bias):
    kind, B, N, M, radius, K, kernel, C, r, Cout = case
    F = kernel[0] * kernel[1] * kernel[2] + 1
    idx, cnt, filt = _graph(kind, B, N, M, radius, K, kernel, dev)
    rng = np.random.RandomState(C * 7 + Cout)
    x = rng.randn(B, N, C).astype(np.float32)
    dwf = rng.randn(F, C, r).astype(np.float32)
    w = (rng.randn(C * r, Cout) / np.sqrt(C * r)).astype(np.float32)
    bias = rng.randn(Cout).astype(np.float32) if with_bias else None
    assert tf_conv3d.separable_train_supported(_t(x, dev), _t(dwf, dev), idx, Cout)
    dw, y, partial = tf_conv3d._separable_conv3d_train_impl(_t(x, dev), _t(dwf, dev), _t(w, dev), None if bias is None else _t(bias, dev),
                                                            idx, cnt, filt)
    torch.cuda.synchronize()
    assert _lib.lib().sph3d_separable_conv3d_ring_failures() == 0
    want_dw = oracle.depthwise_conv3d(x, dwf, _n(idx), _n(cnt), _n(filt))
    np.testing.assert_allclose(_n(dw), want_dw, rtol=1e-5, atol=1e-5)                 # north_star: 1e-5 on conv activations
    # ... and the separate op's (same gather; the lane mapping may differ per shape: summation order only)
    np.testing.assert_allclose(_n(dw), _n(tf_conv3d.depthwise_conv3d(_t(x, dev), _t(dwf, dev), idx, cnt, filt)), rtol=2e-6, atol=2e-6)
    want_y = want_dw.astype(np.float64).reshape(-1, C * r) @ w.astype(np.float64)
    if bias is not None:
        want_y = want_y + bias
    mag = max(1.0, float(np.abs(want_y).max()))
    np.testing.assert_allclose(_n(y).reshape(-1, Cout) / mag, want_y / mag, rtol=1e-5, atol=1e-5)
    z = np.where(want_y > 0, want_y, np.expm1(np.minimum(want_y, 0)))
    nblk = _lib.lib().sph3d_separable_conv3d_train_blocks(Cout)
    assert partial.shape == (nblk, 2, Cout)
    p = _n(partial).astype(np.float64).sum(0)
    rows = want_y.shape[0]
    np.testing.assert_allclose(p[0] / rows, z.sum(0) / rows, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(z).max()))
    np.testing.assert_allclose(p[1] / rows, (z * z).sum(0) / rows, rtol=1e-5, atol=1e-5 * max(1.0, (z * z).max()))


def test_training_kernel_rejects_uncovered_shapes(dev):
    idx, cnt, filt = _graph("uniform", 2, 300, 300, 0.3, 16, [8, 2, 2], dev)
    x = torch.randn(2, 300, 256, device=dev)
    dwf = torch.randn(33, 256, 2, device=dev)
    assert not tf_conv3d.separable_train_supported(x, dwf, idx, 128)                                    # C > 128
    assert not tf_conv3d.separable_train_supported(x[..., :64].contiguous(), dwf[:, :64].contiguous(), idx, 96)    # Cout not a power of two
    assert not tf_conv3d.separable_train_supported(x[..., :64].contiguous(), dwf[:, :64].contiguous(), idx, 512)
    with pytest.raises(_lib.Sph3dError):
        tf_conv3d._separable_conv3d_train_impl(x, dwf, torch.randn(512, 128, device=dev), None, idx, cnt, filt)


@pytest.mark.parametrize("with_bias", [False, True], ids=["nobias", "bias"])
@pytest.mark.parametrize("shape", [(64, 2, 128), (128, 2, 128), (128, 2, 256), (32, 1, 64)], ids=lambda s: "C%d-r%d-Co%d" % s)
def test_training_layer_fused_equals_layer_by_layer(dev, shape, with_bias):
    """s3g_util.separable_conv3d(is_training=True, with_bn=True): the one-kernel forward against the separate ops — same variables,
    output, moving statistics and every gradient (the backward pass IS the separate ops')."""
    C, r, Cout = shape
    B, N = 4, 1024
    idx, cnt, filt = _graph("s3dis", B, N, N, 0.15, 64, [8, 2, 2], dev)
    feat = torch.randn(B, N, C, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    gout = torch.randn(B, N, Cout, device=dev, generator=torch.Generator(device=dev).manual_seed(4))

    def run(fuse):
        old = s3g_util.FUSE_SEPARABLE_TRAINING
        s3g_util.FUSE_SEPARABLE_TRAINING = fuse
        try:
            store = s3g_util.VariableStore(device=dev, seed=11)
            x = feat.clone().requires_grad_(True)
            names = []
            _lib.timing_start()
            with s3g_util.variable_store(store):
                out = s3g_util.separable_conv3d(x, Cout, 33, r, 'c1', idx, cnt, filt, with_bn=True, with_bias=with_bias, is_training=True)
            names = [n for n, *_ in _lib.timing_stop()]
            (out * gout).sum().backward()
            grads = {k: p.grad.clone() for k, p in store.params.items()}
            stats = {k: b.clone() for k, b in store.named_buffers()}
            return out.detach(), x.grad.clone(), grads, stats, names
        finally:
            s3g_util.FUSE_SEPARABLE_TRAINING = old

    out_f, gx_f, g_f, st_f, names_f = run(True)
    out_u, gx_u, g_u, st_u, names_u = run(False)
    assert "sph3d_separable_conv3d_train" in names_f and "sph3d_depthwise_conv3d" not in names_f
    assert "sph3d_separable_conv3d_train" not in names_u and "sph3d_depthwise_conv3d" in names_u
    assert _lib.lib().sph3d_separable_conv3d_ring_failures() == 0
    assert set(g_f) == set(g_u) and set(st_f) == set(st_u)

    def close(a, b, tol=2e-5):
        mag = max(1.0, float(b.abs().max()))
        np.testing.assert_allclose(_n(a) / mag, _n(b) / mag, rtol=tol, atol=tol)

    close(out_f, out_u)
    close(gx_f, gx_u, 5e-5)
    for k in g_u:
        close(g_f[k], g_u[k], 5e-5)
    for k in st_u:
        close(st_f[k], st_u[k])


def test_training_layer_auto_mode_takes_large_layers_only(dev):
    B, N = 2, 1024
    idx, cnt, filt = _graph("s3dis", B, N, N, 0.15, 32, [8, 2, 2], dev)
    feat = torch.randn(B, N, 64, device=dev)
    assert s3g_util.FUSE_SEPARABLE_TRAINING is False                 # the default: measured, it does not pay (sph3gcn_util.py)
    for rows_min, want in ((1024, True), (1 << 20, False)):
        old, oldf = s3g_util._FUSED_TRAIN_MIN_ROWS, s3g_util.FUSE_SEPARABLE_TRAINING
        s3g_util._FUSED_TRAIN_MIN_ROWS, s3g_util.FUSE_SEPARABLE_TRAINING = rows_min, "auto"
        try:
            store = s3g_util.VariableStore(device=dev, seed=1)
            _lib.timing_start()
            with s3g_util.variable_store(store):
                s3g_util.separable_conv3d(feat, 128, 33, 2, 'c1', idx, cnt, filt, with_bn=True, is_training=True)
            names = [n for n, *_ in _lib.timing_stop()]
        finally:
            s3g_util._FUSED_TRAIN_MIN_ROWS, s3g_util.FUSE_SEPARABLE_TRAINING = old, oldf
        assert ("sph3d_separable_conv3d_train" in names) == want


def test_ring_kernel_many_launches_no_failure(dev):
    """the claim / ring protocol under repetition and with other work on a second stream (no hang, no give-up flag)"""
    B, N, C, r, Cout = 8, 2048, 128, 2, 128
    idx, cnt, filt = _graph("s3dis", B, N, N, 0.12, 64, [8, 2, 2], dev)
    x = torch.randn(B, N, C, device=dev)
    dwf = torch.randn(33, C, r, device=dev)
    w = torch.randn(C * r, Cout, device=dev) / 16
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=dev)
    ref = None
    for it in range(30):
        with torch.cuda.stream(side):
            a @ a
        dw, y, partial = tf_conv3d._separable_conv3d_train_impl(x, dwf, w, None, idx, cnt, filt)
        if ref is None:
            ref = (dw.clone(), y.clone())
        else:
            assert torch.equal(dw, ref[0]) and torch.equal(y, ref[1])          # deterministic: fixed k order per output
    torch.cuda.synchronize()
    assert _lib.lib().sph3d_separable_conv3d_ring_failures() == 0
