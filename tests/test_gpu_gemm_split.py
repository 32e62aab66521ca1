"""The pointwise products on the bf16 matrix pipe with three-way split operands (csrc/gemm.hip: gemm_split_mfma) against float64
and against the exact fp32-MFMA kernels: same error size, on every product shape of the S3DIS plan, for operands spanning many
binades, with the statistics epilogue, and for the one-hot (identity) operand that pins the lane -> element layout."""
import numpy as np
import pytest
import torch

from sph3d_gcn_amd import _lib, tf_gemm, tf_norm

pytestmark = pytest.mark.gpu

SHAPES = [(131072, 128, 128), (131072, 256, 128), (32768, 256, 256), (32768, 512, 256), (12288, 512, 256), (6144, 512, 512),
          (6144, 1024, 512), (2048, 1024, 512), (6144, 2048, 256), (12288, 1024, 256), (32768, 1024, 128), (4096, 64, 128),
          (2048, 144, 64), (1024, 48, 192),
          # ragged shapes (element-wise bounds): the ModelNet plan's 35 / 67 / 131-channel layers and odd sizes in every dimension
          (320000, 70, 64), (80000, 134, 128), (20000, 262, 256), (4096, 35, 67), (1000, 131, 262), (5000, 48, 200), (777, 100, 33),
          (4100, 64, 128), (4096, 72, 96)]


@pytest.fixture
def modes():
    l = _lib.lib()
    prev = l.sph3d_pointwise_gemm_mode(-1)
    yield l
    l.sph3d_pointwise_gemm_mode(prev)


def _products(x, w, dy):
    return (tf_gemm._pointwise_gemm_impl(x, w, False), tf_gemm._pointwise_gemm_impl(dy, w, True), tf_gemm._pointwise_gemm_tn_impl(x, dy))


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "R%d-%d-%d" % s)
def test_split_products_vs_float64_and_fp32_kernels(dev, modes, shape):
    R, Ci, Co = shape
    g = torch.Generator(device=dev).manual_seed(R + Ci)
    # operands over ~12 binades with both signs: the pieces' exponents differ from element to element
    x = torch.randn(R, Ci, device=dev, generator=g) * torch.exp2(torch.randint(-6, 6, (R, Ci), device=dev, generator=g).float())
    w = torch.randn(Ci, Co, device=dev, generator=g) / Ci ** 0.5
    dy = torch.randn(R, Co, device=dev, generator=g) * torch.exp2(torch.randint(-6, 6, (R, Co), device=dev, generator=g).float())
    assert modes.sph3d_pointwise_gemm_mode(1) in (0, 1)
    got = _products(x, w, dy)
    modes.sph3d_pointwise_gemm_mode(0)
    ref32 = _products(x, w, dy)
    xd, wd, dyd = x.double(), w.double(), dy.double()
    want = (xd @ wd, dyd @ wd.t(), xd.t() @ dyd)
    mags = (xd.abs() @ wd.abs(), dyd.abs() @ wd.abs().t(), xd.abs().t() @ dyd.abs())       # sum of the terms' magnitudes per element
    for name, a, b, c, mag in zip(("NN", "NT", "TN"), got, ref32, want, mags):
        e_split = float(((a.double() - c).abs() / mag).max())
        e_f32 = float(((b.double() - c).abs() / mag).max())
        # |error| <= 1e-5 of the element's term magnitudes (north_star's bound); and no worse than 4x the fp32 kernel's own error
        assert e_split <= 1e-5, (name, e_split)
        assert e_split <= max(4 * e_f32, 4e-7), (name, e_split, e_f32)


def test_split_kernel_operand_layout_identity(dev, modes):
    """A = one-hot rows (a permuted identity), asymmetric B: every output element is ONE product, so any lane / k mix-up of the
    fragment layout moves or loses values; exact equality (a single bf16-exact product per element)"""
    modes.sph3d_pointwise_gemm_mode(1)
    R, K, N = 4096, 256, 128
    perm = torch.randperm(K, device=dev)
    x = torch.zeros(R, K, device=dev)
    x[torch.arange(R, device=dev), perm[torch.arange(R, device=dev) % K]] = 1.0
    w = (torch.arange(K * N, device=dev, dtype=torch.float32).reshape(K, N) % 251) + 0.5        # exactly representable in 3 bf16 pieces
    y = tf_gemm._pointwise_gemm_impl(x, w, False)
    assert torch.equal(y, w[perm[torch.arange(R, device=dev) % K]])
    # transposed-weight product and the weight gradient with the same one-hot operand
    wt = w.t().contiguous()                                                                      # [N, K] stored k-contiguous
    y2 = tf_gemm._pointwise_gemm_impl(x, wt, True)
    assert torch.equal(y2, wt.t()[perm[torch.arange(R, device=dev) % K]])
    dy = (torch.arange(R * N, device=dev, dtype=torch.float32).reshape(R, N) % 127) - 63.0
    dw = tf_gemm._pointwise_gemm_tn_impl(x, dy)
    want = torch.zeros(K, N, device=dev, dtype=torch.float64).index_add_(0, perm[torch.arange(R, device=dev) % K], dy.double())
    assert torch.equal(dw.double(), want)                                                        # sums of 16 small integers: exact


def test_split_statistics_epilogue(dev, modes):
    modes.sph3d_pointwise_gemm_mode(1)
    R, Ci, Co = 32768, 256, 128
    x = torch.randn(R, Ci, device=dev)
    w = torch.randn(Ci, Co, device=dev) / 16
    bias = torch.randn(Co, device=dev)
    y, partial = tf_norm._gemm_bnstats_impl(x, w, bias)
    yd = x.double() @ w.double() + bias.double()
    z = torch.where(yd > 0, yd, torch.expm1(yd))
    np.testing.assert_allclose(y.cpu().numpy(), yd.cpu().numpy(), rtol=1e-5, atol=1e-5)
    p = partial.double().sum(0)
    np.testing.assert_allclose((p[0] / R).cpu().numpy(), (z.sum(0) / R).cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose((p[1] / R).cpu().numpy(), ((z * z).sum(0) / R).cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_split_non_finite_operands_never_give_finite_outputs(dev, modes):
    modes.sph3d_pointwise_gemm_mode(1)
    x = torch.randn(1024, 64, device=dev)
    x[5, 7] = float("inf")
    x[9, 3] = float("nan")
    w = torch.randn(64, 64, device=dev)
    y = tf_gemm._pointwise_gemm_impl(x, w, False)
    assert not torch.isfinite(y[5]).any() and not torch.isfinite(y[9]).any()
    assert torch.isfinite(y[torch.tensor([0, 1, 2, 3, 4, 6, 7, 8, 10], device=dev)]).all()


def test_split_products_scale_exactly_by_powers_of_two_at_full_size(dev, modes):
    """size-independent property at the bench's level-0 shape: x = h + m + l is cut by truncation, so 2^k x cuts into 2^k h, 2^k m,
    2^k l and every partial product and fp32 accumulation scales exactly — the products of scaled operands are the scaled products,
    bit for bit (NN, NT, TN incl. the split-K slab sum)"""
    modes.sph3d_pointwise_gemm_mode(1)
    R, Ci, Co = 131072, 256, 128
    g = torch.Generator(device=dev).manual_seed(17)
    x = torch.randn(R, Ci, device=dev, generator=g)
    w = torch.randn(Ci, Co, device=dev, generator=g) / 16
    dy = torch.randn(R, Co, device=dev, generator=g)
    y, dx, dw = _products(x, w, dy)
    y2, dx2, dw2 = _products(x * 8.0, w * 0.25, dy * 8.0)
    assert torch.equal(y2, y * 2.0)
    assert torch.equal(dx2, dx * 2.0)
    assert torch.equal(dw2, dw * 64.0)
