"""The in-kernel split-K exchange of the split products (csrc/gemm.hip: gemm_split_mfma<..., XS>): products whose 64 x 64 tile grid
is too small to fill the chip run `nsplit` workgroups per tile; split 0 adds the others' accumulators in split order.  Checked:
against float64 on the shapes that take the path (forward, statistics epilogue, weight gradient), bit-equal results from call to
call (fixed summation order, counters back at zero), interleaved shapes on one stream, and two streams at once."""
import numpy as np
import pytest
import torch

from sph3d_gcn_amd import _lib, tf_gemm, tf_norm

pytestmark = pytest.mark.gpu

XS_SHAPES = [(2048, 1024, 512), (6144, 2048, 256), (2048, 512, 512), (1024, 2048, 128), (4096, 1024, 256)]


def _ops(dev, R, Ci, Co, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(R, Ci, device=dev, generator=g)
    w = torch.randn(Ci, Co, device=dev, generator=g) / Ci ** 0.5
    dy = torch.randn(R, Co, device=dev, generator=g)
    return x, w, dy


@pytest.mark.parametrize("shape", XS_SHAPES, ids=lambda s: "R%d-%d-%d" % s)
def test_exchange_products_vs_float64_and_repeatable(dev, shape):
    R, Ci, Co = shape
    x, w, dy = _ops(dev, R, Ci, Co, R + Co)
    xd, wd, dyd = x.double(), w.double(), dy.double()
    for name, fn, want, mag in (("NN", lambda: tf_gemm._pointwise_gemm_impl(x, w, False), xd @ wd, xd.abs() @ wd.abs()),
                                ("NT", lambda: tf_gemm._pointwise_gemm_impl(dy, w, True), dyd @ wd.t(), dyd.abs() @ wd.abs().t()),
                                ("TN", lambda: tf_gemm._pointwise_gemm_tn_impl(x, dy), xd.t() @ dyd, xd.abs().t() @ dyd.abs())):
        first = fn()
        err = float(((first.double() - want).abs() / mag).max())
        assert err <= 4e-7, (name, err)
        for _ in range(5):
            assert torch.equal(fn(), first), name                  # fixed summation order; the counters were left at zero


def test_exchange_statistics_epilogue(dev):
    R, Ci, Co = 2048, 1024, 512
    x, w, _ = _ops(dev, R, Ci, Co, 3)
    bias = torch.randn(Co, device=dev)
    y, partial = tf_norm._gemm_bnstats_impl(x, w, bias)
    yd = x.double() @ w.double() + bias.double()
    z = torch.where(yd > 0, yd, torch.expm1(yd))
    np.testing.assert_allclose(y.cpu().numpy(), yd.cpu().numpy(), rtol=1e-5, atol=1e-5)
    p = partial.double().sum(0)
    np.testing.assert_allclose((p[0] / R).cpu().numpy(), (z.sum(0) / R).cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose((p[1] / R).cpu().numpy(), ((z * z).sum(0) / R).cpu().numpy(), rtol=1e-5, atol=1e-6)
    y2, partial2 = tf_norm._gemm_bnstats_impl(x, w, bias)
    assert torch.equal(y, y2) and torch.equal(partial, partial2)


def test_exchange_interleaved_shapes_and_streams(dev):
    """different tile counts / split counts one after the other on a stream share the counters and slabs; two streams have their
    own buffers"""
    ops = [_ops(dev, *s, seed=7 + i) for i, s in enumerate(XS_SHAPES)]
    ref = [(tf_gemm._pointwise_gemm_impl(x, w, False), tf_gemm._pointwise_gemm_tn_impl(x, dy)) for x, w, dy in ops]
    torch.cuda.synchronize()
    for _ in range(3):
        for (x, w, dy), (y0, dw0) in zip(ops, ref):
            assert torch.equal(tf_gemm._pointwise_gemm_impl(x, w, False), y0)
            assert torch.equal(tf_gemm._pointwise_gemm_tn_impl(x, dy), dw0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for k in range(4):
        for st, (x, w, dy) in ((s1, ops[0]), (s2, ops[1])):
            with torch.cuda.stream(st):
                outs.append((st is s2, tf_gemm._pointwise_gemm_impl(x, w, False), tf_gemm._pointwise_gemm_tn_impl(x, dy)))
    torch.cuda.synchronize()
    for second, y, dw in outs:
        assert torch.equal(y, ref[1 if second else 0][0]) and torch.equal(dw, ref[1 if second else 0][1])
    assert _lib.lib().sph3d_pointwise_gemm_exchange_failures() == 0              # no launch ever gave up waiting
    assert _lib.lib().sph3d_release_stream_scratch(s1.cuda_stream) >= 1          # the streams' exchange buffers (library-owned)
    assert _lib.lib().sph3d_release_stream_scratch(s2.cuda_stream) >= 1


def test_exchange_under_stream_capture(dev):
    """no allocation under capture: on a stream that has no exchange buffer yet the ordinary kernel is captured (same values within the
    products' rounding); on a stream whose buffer exists the exchange kernel is captured and its counters are back at zero after
    every replay (bit-equal replays, bit-equal to the eager call)"""
    l = _lib.lib()
    R, Ci, Co = 2048, 1024, 512
    x, w, _ = _ops(dev, R, Ci, Co, 11)
    want = (x.double() @ w.double())
    mag = x.double().abs() @ w.double().abs()

    def capture(stream, y):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            with torch.cuda.graph(g, stream=stream):
                rc = l.sph3d_pointwise_gemm(R, Ci, Co, _lib.ptr(x), _lib.ptr(w), None, 0, 0, _lib.ptr(y), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        return g

    fresh = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    y0 = torch.zeros(R, Co, device=dev)
    g0 = capture(fresh, y0)                              # no buffer on this stream: nothing may be allocated now
    g0.replay(); torch.cuda.synchronize()
    assert float(((y0.double() - want).abs() / mag).max()) <= 4e-7
    warm = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(warm):
        eager = tf_gemm._pointwise_gemm_impl(x, w, False)  # allocates the stream's buffer
    torch.cuda.synchronize()
    y1 = torch.zeros(R, Co, device=dev)
    g1 = capture(warm, y1)
    for _ in range(3):
        y1.zero_()
        g1.replay(); torch.cuda.synchronize()
        assert torch.equal(y1, eager)
    l.sph3d_release_stream_scratch(fresh.cuda_stream)
    l.sph3d_release_stream_scratch(warm.cuda_stream)


def test_exchange_with_bias_and_elu_epilogue(dev):
    R, Ci, Co = 2048, 1024, 512
    x, w, _ = _ops(dev, R, Ci, Co, 5)
    bias = torch.randn(Co, device=dev)
    y = tf_gemm._gemm_bias_act_impl(x, w, bias, 1)
    want = torch.nn.functional.elu(x.double() @ w.double() + bias.double())
    torch.testing.assert_close(y.double(), want, rtol=1e-5, atol=2e-5)
    assert torch.equal(tf_gemm._gemm_bias_act_impl(x, w, bias, 1), y)
