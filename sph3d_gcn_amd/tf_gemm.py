"""Pointwise 1x1 feature GEMM — the ``tf.matmul`` inside separable_conv3d / pointwise_conv3d /
fully_connected (utils/sph3gcn_util.py:146-150, 204-206, 260; cuBLAS SGEMM in the reference).

``matmul(x, w)`` computes x[R,Cin] @ w[Cin,Cout] in exact fp32 with libsph3d's hand-written fp32-MFMA kernels
(sph3d_pointwise_gemm*): custom op ``sph3d::pointwise_gemm`` with its two backward products.  (The library yardstick,
torch.matmul on rocBLAS / hipBLASLt, lives in tools/exp_gemm.py, not here.)
"""
import torch

from . import _lib


def _pointwise_gemm_impl(x: torch.Tensor, w: torch.Tensor, trans_w: bool) -> torch.Tensor:
    """y[R,Cout] = x[R,Cin] @ w[Cin,Cout]   (trans_w: w is stored [Cout,Cin])"""
    _lib.require_device(x, w)
    x, w = _lib.f32(x), _lib.f32(w)
    R, Cin = x.shape
    Cout = w.shape[0] if trans_w else w.shape[1]
    y = torch.empty((R, Cout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().sph3d_pointwise_gemm(R, Cin, Cout, _lib.ptr(x), _lib.ptr(w), None, 0, int(trans_w),
                                               _lib.ptr(y), _lib.stream_ptr()))
    return y


_pointwise_gemm = torch.library.custom_op("sph3d::pointwise_gemm", mutates_args=())(_pointwise_gemm_impl)


@_pointwise_gemm.register_fake
def _(x, w, trans_w):
    return x.new_empty((x.shape[0], w.shape[0] if trans_w else w.shape[1]))


def _pointwise_gemm_tn_impl(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """dw[Cin,Cout] = x[R,Cin]^T @ dy[R,Cout]"""
    _lib.require_device(x, dy)
    x, dy = _lib.f32(x), _lib.f32(dy)
    R, Cin = x.shape
    Cout = dy.shape[1]
    dw = torch.empty((Cin, Cout), dtype=torch.float32, device=x.device)
    l = _lib.lib()
    wsb = l.sph3d_pointwise_gemm_tn_workspace(R, Cin, Cout)
    ws = _lib.scratch(wsb, x.device)
    _lib.check(l.sph3d_pointwise_gemm_tn(R, Cin, Cout, _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(ws), wsb,
                                         _lib.stream_ptr()))
    return dw


_pointwise_gemm_tn = torch.library.custom_op("sph3d::pointwise_gemm_tn", mutates_args=())(_pointwise_gemm_tn_impl)


@_pointwise_gemm_tn.register_fake
def _(x, dy):
    return x.new_empty((x.shape[1], dy.shape[1]))


def _gemm_setup(ctx, inputs, output):
    x, w, trans_w = inputs
    ctx.save_for_backward(x, w)
    ctx.trans_w = trans_w


def _gemm_backward(ctx, dy):
    x, w = ctx.saved_tensors
    if ctx.trans_w:
        raise RuntimeError("gradient of the transposed-weight form is not needed by the path")
    dx = _pointwise_gemm(dy, w, True) if ctx.needs_input_grad[0] else None
    dw = _pointwise_gemm_tn(x, dy) if ctx.needs_input_grad[1] else None
    return dx, dw, None


_pointwise_gemm.register_autograd(_gemm_backward, setup_context=_gemm_setup)


class _PointwiseGemmFn(torch.autograd.Function):      # eager fast path (see tf_conv3d._DepthwiseConv3dFn)
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return _pointwise_gemm_impl(x, w, False)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = _pointwise_gemm_impl(dy, w, True) if ctx.needs_input_grad[0] else None
        dw = _pointwise_gemm_tn_impl(x, dy) if ctx.needs_input_grad[1] else None
        return dx, dw


def matmul(x, w):
    return _PointwiseGemmFn.apply(x, w)


# ---- GEMM with the bias / ELU epilogue of the library (layers with biases and no batch norm) --------------------------
def _gemm_bias_act_impl(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, act: int) -> torch.Tensor:
    """elu?(x[R,Cin] @ w[Cin,Cout] + bias[Cout]); act = 0 (none) | 1 (ELU): bias and activation run in the GEMM's epilogue"""
    _lib.require_device(x, w, bias)
    x, w, bias = _lib.f32(x), _lib.f32(w), _lib.f32(bias)
    R, Cin = x.shape
    Cout = w.shape[1]
    y = torch.empty((R, Cout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().sph3d_pointwise_gemm(R, Cin, Cout, _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), int(act), 0,
                                               _lib.ptr(y), _lib.stream_ptr()))
    return y


_gemm_bias_act = torch.library.custom_op("sph3d::pointwise_gemm_bias_act", mutates_args=())(_gemm_bias_act_impl)


@_gemm_bias_act.register_fake
def _(x, w, bias, act):
    return x.new_empty((x.shape[0], w.shape[1]))


class _GemmBiasActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, act):
        out = _gemm_bias_act_impl(x, w, bias, act)
        ctx.save_for_backward(x, w, out)
        ctx.act = act
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, out = ctx.saved_tensors
        # ELU'(z) from the output: 1 for z > 0, elu(z) + 1 = exp(z) otherwise
        g = dout * torch.where(out > 0, torch.ones_like(out), out + 1.0) if ctx.act == 1 else dout
        g = g.contiguous()
        dx = _pointwise_gemm_impl(g, w, True) if ctx.needs_input_grad[0] else None
        dw = _pointwise_gemm_tn_impl(x, g) if ctx.needs_input_grad[1] else None
        db = g.sum(0) if ctx.needs_input_grad[2] else None
        return dx, dw, db, None


def matmul_bias_act(x, w, bias, elu=False):
    """elu?(x @ w + bias) with the bias and the activation in the GEMM's epilogue"""
    return _GemmBiasActFn.apply(x, w, bias, 1 if elu else 0)


# ---- the 1x1 layer with few outputs over two operand halves: the logits layer without its concatenation (csrc/skinny.hip) ----
def skinny_supported(R, K1, K2, N):
    return bool(_lib.lib().sph3d_pointwise_gemm_skinny_supported(int(R), int(K1), int(K2), int(N)))


def _skinny_impl(a1: torch.Tensor, a2: torch.Tensor, w: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """a1[R,K1] @ w[:K1] + a2[R,K2] @ w[K1:] + bias -> [R,N]   (a2 / bias with zero elements: absent)"""
    _lib.require_device(a1, w)
    a1, w = _lib.f32(a1), _lib.f32(w)
    a2 = None if a2 is None or a2.numel() == 0 else _lib.f32(a2)
    bias = None if bias is None or bias.numel() == 0 else _lib.f32(bias)
    R, K1 = a1.shape
    K2 = 0 if a2 is None else a2.shape[1]
    N = w.shape[1]
    if w.shape[0] != K1 + K2:
        raise ValueError("weights should be [K1 + K2, N]")
    y = torch.empty((R, N), dtype=torch.float32, device=a1.device)
    _lib.check(_lib.lib().sph3d_pointwise_gemm_skinny(R, K1, K2, N, _lib.ptr(a1), _lib.ptr(a2), _lib.ptr(w), _lib.ptr(bias),
                                                      _lib.ptr(y), _lib.stream_ptr()))
    return y


def _skinny_tn_impl(a1: torch.Tensor, a2: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """[a1 | a2]^T @ dy -> [K1 + K2, N]"""
    _lib.require_device(a1, dy)
    a1, dy = _lib.f32(a1), _lib.f32(dy)
    a2 = None if a2 is None or a2.numel() == 0 else _lib.f32(a2)
    R, K1 = a1.shape
    K2 = 0 if a2 is None else a2.shape[1]
    N = dy.shape[1]
    l = _lib.lib()
    wsb = l.sph3d_pointwise_gemm_skinny_tn_workspace(R, K1, K2, N)
    ws = _lib.scratch(wsb, a1.device)
    dw = torch.empty((K1 + K2, N), dtype=torch.float32, device=a1.device)
    _lib.check(l.sph3d_pointwise_gemm_skinny_tn(R, K1, K2, N, _lib.ptr(a1), _lib.ptr(a2), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(ws), wsb,
                                                _lib.stream_ptr()))
    return dw


_skinny = torch.library.custom_op("sph3d::pointwise_gemm_skinny", mutates_args=())(_skinny_impl)


@_skinny.register_fake
def _(a1, a2, w, bias):
    return a1.new_empty((a1.shape[0], w.shape[1]))


_skinny_tn = torch.library.custom_op("sph3d::pointwise_gemm_skinny_tn", mutates_args=())(_skinny_tn_impl)


@_skinny_tn.register_fake
def _(a1, a2, dy):
    return a1.new_empty((a1.shape[1] + a2.shape[1], dy.shape[1]))


class _SkinnyLinear2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a1, a2, w, bias):
        ctx.save_for_backward(a1, a2, w)
        ctx.has_bias = bias is not None
        return _skinny_impl(a1, a2, w, bias)

    @staticmethod
    def backward(ctx, dy):
        a1, a2, w = ctx.saved_tensors
        dy = dy.contiguous()
        K1 = a1.shape[1]
        # the input gradients: the general product with the matching rows of w, one call per half (they are written where autograd
        # wants them: no slices of a concatenated gradient to copy)
        da1 = _pointwise_gemm_impl(dy, w[:K1], True) if ctx.needs_input_grad[0] else None
        da2 = _pointwise_gemm_impl(dy, w[K1:], True) if (a2 is not None and ctx.needs_input_grad[1]) else None
        dw = _skinny_tn_impl(a1, a2, dy) if ctx.needs_input_grad[2] else None
        db = dy.sum(0) if (ctx.has_bias and ctx.needs_input_grad[3]) else None
        return da1, da2, dw, db


def linear_concat2(a1, a2, w, bias=None):
    """[a1 | a2] @ w + bias for few output columns (N <= 16), the concatenation never materialised; a2 may be None"""
    return _SkinnyLinear2Fn.apply(a1, a2, w, bias)
