"""Fused ELU + batch normalisation — the tail of separable_conv3d / pointwise_conv3d
(utils/sph3gcn_util.py:152-161: activation_fn=tf.nn.elu then tf.layers.batch_normalization(momentum=0.99,
epsilon=1e-3); stock TF ops in the reference, SURVEY §8f item 3 here).

``elu_batch_norm(y, gamma, beta, moving_mean, moving_var, training)`` == batch_norm(elu(y)) over all but the last axis,
as ONE custom op (``sph3d::elu_bn``) whose backward needs only y.
"""
from typing import Optional, Tuple

import torch

from . import _lib

MOMENTUM = 0.99       # tf.layers.batch_normalization(momentum=0.99): weight of the OLD moving statistics
EPSILON = 1e-3        # TF default epsilon


def supported(C):
    return C % 4 == 0 and C <= 1024


def _fwd_impl(y: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, moving_mean: torch.Tensor,
              moving_var: torch.Tensor, training: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    _lib.require_device(y, gamma, beta, moving_mean, moving_var)
    y = _lib.f32(y)
    C = y.shape[-1]
    R = y.numel() // C
    out = torch.empty_like(y)
    save_mean = torch.empty((C,), dtype=torch.float32, device=y.device)
    save_rstd = torch.empty((C,), dtype=torch.float32, device=y.device)
    l = _lib.lib()
    wsb = l.sph3d_elu_bn_workspace(R, C)
    ws = _lib.scratch(wsb, y.device)
    _lib.check(l.sph3d_elu_bn_forward(R, C, _lib.ptr(y), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(moving_mean),
                                      _lib.ptr(moving_var), 1.0 - MOMENTUM, EPSILON, 1 if training else 0, _lib.ptr(out),
                                      _lib.ptr(save_mean), _lib.ptr(save_rstd), _lib.ptr(ws), wsb, _lib.stream_ptr()))
    return out, save_mean, save_rstd


def _bwd_impl(y: torch.Tensor, dout: torch.Tensor, gamma: torch.Tensor, save_mean: torch.Tensor,
              save_rstd: torch.Tensor, training: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    _lib.require_device(y, dout, gamma)
    y, dout = _lib.f32(y), _lib.f32(dout)
    C = y.shape[-1]
    R = y.numel() // C
    dy = torch.empty_like(y)
    dgamma = torch.empty((C,), dtype=torch.float32, device=y.device)
    dbeta = torch.empty((C,), dtype=torch.float32, device=y.device)
    l = _lib.lib()
    wsb = l.sph3d_elu_bn_workspace(R, C)
    ws = _lib.scratch(wsb, y.device)
    _lib.check(l.sph3d_elu_bn_backward(R, C, _lib.ptr(y), _lib.ptr(dout), _lib.ptr(gamma), _lib.ptr(save_mean),
                                       _lib.ptr(save_rstd), 1 if training else 0, _lib.ptr(dy), _lib.ptr(dgamma),
                                       _lib.ptr(dbeta), _lib.ptr(ws), wsb, _lib.stream_ptr()))
    return dy, dgamma, dbeta


_elu_bn = torch.library.custom_op("sph3d::elu_bn", mutates_args=("moving_mean", "moving_var"))(_fwd_impl)
_elu_bn_grad = torch.library.custom_op("sph3d::elu_bn_grad", mutates_args=())(_bwd_impl)


@_elu_bn.register_fake
def _(y, gamma, beta, moving_mean, moving_var, training):
    return torch.empty_like(y), torch.empty_like(gamma), torch.empty_like(gamma)


@_elu_bn_grad.register_fake
def _(y, dout, gamma, save_mean, save_rstd, training):
    return torch.empty_like(y), torch.empty_like(gamma), torch.empty_like(gamma)


# (sph3d::elu_bn updates the moving statistics in place, so the dispatcher accepts no autograd formula for it: the
# gradient wiring of the fused tail is _EluBnFn below, over the functional sph3d::elu_bn_grad.)


class _EluBnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, moving_mean, moving_var, training):
        out, save_mean, save_rstd = _fwd_impl(y, gamma, beta, moving_mean, moving_var, training)
        ctx.save_for_backward(y, gamma, save_mean, save_rstd)
        ctx.training = training
        return out

    @staticmethod
    def backward(ctx, dout):
        y, gamma, save_mean, save_rstd = ctx.saved_tensors
        dy, dgamma, dbeta = _bwd_impl(y, dout, gamma, save_mean, save_rstd, ctx.training)
        return dy, dgamma, dbeta, None, None, None


def elu_batch_norm(y, gamma, beta, moving_mean, moving_var, training=True):
    return _EluBnFn.apply(y, gamma, beta, moving_mean, moving_var, bool(training))


# ---- layer tail with the statistics in the GEMM's epilogue (first half) ---------------------------------
def gemm_bn_blocks(R, Cin, Cout):
    """row blocks of partial statistics sph3d_pointwise_gemm_bnstats writes for this shape; 0 = shape not covered"""
    return int(_lib.lib().sph3d_pointwise_gemm_bnstats_blocks(int(R), int(Cin), int(Cout)))


def _gemm_bnstats_impl(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """y[R,Cout] = x @ w (+ bias) and partial[nblk, 2, Cout] = per row block (sum elu(y), sum elu(y)^2)"""
    _lib.require_device(x, w)
    x, w = _lib.f32(x), _lib.f32(w)
    if bias is not None:
        bias = _lib.f32(bias)
    R, Cin = x.shape
    Cout = w.shape[1]
    nblk = gemm_bn_blocks(R, Cin, Cout)
    if nblk == 0:
        raise ValueError("pointwise_gemm_bnstats: shape (%d, %d -> %d) is not covered (whole tiles needed)" % (R, Cin, Cout))
    y = torch.empty((R, Cout), dtype=torch.float32, device=x.device)
    partial = torch.empty((nblk, 2, Cout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().sph3d_pointwise_gemm_bnstats(R, Cin, Cout, _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(y),
                                                        _lib.ptr(partial), _lib.stream_ptr()))
    return y, partial


def _elu_bn_partials_impl(y: torch.Tensor, partial: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                          moving_mean: torch.Tensor, moving_var: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """training-mode elu_bn whose statistics come from `partial` instead of a pass over y"""
    _lib.require_device(y, partial, gamma, beta, moving_mean, moving_var)
    y, partial = _lib.f32(y), _lib.f32(partial)
    C = y.shape[-1]
    R = y.numel() // C
    out = torch.empty_like(y)
    save_mean = torch.empty((C,), dtype=torch.float32, device=y.device)
    save_rstd = torch.empty((C,), dtype=torch.float32, device=y.device)
    _lib.check(_lib.lib().sph3d_elu_bn_forward_partials(R, C, partial.shape[0], _lib.ptr(partial), _lib.ptr(y), _lib.ptr(gamma),
                                                         _lib.ptr(beta), _lib.ptr(moving_mean), _lib.ptr(moving_var),
                                                         1.0 - MOMENTUM, EPSILON, _lib.ptr(out), _lib.ptr(save_mean),
                                                         _lib.ptr(save_rstd), _lib.stream_ptr()))
    return out, save_mean, save_rstd


_gemm_bnstats = torch.library.custom_op("sph3d::pointwise_gemm_bnstats", mutates_args=())(_gemm_bnstats_impl)
_elu_bn_partials = torch.library.custom_op("sph3d::elu_bn_partials", mutates_args=("moving_mean", "moving_var"))(_elu_bn_partials_impl)


@_gemm_bnstats.register_fake
def _(x, w, bias=None):
    nblk = gemm_bn_blocks(x.shape[0], x.shape[1], w.shape[1])
    return x.new_empty((x.shape[0], w.shape[1])), x.new_empty((nblk, 2, w.shape[1]))


@_elu_bn_partials.register_fake
def _(y, partial, gamma, beta, moving_mean, moving_var):
    return torch.empty_like(y), torch.empty_like(gamma), torch.empty_like(gamma)


class _GemmEluBnFn(torch.autograd.Function):
    """out = batch_norm(elu(x @ w)) in training mode: GEMM (statistics in its epilogue) -> finalize -> apply.  Saves the raw
    product y; the backward is elu_bn's followed by the GEMM's two gradient products."""

    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, moving_mean, moving_var):
        y, partial = _gemm_bnstats_impl(x, w, bias)
        out, save_mean, save_rstd = _elu_bn_partials_impl(y, partial, gamma, beta, moving_mean, moving_var)
        ctx.save_for_backward(x, w, y, gamma, save_mean, save_rstd)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import tf_gemm
        x, w, y, gamma, save_mean, save_rstd = ctx.saved_tensors
        dy, dgamma, dbeta = _bwd_impl(y, dout, gamma, save_mean, save_rstd, True)
        dx = tf_gemm._pointwise_gemm_impl(dy, w, True) if ctx.needs_input_grad[0] else None
        dw = tf_gemm._pointwise_gemm_tn_impl(x, dy) if ctx.needs_input_grad[1] else None
        db = dy.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, dgamma, dbeta, None, None


def gemm_elu_batch_norm(x, w, gamma, beta, moving_mean, moving_var, bias=None):
    """x[R,Cin] @ w[Cin,Cout] (+ bias) -> ELU -> batch norm (training statistics), one fused tail"""
    return _GemmEluBnFn.apply(x, w, bias, gamma, beta, moving_mean, moving_var)
