"""Fused ELU + batch normalisation — the tail of separable_conv3d / pointwise_conv3d
(utils/sph3gcn_util.py:152-161: activation_fn=tf.nn.elu then tf.layers.batch_normalization(momentum=0.99,
epsilon=1e-3); stock TF ops in the reference, SURVEY §8f item 3 here).

``elu_batch_norm(y, gamma, beta, moving_mean, moving_var, training)`` == batch_norm(elu(y)) over all but the last axis,
as ONE custom op (``sph3d::elu_bn``) whose backward needs only y.
"""
from typing import Tuple

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
    ws = torch.empty((wsb,), dtype=torch.uint8, device=y.device)
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
    ws = torch.empty((wsb,), dtype=torch.uint8, device=y.device)
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
