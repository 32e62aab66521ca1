"""Graph un-pooling (feature interpolation) — mirrors tf_ops/unpooling/tf_unpool3d.py:9-28.

Custom ops ``sph3d::mean_interpolate`` / ``sph3d::weighted_interpolate`` (+ ``_grad``).
As in the reference (:21-28) ``weight`` receives no gradient.
"""
import torch

from . import _lib, _tgraph


def _check(input, nn_index, nn_count):
    if input.dim() != 3:
        raise ValueError("rank of input should be 3")
    if nn_index.dim() != 3:
        raise ValueError("rank of nn_index should be 3")
    if nn_count.dim() != 2:
        raise ValueError("rank of nn_count should be 2")


def _mean_interpolate_impl(input: torch.Tensor, nn_index: torch.Tensor, nn_count: torch.Tensor) -> torch.Tensor:
    _lib.require_device(input, nn_index, nn_count)
    _check(input, nn_index, nn_count)
    input, nn_index, nn_count = _lib.f32(input), _lib.i32(nn_index), _lib.i32(nn_count)
    B, M, C = input.shape
    N, K = nn_index.shape[1], nn_index.shape[2]
    output = torch.empty((B, N, C), dtype=torch.float32, device=input.device)
    _lib.check(_lib.lib().sph3d_mean_interpolate(B, N, M, C, K, _lib.ptr(nn_index), _lib.ptr(nn_count),
                                                 _lib.ptr(input), _lib.ptr(output), _lib.stream_ptr()))
    return output


_mean_interpolate = torch.library.custom_op("sph3d::mean_interpolate", mutates_args=())(_mean_interpolate_impl)


@_mean_interpolate.register_fake
def _(input, nn_index, nn_count):
    return input.new_empty((input.shape[0], nn_index.shape[1], input.shape[2]))


def _mean_interpolate_grad_impl(input: torch.Tensor, grad_output: torch.Tensor, nn_index: torch.Tensor,
                           nn_count: torch.Tensor) -> torch.Tensor:
    _lib.require_device(input, grad_output, nn_index, nn_count)
    grad_output, nn_index, nn_count = _lib.f32(grad_output), _lib.i32(nn_index), _lib.i32(nn_count)
    B, M, C = input.shape
    N, K = nn_index.shape[1], nn_index.shape[2]
    grad_input = torch.empty((B, M, C), dtype=torch.float32, device=input.device)
    offsets, ent_key, ent_scale, _ = _tgraph.transpose(nn_index, nn_count, M)   # source points = the M coarse points
    _lib.check(_lib.lib().sph3d_scatter_grad_t(B, M, N, C, _lib.ptr(offsets), _lib.ptr(ent_key), _lib.ptr(ent_scale),
                                               _lib.ptr(grad_output), _lib.ptr(grad_input), _lib.stream_ptr()))
    return grad_input


_mean_interpolate_grad = torch.library.custom_op("sph3d::mean_interpolate_grad", mutates_args=())(_mean_interpolate_grad_impl)


@_mean_interpolate_grad.register_fake
def _(input, grad_output, nn_index, nn_count):
    return torch.empty_like(input)


def _mean_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _mean_backward(ctx, grad_output):
    input, nn_index, nn_count = ctx.saved_tensors
    return _mean_interpolate_grad(input, grad_output, nn_index, nn_count), None, None


_mean_interpolate.register_autograd(_mean_backward, setup_context=_mean_setup)


def _weighted_interpolate_impl(input: torch.Tensor, weight: torch.Tensor, nn_index: torch.Tensor,
                          nn_count: torch.Tensor) -> torch.Tensor:
    _lib.require_device(input, weight, nn_index, nn_count)
    _check(input, nn_index, nn_count)
    input, weight = _lib.f32(input), _lib.f32(weight)
    nn_index, nn_count = _lib.i32(nn_index), _lib.i32(nn_count)
    B, M, C = input.shape
    N, K = nn_index.shape[1], nn_index.shape[2]
    output = torch.empty((B, N, C), dtype=torch.float32, device=input.device)
    _lib.check(_lib.lib().sph3d_weighted_interpolate(B, N, M, C, K, _lib.ptr(nn_index), _lib.ptr(nn_count),
                                                     _lib.ptr(input), _lib.ptr(weight), _lib.ptr(output),
                                                     _lib.stream_ptr()))
    return output


_weighted_interpolate = torch.library.custom_op("sph3d::weighted_interpolate", mutates_args=())(_weighted_interpolate_impl)


@_weighted_interpolate.register_fake
def _(input, weight, nn_index, nn_count):
    return input.new_empty((input.shape[0], nn_index.shape[1], input.shape[2]))


def _weighted_interpolate_grad_impl(input: torch.Tensor, grad_output: torch.Tensor, weight: torch.Tensor,
                               nn_index: torch.Tensor, nn_count: torch.Tensor) -> torch.Tensor:
    _lib.require_device(input, grad_output, weight, nn_index, nn_count)
    grad_output, weight = _lib.f32(grad_output), _lib.f32(weight)
    nn_index, nn_count = _lib.i32(nn_index), _lib.i32(nn_count)
    B, M, C = input.shape
    N, K = nn_index.shape[1], nn_index.shape[2]
    grad_input = torch.empty((B, M, C), dtype=torch.float32, device=input.device)
    offsets, ent_key, ent_scale, _ = _tgraph.transpose(nn_index, nn_count, M, weight=weight)
    _lib.check(_lib.lib().sph3d_scatter_grad_t(B, M, N, C, _lib.ptr(offsets), _lib.ptr(ent_key), _lib.ptr(ent_scale),
                                               _lib.ptr(grad_output), _lib.ptr(grad_input), _lib.stream_ptr()))
    return grad_input


_weighted_interpolate_grad = torch.library.custom_op("sph3d::weighted_interpolate_grad", mutates_args=())(_weighted_interpolate_grad_impl)


@_weighted_interpolate_grad.register_fake
def _(input, grad_output, weight, nn_index, nn_count):
    return torch.empty_like(input)


def _w_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _w_backward(ctx, grad_output):
    input, weight, nn_index, nn_count = ctx.saved_tensors
    return _weighted_interpolate_grad(input, grad_output, weight, nn_index, nn_count), None, None, None


_weighted_interpolate.register_autograd(_w_backward, setup_context=_w_setup)


class _MeanInterpolateFn(torch.autograd.Function):      # eager fast path (see tf_conv3d._DepthwiseConv3dFn)
    @staticmethod
    def forward(ctx, input, nn_index, nn_count):
        ctx.save_for_backward(input, nn_index, nn_count)
        return _mean_interpolate_impl(input, nn_index, nn_count)

    @staticmethod
    def backward(ctx, grad_output):
        input, nn_index, nn_count = ctx.saved_tensors
        return _mean_interpolate_grad_impl(input, grad_output, nn_index, nn_count), None, None


class _WeightedInterpolateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, weight, nn_index, nn_count):
        ctx.save_for_backward(input, weight, nn_index, nn_count)
        return _weighted_interpolate_impl(input, weight, nn_index, nn_count)

    @staticmethod
    def backward(ctx, grad_output):
        input, weight, nn_index, nn_count = ctx.saved_tensors
        return _weighted_interpolate_grad_impl(input, grad_output, weight, nn_index, nn_count), None, None, None


def mean_interpolate(input, nn_index, nn_count):
    return _MeanInterpolateFn.apply(input, nn_index, nn_count)


def mean_interpolate_grad(input, grad_output, nn_index, nn_count):
    return _mean_interpolate_grad_impl(input, grad_output, nn_index, nn_count)


def weighted_interpolate(input, weight, nn_index, nn_count):
    return _WeightedInterpolateFn.apply(input, weight, nn_index, nn_count)


def weighted_interpolate_grad(input, grad_output, weight, nn_index, nn_count):
    return _weighted_interpolate_grad_impl(input, grad_output, weight, nn_index, nn_count)
