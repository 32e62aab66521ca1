"""Graph pooling — mirrors tf_ops/pooling/tf_pool3d.py:9-28.

Custom ops ``sph3d::max_pool3d`` (+ ``max_pool3d_grad``) and ``sph3d::avg_pool3d``
(+ ``avg_pool3d_grad``); gradients wired as the reference's RegisterGradient blocks.
"""
from typing import Tuple

import torch

from . import _lib, _tgraph


def _check_pool(input, nn_index, nn_count):
    if input.dim() != 3:
        raise ValueError("rank of input should be 3")
    if nn_index.dim() != 3:
        raise ValueError("rank of nn_index should be 3")
    if nn_count.dim() != 2:
        raise ValueError("rank of nn_count should be 2")


def _max_pool3d_impl(input: torch.Tensor, nn_index: torch.Tensor, nn_count: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    _lib.require_device(input, nn_index, nn_count)
    _check_pool(input, nn_index, nn_count)
    input, nn_index, nn_count = _lib.f32(input), _lib.i32(nn_index), _lib.i32(nn_count)
    B, N, C = input.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    output = torch.empty((B, M, C), dtype=torch.float32, device=input.device)
    max_index = torch.empty((B, M, C), dtype=torch.int32, device=input.device)
    _lib.check(_lib.lib().sph3d_max_pool3d(B, N, M, C, K, _lib.ptr(nn_index), _lib.ptr(nn_count), _lib.ptr(input),
                                           _lib.ptr(output), _lib.ptr(max_index), _lib.stream_ptr()))
    return output, max_index


_max_pool3d = torch.library.custom_op("sph3d::max_pool3d", mutates_args=())(_max_pool3d_impl)


@_max_pool3d.register_fake
def _(input, nn_index, nn_count):
    shape = (input.shape[0], nn_index.shape[1], input.shape[2])
    return input.new_empty(shape), input.new_empty(shape, dtype=torch.int32)


def _max_pool3d_grad_impl(input: torch.Tensor, grad_output: torch.Tensor, max_index: torch.Tensor) -> torch.Tensor:
    _lib.require_device(input, grad_output, max_index)
    grad_output, max_index = _lib.f32(grad_output), _lib.i32(max_index)
    B, N, C = input.shape
    M = grad_output.shape[1]
    grad_input = torch.empty((B, N, C), dtype=torch.float32, device=input.device)
    _lib.check(_lib.lib().sph3d_max_pool3d_grad(B, N, M, C, _lib.ptr(max_index), _lib.ptr(grad_output),
                                                _lib.ptr(grad_input), _lib.stream_ptr()))
    return grad_input


_max_pool3d_grad = torch.library.custom_op("sph3d::max_pool3d_grad", mutates_args=())(_max_pool3d_grad_impl)


@_max_pool3d_grad.register_fake
def _(input, grad_output, max_index):
    return torch.empty_like(input)


def _max_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], output[1])
    ctx.mark_non_differentiable(output[1])


def _max_backward(ctx, grad_output, grad_index):
    input, max_index = ctx.saved_tensors
    return _max_pool3d_grad(input, grad_output, max_index), None, None


_max_pool3d.register_autograd(_max_backward, setup_context=_max_setup)


def _avg_pool3d_impl(input: torch.Tensor, nn_index: torch.Tensor, nn_count: torch.Tensor) -> torch.Tensor:
    _lib.require_device(input, nn_index, nn_count)
    _check_pool(input, nn_index, nn_count)
    input, nn_index, nn_count = _lib.f32(input), _lib.i32(nn_index), _lib.i32(nn_count)
    B, N, C = input.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    output = torch.empty((B, M, C), dtype=torch.float32, device=input.device)
    _lib.check(_lib.lib().sph3d_avg_pool3d(B, N, M, C, K, _lib.ptr(nn_index), _lib.ptr(nn_count), _lib.ptr(input),
                                           _lib.ptr(output), _lib.stream_ptr()))
    return output


_avg_pool3d = torch.library.custom_op("sph3d::avg_pool3d", mutates_args=())(_avg_pool3d_impl)


@_avg_pool3d.register_fake
def _(input, nn_index, nn_count):
    return input.new_empty((input.shape[0], nn_index.shape[1], input.shape[2]))


def _avg_pool3d_grad_impl(input: torch.Tensor, grad_output: torch.Tensor, nn_index: torch.Tensor,
                     nn_count: torch.Tensor) -> torch.Tensor:
    _lib.require_device(input, grad_output, nn_index, nn_count)
    grad_output, nn_index, nn_count = _lib.f32(grad_output), _lib.i32(nn_index), _lib.i32(nn_count)
    B, N, C = input.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    grad_input = torch.empty((B, N, C), dtype=torch.float32, device=input.device)
    offsets, ent_key, ent_scale, _ = _tgraph.transpose(nn_index, nn_count, N)
    _lib.check(_lib.lib().sph3d_scatter_grad_t(B, N, M, C, _lib.ptr(offsets), _lib.ptr(ent_key), _lib.ptr(ent_scale),
                                               _lib.ptr(grad_output), _lib.ptr(grad_input), _lib.stream_ptr()))
    return grad_input


_avg_pool3d_grad = torch.library.custom_op("sph3d::avg_pool3d_grad", mutates_args=())(_avg_pool3d_grad_impl)


@_avg_pool3d_grad.register_fake
def _(input, grad_output, nn_index, nn_count):
    return torch.empty_like(input)


def _avg_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _avg_backward(ctx, grad_output):
    input, nn_index, nn_count = ctx.saved_tensors
    return _avg_pool3d_grad(input, grad_output, nn_index, nn_count), None, None


_avg_pool3d.register_autograd(_avg_backward, setup_context=_avg_setup)


def _max_pool3d_grad_t_impl(input, grad_output, max_index, nn_count, tg, addend=None):
    """the gradient as a gather over the transposed pooling graph tg = (offsets, ent_key, ...) (sph3d_max_pool3d_grad_t);
    addend: another gradient of `input` ([B,N,C]), added in by the same kernel"""
    grad_output, max_index = _lib.f32(grad_output), _lib.i32(max_index)
    addend = None if addend is None else _lib.f32(addend)
    B, N, C = input.shape
    M = grad_output.shape[1]
    grad_input = torch.empty((B, N, C), dtype=torch.float32, device=input.device)
    _lib.check(_lib.lib().sph3d_max_pool3d_grad_t(B, N, M, C, _lib.ptr(tg[0]), _lib.ptr(tg[1]), _lib.ptr(nn_count),
                                                  _lib.ptr(max_index), _lib.ptr(grad_output), _lib.ptr(addend), _lib.ptr(grad_input),
                                                  _lib.stream_ptr()))
    return grad_input


class _MaxPool3dFn(torch.autograd.Function):      # eager fast path (see tf_conv3d._DepthwiseConv3dFn)
    @staticmethod
    def forward(ctx, input, nn_index, nn_count):
        output, max_index = _max_pool3d_impl(input, nn_index, nn_count)
        ctx.save_for_backward(input, max_index, nn_index, nn_count)
        ctx.mark_non_differentiable(max_index)
        return output, max_index

    @staticmethod
    def backward(ctx, grad_output, grad_index):
        input, max_index, nn_index, nn_count = ctx.saved_tensors
        # a transposed pooling graph someone built ahead of time WITH the promise that no row repeats a point (the harness
        # does, on its graph stream: rows of the ball query): gather, no atomics; otherwise the reference's scatter, which adds
        # a point's gradient once however often a row lists it (tf_pool3d_gpu.cu:38-50)
        tg = None
        if input.is_cuda and nn_index.dtype == torch.int32 and nn_count.dtype == torch.int32:
            tg = _tgraph.peek(nn_index, nn_count, input.shape[1], need_unique_rows=True)
        if tg is not None:
            return _max_pool3d_grad_t_impl(input, grad_output, max_index, nn_count, tg), None, None
        return _max_pool3d_grad_impl(input, grad_output, max_index), None, None


class _AvgPool3dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, nn_index, nn_count):
        ctx.save_for_backward(input, nn_index, nn_count)
        return _avg_pool3d_impl(input, nn_index, nn_count)

    @staticmethod
    def backward(ctx, grad_output):
        input, nn_index, nn_count = ctx.saved_tensors
        return _avg_pool3d_grad_impl(input, grad_output, nn_index, nn_count), None, None


def max_pool3d(input, nn_index, nn_count):
    return _MaxPool3dFn.apply(input, nn_index, nn_count)


class _MaxPool3dSkipFn(torch.autograd.Function):
    """max_pool3d whose input is ALSO used by a skip connection: returns (pooled, max_index, input as the skip tensor); the two
    gradients of the input — through the pooling and through the skip — meet inside the pooling gradient's gather kernel
    instead of in autograd's elementwise accumulation (three tensors through memory)."""

    @staticmethod
    def forward(ctx, input, nn_index, nn_count):
        output, max_index = _max_pool3d_impl(input, nn_index, nn_count)
        ctx.save_for_backward(input, max_index, nn_index, nn_count)
        ctx.mark_non_differentiable(max_index)
        return output, max_index, input.view_as(input)

    @staticmethod
    def backward(ctx, grad_output, grad_index, grad_skip):
        input, max_index, nn_index, nn_count = ctx.saved_tensors
        tg = None
        if input.is_cuda and nn_index.dtype == torch.int32 and nn_count.dtype == torch.int32:
            tg = _tgraph.peek(nn_index, nn_count, input.shape[1], need_unique_rows=True)
        if grad_output is None:
            return grad_skip, None, None
        if tg is not None:
            skip = grad_skip.contiguous() if grad_skip is not None else None
            return _max_pool3d_grad_t_impl(input, grad_output, max_index, nn_count, tg, addend=skip), None, None
        g = _max_pool3d_grad_impl(input, grad_output, max_index)
        return (g if grad_skip is None else g + grad_skip), None, None


def max_pool3d_with_skip(input, nn_index, nn_count):
    """-> (pooled, max_index, skip): `skip` is `input` for whoever else consumes it (see _MaxPool3dSkipFn)"""
    return _MaxPool3dSkipFn.apply(input, nn_index, nn_count)


def max_pool3d_grad(input, grad_output, max_index):
    return _max_pool3d_grad_impl(input, grad_output, max_index)


def avg_pool3d(input, nn_index, nn_count):
    return _AvgPool3dFn.apply(input, nn_index, nn_count)


def avg_pool3d_grad(input, grad_output, nn_index, nn_count):
    return _avg_pool3d_grad_impl(input, grad_output, nn_index, nn_count)
