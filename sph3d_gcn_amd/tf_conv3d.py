"""Depthwise spherical convolution — mirrors tf_ops/convolution/tf_conv3d.py:10-32.

PyTorch custom ops ``sph3d::depthwise_conv3d`` and ``sph3d::depthwise_conv3d_grad``; the
gradient wiring is the reference's @RegisterGradient("DepthwiseConv3d") (:23-32):
[grad_input, grad_filter, None, None, None].
"""
from typing import Tuple

import torch
import torch.nn.functional as Fn

from . import _lib, _tgraph


def _channel_pad(C, r):
    """channels to add so that the 4-channels-per-lane kernels apply (they need C % 4 == 0 and r in {1, 2}); measured on
    the ModelNet level-0 shapes: C=67 r=1 gradient 7.5 -> 1.5 ms, C=35 r=2 forward 0.88 -> 0.39 ms"""
    return (-C) % 4 if r in (1, 2) else 0


def _check_conv(input, filter, nn_index, nn_count, bin_index):
    # DepthwiseConv3dGpuOp::Compute checks (tf_conv3d.cpp:66-72)
    if input.dim() != 3:
        raise ValueError("rank of input should be 3")
    if filter.dim() != 3:
        raise ValueError("rank of filter should be 3")
    if filter.shape[1] != input.shape[2]:
        raise ValueError("Input Channel Size error!")
    if nn_index.dim() != 3 or bin_index.dim() != 3:
        raise ValueError("rank of nn_index should be 3")
    if nn_count.dim() != 2:
        raise ValueError("rank of nn_count should be 2")


def _depthwise_conv3d_impl(input: torch.Tensor, filter: torch.Tensor, nn_index: torch.Tensor,
                      nn_count: torch.Tensor, bin_index: torch.Tensor) -> torch.Tensor:
    _lib.require_device(input, filter, nn_index, nn_count, bin_index)
    _check_conv(input, filter, nn_index, nn_count, bin_index)
    input, filter = _lib.f32(input), _lib.f32(filter)
    nn_index, nn_count, bin_index = _lib.i32(nn_index), _lib.i32(nn_count), _lib.i32(bin_index)
    B, N, C = input.shape
    F, _, r = filter.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    pad = _channel_pad(C, r)
    if pad:
        # odd channel counts (ModelNet: 35, 67, 131): zero-pad to a multiple of 4 so the vector kernels apply; the padded
        # channels come last in the output (channel = c*r + rho) and are dropped.  Same arithmetic per real channel.
        out = _depthwise_conv3d_impl(Fn.pad(input, (0, pad)), Fn.pad(filter, (0, 0, 0, pad)), nn_index, nn_count, bin_index)
        return out[:, :, :C * r].contiguous()
    output = torch.empty((B, M, C * r), dtype=torch.float32, device=input.device)
    _lib.check(_lib.lib().sph3d_depthwise_conv3d(
        B, N, M, F, C, r, K, _lib.ptr(nn_index), _lib.ptr(nn_count), _lib.ptr(bin_index),
        _lib.ptr(input), _lib.ptr(filter), _lib.ptr(output), _lib.stream_ptr()))
    return output


_depthwise_conv3d = torch.library.custom_op("sph3d::depthwise_conv3d", mutates_args=())(_depthwise_conv3d_impl)


@_depthwise_conv3d.register_fake
def _(input, filter, nn_index, nn_count, bin_index):
    return input.new_empty((input.shape[0], nn_index.shape[1], input.shape[2] * filter.shape[2]))


def _depthwise_conv3d_grad_impl(input: torch.Tensor, filter: torch.Tensor, grad_output: torch.Tensor,
                           nn_index: torch.Tensor, nn_count: torch.Tensor,
                           bin_index: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    _lib.require_device(input, filter, grad_output, nn_index, nn_count, bin_index)
    _check_conv(input, filter, nn_index, nn_count, bin_index)
    input, filter, grad_output = _lib.f32(input), _lib.f32(filter), _lib.f32(grad_output)
    nn_index, nn_count, bin_index = _lib.i32(nn_index), _lib.i32(nn_count), _lib.i32(bin_index)
    B, N, C = input.shape
    F, _, r = filter.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    pad = _channel_pad(C, r)
    if pad:
        gi, gf = _depthwise_conv3d_grad_impl(Fn.pad(input, (0, pad)), Fn.pad(filter, (0, 0, 0, pad)),
                                             Fn.pad(grad_output, (0, pad * r)), nn_index, nn_count, bin_index)
        return gi[:, :, :C].contiguous(), gf[:, :C, :].contiguous()
    grad_input = torch.empty_like(input)
    grad_filter = torch.empty_like(filter)
    # gather over the transposed graph (built once per graph, shared by every gradient that uses it)
    offsets, ent_key, ent_scale, active = _tgraph.transpose(nn_index, nn_count, N, bin_index=bin_index, num_bins=F)
    l = _lib.lib()
    wsb = l.sph3d_depthwise_conv3d_grad_t_workspace(B, N, F, C, r)
    ws = _lib.scratch(wsb, input.device)
    _lib.check(l.sph3d_depthwise_conv3d_grad_t(
        B, N, M, F, C, r, _lib.ptr(offsets), _lib.ptr(ent_key), _lib.ptr(ent_scale),
        _lib.ptr(_tgraph.source_order(nn_index)), _lib.ptr(active),
        _lib.ptr(input), _lib.ptr(filter), _lib.ptr(grad_output), _lib.ptr(grad_input), _lib.ptr(grad_filter),
        _lib.ptr(ws), wsb, _lib.stream_ptr()))
    return grad_input, grad_filter


_depthwise_conv3d_grad = torch.library.custom_op("sph3d::depthwise_conv3d_grad", mutates_args=())(_depthwise_conv3d_grad_impl)


@_depthwise_conv3d_grad.register_fake
def _(input, filter, grad_output, nn_index, nn_count, bin_index):
    return torch.empty_like(input), torch.empty_like(filter)


def _conv_setup(ctx, inputs, output):
    input, filter, nn_index, nn_count, bin_index = inputs
    ctx.save_for_backward(input, filter, nn_index, nn_count, bin_index)


def _conv_backward(ctx, grad_output):
    input, filter, nn_index, nn_count, bin_index = ctx.saved_tensors
    grad_input, grad_filter = _depthwise_conv3d_grad(input, filter, grad_output, nn_index, nn_count, bin_index)
    return grad_input, grad_filter, None, None, None


_depthwise_conv3d.register_autograd(_conv_backward, setup_context=_conv_setup)


class _DepthwiseConv3dFn(torch.autograd.Function):
    """Eager fast path of the public API: same implementation functions as the registered custom ops, without the
    dispatcher round trip (the step issues ~230 op calls; the torch.library path costs ~4x more host time each)."""

    @staticmethod
    def forward(ctx, input, filter, nn_index, nn_count, bin_index):
        ctx.save_for_backward(input, filter, nn_index, nn_count, bin_index)
        return _depthwise_conv3d_impl(input, filter, nn_index, nn_count, bin_index)

    @staticmethod
    def backward(ctx, grad_output):
        input, filter, nn_index, nn_count, bin_index = ctx.saved_tensors
        gi, gf = _depthwise_conv3d_grad_impl(input, filter, grad_output, nn_index, nn_count, bin_index)
        return gi, gf, None, None, None


def depthwise_conv3d(input, filter, nn_index, nn_count, bin_index):
    """Depthwise half of the separable spherical convolution (public signature of tf_conv3d.py:9-20).

    input      [B, N, C]  fp32  features of the N graph nodes
    filter     [F, C, r]  fp32  one weight per (kernel bin, channel, multiplier); F = n*p*q + 1, bin 0 = the centre
    nn_index   [B, M, K]  int32 neighbours of each of the M output nodes (first nn_count entries valid)
    nn_count   [B, M]     int32
    bin_index  [B, M, K]  int32 kernel bin of each neighbour (spherical_kernel / the cube search)
    returns    [B, M, C*r] fp32: out[b,m,c*r+j] = 1/nn_count[b,m] * sum_k input[b,nn_index[b,m,k],c] * filter[bin_index[b,m,k],c,j]

    Differentiable in `input` and `filter` (gradients through the transposed graph, see _tgraph.py).
    """
    return _DepthwiseConv3dFn.apply(input, filter, nn_index, nn_count, bin_index)


def depthwise_conv3d_grad(input, filter, grad_output, nn_index, nn_count, bin_index):
    return _depthwise_conv3d_grad_impl(input, filter, grad_output, nn_index, nn_count, bin_index)


# ---- the whole separable layer in one kernel, inference only (csrc/sepconv.hip) ---------------------------
def separable_fused_supported_dims(N, F, C, r, K, Cout):
    """shapes sph3d_separable_conv3d_fused covers, by dimensions"""
    return bool(_lib.lib().sph3d_separable_conv3d_fused_supported(int(N), int(F), int(C), int(r), int(K), int(Cout)))


def separable_fused_supported(input, filter, nn_index, num_out_channels):
    """shapes sph3d_separable_conv3d_fused covers: r in {1, 2}, C % 4 == 0 and C <= 128 or a multiple of 128, Cout <= 512 in
    multiples of 16 (every separable layer of the S3DIS / ShapeNet plans)"""
    if not (input.is_cuda and input.dim() == 3 and filter.dim() == 3 and nn_index.dim() == 3):
        return False
    return bool(_lib.lib().sph3d_separable_conv3d_fused_supported(input.shape[1], filter.shape[0], input.shape[2],
                                                                  filter.shape[2], nn_index.shape[2], int(num_out_channels)))


def _separable_conv3d_fused_impl(input: torch.Tensor, filter: torch.Tensor, weights: torch.Tensor, bias: torch.Tensor,
                                 scale: torch.Tensor, shift: torch.Tensor, act: int, nn_index: torch.Tensor,
                                 nn_count: torch.Tensor, bin_index: torch.Tensor) -> torch.Tensor:
    _lib.require_device(input, filter, weights, nn_index, nn_count, bin_index)
    _check_conv(input, filter, nn_index, nn_count, bin_index)
    input, filter, weights = _lib.f32(input), _lib.f32(filter), _lib.f32(weights)
    nn_index, nn_count, bin_index = _lib.i32(nn_index), _lib.i32(nn_count), _lib.i32(bin_index)
    B, N, C = input.shape
    F, _, r = filter.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    Cout = weights.shape[1]
    if weights.shape[0] != C * r:
        raise ValueError("pointwise weights should be [C*r, Cout]")
    opt = [None if t is None or t.numel() == 0 else _lib.f32(t) for t in (bias, scale, shift)]
    out = torch.empty((B, M, Cout), dtype=torch.float32, device=input.device)
    _lib.check(_lib.lib().sph3d_separable_conv3d_fused(B, N, M, F, C, r, K, Cout, int(act), _lib.ptr(nn_index),
                                                       _lib.ptr(nn_count), _lib.ptr(bin_index), _lib.ptr(input),
                                                       _lib.ptr(filter), _lib.ptr(weights), _lib.ptr(opt[0]), _lib.ptr(opt[1]),
                                                       _lib.ptr(opt[2]), _lib.ptr(out), _lib.stream_ptr()))
    return out


_separable_conv3d_fused = torch.library.custom_op("sph3d::separable_conv3d_fused", mutates_args=())(_separable_conv3d_fused_impl)


@_separable_conv3d_fused.register_fake
def _(input, filter, weights, bias, scale, shift, act, nn_index, nn_count, bin_index):
    return input.new_empty((input.shape[0], nn_index.shape[1], weights.shape[1]))


def separable_conv3d_fused(input, filter, weights, nn_index, nn_count, bin_index, bias=None, elu=True, scale=None, shift=None):
    """elu?(depthwise_conv3d(input, filter) @ weights + bias) * scale + shift  -> [B, M, Cout], no gradient (inference):
    the depthwise tensor stays in LDS.  bias / scale / shift: [Cout] or None."""
    e = input.new_empty(0)
    with torch.no_grad():
        return _separable_conv3d_fused_impl(input, filter, weights, e if bias is None else bias, e if scale is None else scale,
                                            e if shift is None else shift, 1 if elu else 0, nn_index, nn_count, bin_index)


# ---- the separable layer in TRAINING mode as one kernel + the ELU/BN apply (csrc/sepring.hip) --------------------
def separable_train_supported(input, filter, nn_index, num_out_channels):
    """shapes sph3d_separable_conv3d_train covers: r in {1, 2}, C % 4 == 0, C <= 128, C*r <= 256, Cout in {16 .. 256} a power of two"""
    if not (input.is_cuda and input.dim() == 3 and filter.dim() == 3 and nn_index.dim() == 3):
        return False
    return bool(_lib.lib().sph3d_separable_conv3d_train_supported(input.shape[1], filter.shape[0], input.shape[2],
                                                                  filter.shape[2], nn_index.shape[2], int(num_out_channels)))


def _separable_conv3d_train_impl(input, filter, weights, bias, nn_index, nn_count, bin_index):
    """-> depthwise [B, M, C*r], y [B, M, Cout] = depthwise @ weights (+ bias), partial [nblk, 2, Cout] (sum elu(y), sum elu(y)^2)"""
    _lib.require_device(input, filter, weights, nn_index, nn_count, bin_index)
    _check_conv(input, filter, nn_index, nn_count, bin_index)
    input, filter, weights = _lib.f32(input), _lib.f32(filter), _lib.f32(weights)
    nn_index, nn_count, bin_index = _lib.i32(nn_index), _lib.i32(nn_count), _lib.i32(bin_index)
    B, N, C = input.shape
    F, _, r = filter.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    Cout = weights.shape[1]
    if weights.shape[0] != C * r:
        raise ValueError("pointwise weights should be [C*r, Cout]")
    l = _lib.lib()
    nblk = l.sph3d_separable_conv3d_train_blocks(Cout)
    dw = torch.empty((B, M, C * r), dtype=torch.float32, device=input.device)
    y = torch.empty((B, M, Cout), dtype=torch.float32, device=input.device)
    partial = torch.empty((nblk, 2, Cout), dtype=torch.float32, device=input.device)
    _lib.check(l.sph3d_separable_conv3d_train(B, N, M, F, C, r, K, Cout, _lib.ptr(nn_index), _lib.ptr(nn_count), _lib.ptr(bin_index),
                                              _lib.ptr(input), _lib.ptr(filter), _lib.ptr(weights),
                                              _lib.ptr(None if bias is None else _lib.f32(bias)), _lib.ptr(dw), _lib.ptr(y),
                                              _lib.ptr(partial), _lib.stream_ptr()))
    return dw, y, partial


class _SeparableTrainFn(torch.autograd.Function):
    """batch_norm(elu(depthwise_conv3d(input, filter) @ weights + bias)) with the batch's statistics: forward = the fused kernel
    + the statistics finalize / apply; backward = exactly the separate ops' backward passes (ELU+BN gradient, the product's two
    gradients, the depthwise gradients over the transposed graph)."""

    @staticmethod
    def forward(ctx, input, filter, weights, bias, gamma, beta, moving_mean, moving_var, nn_index, nn_count, bin_index):
        from . import tf_norm
        dw, y, partial = _separable_conv3d_train_impl(input, filter, weights, bias, nn_index, nn_count, bin_index)
        Cout = y.shape[-1]
        out, save_mean, save_rstd = tf_norm._elu_bn_partials_impl(y.view(-1, Cout), partial, gamma, beta, moving_mean, moving_var)
        ctx.save_for_backward(input, filter, weights, dw, y, gamma, save_mean, save_rstd, nn_index, nn_count, bin_index)
        ctx.has_bias = bias is not None
        return out.view(y.shape)

    @staticmethod
    def backward(ctx, dout):
        from . import tf_gemm, tf_norm
        input, filter, weights, dw, y, gamma, save_mean, save_rstd, nn_index, nn_count, bin_index = ctx.saved_tensors
        Cout = y.shape[-1]
        dy, dgamma, dbeta = tf_norm._bwd_impl(y.view(-1, Cout), dout.reshape(-1, Cout), gamma, save_mean, save_rstd, True)
        ddw = tf_gemm._pointwise_gemm_impl(dy, weights, True).view(dw.shape)
        dweights = tf_gemm._pointwise_gemm_tn_impl(dw.view(-1, dw.shape[-1]), dy) if ctx.needs_input_grad[2] else None
        dbias = dy.sum(0) if (ctx.has_bias and ctx.needs_input_grad[3]) else None
        gi, gf = _depthwise_conv3d_grad_impl(input, filter, ddw, nn_index, nn_count, bin_index)
        return gi, gf, dweights, dbias, dgamma, dbeta, None, None, None, None, None


def separable_conv3d_elu_bn_train(input, filter, weights, gamma, beta, moving_mean, moving_var, nn_index, nn_count, bin_index,
                                  bias=None):
    """training-mode separable layer with the ELU -> batch-norm tail: [B, M, Cout]; updates the moving statistics"""
    return _SeparableTrainFn.apply(input, filter, weights, bias, gamma, beta, moving_mean, moving_var, nn_index, nn_count, bin_index)


# ---- the depthwise convolution over a channel concatenation [a | b] that is never materialised (a decoder level's input) ------
def concat_supported(input_a, input_b, filter):
    if not (input_a.is_cuda and input_b.is_cuda and input_a.dim() == 3 and input_a.shape[:2] == input_b.shape[:2] and filter.dim() == 3):
        return False
    return bool(_lib.lib().sph3d_depthwise_conv3d_cat_supported(filter.shape[0], input_a.shape[2], input_b.shape[2], filter.shape[2]))


def _depthwise_conv3d_cat_impl(input_a, input_b, filter, nn_index, nn_count, bin_index):
    _lib.require_device(input_a, input_b, filter, nn_index, nn_count, bin_index)
    input_a, input_b, filter = _lib.f32(input_a), _lib.f32(input_b), _lib.f32(filter)
    nn_index, nn_count, bin_index = _lib.i32(nn_index), _lib.i32(nn_count), _lib.i32(bin_index)
    B, N, Ca = input_a.shape
    Cb = input_b.shape[2]
    F, Cf, r = filter.shape
    if Cf != Ca + Cb:
        raise ValueError("Input Channel Size error!")
    M, K = nn_index.shape[1], nn_index.shape[2]
    output = torch.empty((B, M, (Ca + Cb) * r), dtype=torch.float32, device=input_a.device)
    _lib.check(_lib.lib().sph3d_depthwise_conv3d_cat(B, N, M, F, Ca, Cb, r, K, _lib.ptr(nn_index), _lib.ptr(nn_count),
                                                     _lib.ptr(bin_index), _lib.ptr(input_a), _lib.ptr(input_b), _lib.ptr(filter),
                                                     _lib.ptr(output), _lib.stream_ptr()))
    return output


def _depthwise_conv3d_cat_grad_impl(input_a, input_b, filter, grad_output, nn_index, nn_count, bin_index):
    input_a, input_b, filter, grad_output = _lib.f32(input_a), _lib.f32(input_b), _lib.f32(filter), _lib.f32(grad_output)
    nn_index, nn_count, bin_index = _lib.i32(nn_index), _lib.i32(nn_count), _lib.i32(bin_index)
    B, N, Ca = input_a.shape
    Cb = input_b.shape[2]
    F, _, r = filter.shape
    M = nn_index.shape[1]
    grad_a, grad_b, grad_filter = torch.empty_like(input_a), torch.empty_like(input_b), torch.empty_like(filter)
    offsets, ent_key, ent_scale, active = _tgraph.transpose(nn_index, nn_count, N, bin_index=bin_index, num_bins=F)
    l = _lib.lib()
    wsb = l.sph3d_depthwise_conv3d_grad_t_workspace(B, N, F, Ca + Cb, r)
    ws = _lib.scratch(wsb, input_a.device)
    _lib.check(l.sph3d_depthwise_conv3d_grad_t_cat(
        B, N, M, F, Ca, Cb, r, _lib.ptr(offsets), _lib.ptr(ent_key), _lib.ptr(ent_scale), _lib.ptr(_tgraph.source_order(nn_index)),
        _lib.ptr(active), _lib.ptr(input_a), _lib.ptr(input_b), _lib.ptr(filter), _lib.ptr(grad_output), _lib.ptr(grad_a),
        _lib.ptr(grad_b), _lib.ptr(grad_filter), _lib.ptr(ws), wsb, _lib.stream_ptr()))
    return grad_a, grad_b, grad_filter


class _DepthwiseConv3dCatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input_a, input_b, filter, nn_index, nn_count, bin_index):
        ctx.save_for_backward(input_a, input_b, filter, nn_index, nn_count, bin_index)
        return _depthwise_conv3d_cat_impl(input_a, input_b, filter, nn_index, nn_count, bin_index)

    @staticmethod
    def backward(ctx, grad_output):
        input_a, input_b, filter, nn_index, nn_count, bin_index = ctx.saved_tensors
        ga, gb, gf = _depthwise_conv3d_cat_grad_impl(input_a, input_b, filter, grad_output, nn_index, nn_count, bin_index)
        return ga, gb, gf, None, None, None


def depthwise_conv3d_concat(input_a, input_b, filter, nn_index, nn_count, bin_index):
    """depthwise_conv3d(torch.cat((input_a, input_b), 2), filter, ...) without the concatenation (and without slicing its gradient):
    the kernels read / write the two tensors in place.  Shapes outside concat_supported(): concatenate, then the plain op."""
    if concat_supported(input_a, input_b, filter):
        return _DepthwiseConv3dCatFn.apply(input_a, input_b, filter, nn_index, nn_count, bin_index)
    return depthwise_conv3d(torch.cat((input_a, input_b), dim=2), filter, nn_index, nn_count, bin_index)
