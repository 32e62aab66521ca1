"""Op-composition glue with the public signatures of the reference's utils/sph3gcn_util.py.

Every function keeps the reference's name, positional order, keyword names and defaults
(utils/sph3gcn_util.py:20,28,52,88,166,225,276,300,328) so that a model graph written against
``s3g_util`` calls it unchanged; the bodies are re-hosted on PyTorch-ROCm and the custom ops run
libsph3d's HIP kernels.  TF1-isms are mapped as follows:

  * ``scope`` / ``tf.get_variable``  -> a VariableStore (an nn.Module holding a ParameterDict):
    the first call under a scope creates the variables, later calls reuse them.  The active
    store is set with ``with variable_store(store):`` (a default global store exists, like
    TF's default graph).  ``reuse`` is accepted and ignored.
  * ``is_training``                   -> Python bool (None = True).
  * ``tf.add_to_collection('losses', ...)`` / regularisation losses -> ``store.collect_losses()``.
"""
import contextlib
import math
import os

import torch
import torch.nn.functional as F

from . import tf_conv3d, tf_pool3d, tf_unpool3d
from .tf_nnquery import build_sphere_neighbor, build_cube_neighbor
from .tf_sample import farthest_point_sample, inverse_density_sample, random_sample
from .tf_buildkernel import spherical_kernel
from . import tf_gemm, tf_norm, _lib

neighbor_fn = build_sphere_neighbor  # default nn search method

elu = F.elu


# --------------------------------------------------------------------------------------
# variable scopes
# --------------------------------------------------------------------------------------
class VariableStore(torch.nn.Module):
    """Named variables created on first use (the role of tf.get_variable + variable_scope)."""

    def __init__(self, device=None, seed=None):
        super().__init__()
        self.params = torch.nn.ParameterDict()
        self._device = device
        self._gen = None
        if seed is not None:
            self._gen = torch.Generator(device="cpu")
            self._gen.manual_seed(seed)
        self._decay = []       # (name, coefficient): coefficient * l2_loss(var)   ('losses' collection)
        self._reg = []         # names under tf.GraphKeys.REGULARIZATION_LOSSES (BN beta/gamma, scale 1.0)

    def _key(self, name):
        return name.replace(".", "_")

    def has(self, name):
        return self._key(name) in self.params

    def get_variable(self, name, shape, init_fn, trainable=True):
        key = self._key(name)
        if key in self.params:
            return self.params[key]
        t = torch.empty(*shape, dtype=torch.float32)
        init_fn(t, self._gen)
        if self._device is not None:
            t = t.to(self._device)
        p = torch.nn.Parameter(t, requires_grad=trainable)
        self.params[key] = p
        return p

    def get_buffer(self, name, shape, value):
        key = self._key(name)
        if not hasattr(self, "_buf_" + key):
            t = torch.full(shape, float(value), dtype=torch.float32)
            if self._device is not None:
                t = t.to(self._device)
            self.register_buffer("_buf_" + key, t)
        return getattr(self, "_buf_" + key)

    def collect_losses(self):
        """Sum of the 'losses' collection entries created by weight_decay (utils/sph3gcn_util.py:82-84)."""
        if not self._decay:
            return None
        return l2_sum([self.params[self._key(name)] for name, _c in self._decay], [c for _n, c in self._decay])

    def regularization_loss(self):
        """tf.losses.get_regularization_loss(): sum of l2_loss over BN beta/gamma (:330-331)."""
        if not self._reg:
            return None
        return l2_sum([self.params[self._key(name)] for name in self._reg], [1.0] * len(self._reg))


# sum_i coef_i * tf.nn.l2_loss(p_i) over a list of parameters as a handful of multi-tensor kernels instead of five launches per
# parameter and as many in the backward pass (the ModelNet step: ~200 launches of 2-5 us on the stream that carries the
# feature path).  ||p||^2 comes from the multi-tensor 2-norm: within an ulp or two of sum(p * p).
_coef_cache = {}


def l2_sum(params, coefs):
    """sum_i coefs[i] * 0.5 * sum(params[i] ** 2), differentiable in the parameters"""
    key = (tuple(float(c) for c in coefs), params[0].device)
    c = _coef_cache.get(key)
    if c is None:
        c = _coef_cache[key] = torch.tensor(key[0], dtype=torch.float32, device=params[0].device)
    return _L2SumList.apply(c, key[0], *params)


class _L2SumList(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coef_t, coef_list, *params):
        ctx.coef_list = list(coef_list)
        ctx.save_for_backward(*params)
        sq = torch.stack(torch._foreach_norm([p.detach() for p in params])).square()      # ||p_i||^2
        return 0.5 * (sq * coef_t).sum()

    @staticmethod
    def backward(ctx, g):
        grads = torch._foreach_mul(list(ctx.saved_tensors), ctx.coef_list)                 # coef_i * p_i: one multi-tensor launch
        grads = torch._foreach_mul(grads, g)
        return (None, None) + tuple(grads)


_default_store = VariableStore()
_active = [_default_store]


def get_variable_store():
    return _active[-1]


@contextlib.contextmanager
def variable_store(store):
    _active.append(store)
    try:
        yield store
    finally:
        _active.pop()


def _xavier_uniform(t, gen):
    # tf.contrib.layers.xavier_initializer(): uniform, fan_in = prod(shape[:-2])*shape[-2], fan_out = ...*shape[-1]
    shape = t.shape
    receptive = 1
    for s in shape[:-2]:
        receptive *= s
    fan_in = shape[-2] * receptive if len(shape) > 1 else shape[0]
    fan_out = shape[-1] * receptive
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    with torch.no_grad():
        t.uniform_(-limit, limit, generator=gen)


def _truncated_normal(stddev):
    def init(t, gen):
        with torch.no_grad():
            torch.nn.init.trunc_normal_(t, mean=0.0, std=stddev, a=-2 * stddev, b=2 * stddev, generator=gen)
    return init


def _constant(value):
    def init(t, gen):
        with torch.no_grad():
            t.fill_(value)
    return init


def _variable_with_weight_decay(name, shape, stddev, with_decay, use_xavier=True):
    """utils/sph3gcn_util.py:61-85"""
    store = get_variable_store()
    new = not store.has(name)
    var = store.get_variable(name, shape, _xavier_uniform if use_xavier else _truncated_normal(stddev))
    if new and with_decay is not None:
        store._decay.append((name, with_decay))
    return var


# --------------------------------------------------------------------------------------
# graph builders (utils/sph3gcn_util.py:20-58)
# --------------------------------------------------------------------------------------
def build_global_graph(xyz, query, radius):
    nn_uplimit = xyz.shape[1]
    nn_idx, nn_cnt, nn_dst = neighbor_fn(xyz, query, radius=radius, nnsample=nn_uplimit)
    return nn_idx, nn_cnt, nn_dst


def build_graph(xyz, radius, nn_uplimit, num_sample, sample_method=None):
    intra_idx, intra_cnt, intra_dst = neighbor_fn(xyz, xyz, radius=radius, nnsample=nn_uplimit)

    if num_sample is not None:
        if sample_method == 'random':
            sample_index = random_sample(num_sample, xyz)
        elif sample_method == 'FPS':
            sample_index = farthest_point_sample(num_sample, xyz)
        elif sample_method == 'IDS':
            prob = torch.sum(intra_dst, dim=-1) / intra_cnt.float()
            sample_index = inverse_density_sample(num_sample, prob)
        else:
            raise ValueError('Unknown sampling method.')

        batch_size = xyz.shape[0]
        batch_indices = torch.arange(batch_size, dtype=torch.int32, device=xyz.device).view(-1, 1, 1)
        batch_indices = batch_indices.expand(batch_size, num_sample, 1)
        indices = torch.cat([batch_indices, sample_index.unsqueeze(2).to(torch.int32)], dim=2)
    else:
        indices = None

    return intra_idx, intra_cnt, intra_dst, indices


def build_graph_deconv(xyz, xyz_unpool, radius, nn_uplimit):
    intra_idx, intra_cnt, intra_dst = neighbor_fn(xyz, xyz, radius=radius, nnsample=nn_uplimit)
    inter_idx, inter_cnt, inter_dst = neighbor_fn(xyz, xyz_unpool, radius=radius, nnsample=nn_uplimit)
    return intra_idx, intra_cnt, intra_dst, inter_idx, inter_cnt, inter_dst


def build_intra_graph(xyz, radius, nn_uplimit, kernel, with_transpose=True):
    """Intra-level graph + spherical-kernel bins of one point set: what the model graphs obtain from
    ``neighbor_fn(xyz, xyz, ...)`` followed by ``spherical_kernel(xyz, xyz, ...)`` (models/SPH3D_s3dis.py:56-62), from ONE
    fused kernel on the HIP device (tf_nnquery.build_sphere_graph); on other tensors (the CPU-oracle-backed tests) the two
    calls are made one after the other.  -> nn_idx, nn_cnt, nn_dst, filt_idx"""
    from . import tf_nnquery
    if xyz.is_cuda and neighbor_fn is build_sphere_neighbor and tf_nnquery.get_radius_mode() == "compat":
        # (with_transpose: also the counting pass of the transposed graph the convolution GRADIENTS gather over)
        return tf_nnquery.build_sphere_graph(xyz, radius, nn_uplimit, kernel, with_transpose=with_transpose)
    idx, cnt, dst = neighbor_fn(xyz, xyz, radius=radius, nnsample=nn_uplimit)
    return idx, cnt, dst, spherical_kernel(xyz, xyz, idx, cnt, dst, radius, kernel=kernel)


def gather_nd(params, indices):
    """tf.gather_nd for the [B, S, 2] (batch, point) index pairs build_graph returns
    (used by the model graphs at models/SPH3D_s3dis.py:68-72)."""
    if (params.is_cuda and params.dim() >= 2 and params.element_size() == 4 and indices.dtype == torch.int32
            and indices.shape[-1] == 2 and not params.requires_grad):
        # one HIP launch (sph3d_gather_nd) instead of two int64 conversions + an advanced-indexing kernel
        params = params.contiguous()
        indices = indices.contiguous()
        B, N = params.shape[0], params.shape[1]
        row = 1
        for d in params.shape[2:]:
            row *= d
        S = indices.numel() // 2
        out = _lib.empty(tuple(indices.shape[:-1]) + tuple(params.shape[2:]), params.dtype, params.device)
        if S and row:
            _lib.check(_lib.lib().sph3d_gather_nd(B, N, S, row, _lib.ptr(indices), _lib.ptr(params), _lib.ptr(out),
                                                 _lib.stream_ptr()))
        return out
    b = indices[..., 0].long()
    p = indices[..., 1].long()
    return params[b, p]


def gather_pooling_graph(nn_index, nn_count, indices, with_transpose=True):
    """(gather_nd(nn_index, indices), gather_nd(nn_count, indices)) — the pooling graph of models/SPH3D_s3dis.py:68-72 — as ONE
    launch on the HIP device which (with_transpose) also counts the in-edges of the pooling graph's transpose, finished and cached
    here for the max-pool gradient (tf_pool3d).  Same two tensors as the two gather_nd calls."""
    if not (nn_index.is_cuda and nn_index.dim() == 3 and nn_count.dim() == 2 and indices.dtype == torch.int32 and indices.shape[-1] == 2
            and indices.shape[0] == nn_index.shape[0]):
        return gather_nd(nn_index, indices), gather_nd(nn_count, indices)
    from . import _tgraph
    nn_index, nn_count, indices = nn_index.contiguous(), nn_count.contiguous(), indices.contiguous()
    B, N, K = nn_index.shape
    S = indices.shape[1]
    dev = nn_index.device
    out_idx = _lib.empty((B, S, K), torch.int32, dev)
    out_cnt = _lib.empty((B, S), torch.int32, dev)
    l = _lib.lib()
    ws, wsb = None, 0
    if with_transpose:
        wsb = l.sph3d_graph_transpose_workspace(B, N, S, K, 1)
        ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=dev)
    _lib.check(l.sph3d_gather_rows_count(B, N, S, K, _lib.ptr(indices), _lib.ptr(nn_index), _lib.ptr(nn_count), _lib.ptr(out_idx),
                                         _lib.ptr(out_cnt), _lib.ptr(ws), wsb, _lib.stream_ptr()))
    if with_transpose:
        # (rows gathered from the ball query's rows: ascending, distinct neighbour ids)
        _tgraph.transpose(out_idx, out_cnt, N, counted_workspace=ws, unique_rows=True)
    return out_idx, out_cnt


# --------------------------------------------------------------------------------------
# layers (utils/sph3gcn_util.py:88-273)
# --------------------------------------------------------------------------------------
def _finish(outputs, num_out_channels, scope, activation_fn, with_bn, with_bias, reuse, is_training, fused_bias=None):
    if with_bias and fused_bias is None:
        biases = get_variable_store().get_variable(scope + '/biases', [num_out_channels], _constant(0.0))
        outputs = outputs + biases
    if (with_bn and activation_fn is elu and FUSE_ELU_BN and outputs.is_cuda and tf_norm.supported(outputs.shape[-1])):
        # the reference's tail "ELU -> batch norm" (util.py:155-161) as ONE fused op on the HIP path
        return _elu_batch_normalization(outputs, is_training, name=scope + '/bn')
    if activation_fn is not None:
        outputs = activation_fn(outputs)
    if with_bn:
        outputs = batch_normalization(outputs, is_training, name=scope + '/bn', reuse=reuse)
    return outputs


def _fused_tail_applies(x2d, num_out_channels, activation_fn, with_bn, with_bias, is_training):
    """GEMM (+ bias) + ELU + batch norm with the statistics in the GEMM's epilogue (tf_norm.gemm_elu_batch_norm): training
    mode, shapes the statistics kernel covers"""
    training = True if is_training is None else bool(is_training)
    return (FUSE_GEMM_BN and FUSE_ELU_BN and with_bn and activation_fn is elu and training and x2d.is_cuda
            and tf_norm.supported(num_out_channels)
            and tf_norm.gemm_bn_blocks(x2d.shape[0], x2d.shape[1], num_out_channels) > 0)


def _gemm_tail(x2d, kernel, out_shape, num_out_channels, scope, activation_fn, with_bn, with_bias, reuse, is_training):
    """tf.matmul -> (+ biases) -> activation -> batch norm of the three layer kinds (utils/sph3gcn_util.py:146-161,204-222,260-273)"""
    if _fused_tail_applies(x2d, num_out_channels, activation_fn, with_bn, with_bias, is_training):
        store = get_variable_store()
        biases = store.get_variable(scope + '/biases', [num_out_channels], _constant(0.0)) if with_bias else None
        gamma, beta, moving_mean, moving_var = _bn_variables(store, scope + '/bn', num_out_channels)
        out = tf_norm.gemm_elu_batch_norm(x2d, kernel, gamma, beta, moving_mean, moving_var, bias=biases)
        return out.reshape(out_shape)
    if (FUSE_GEMM_BN and with_bias and not with_bn and x2d.is_cuda and (activation_fn is elu or activation_fn is None)):
        # biases (and ELU) in the GEMM's epilogue: the tail of a layer without batch norm (utils/sph3gcn_util.py:152-155)
        biases = get_variable_store().get_variable(scope + '/biases', [num_out_channels], _constant(0.0))
        out = tf_gemm.matmul_bias_act(x2d, kernel, biases, elu=activation_fn is elu)
        return out.reshape(out_shape)
    outputs = tf_gemm.matmul(x2d, kernel).reshape(out_shape)
    return _finish(outputs, num_out_channels, scope, activation_fn, with_bn, with_bias, reuse, is_training)


def separable_conv3d(inputs,
                     num_out_channels,
                     kernel_size,
                     depth_multiplier,
                     scope,
                     nn_index,
                     nn_count,
                     filt_index,
                     use_xavier=True,
                     stddev=1e-3,
                     weight_decay=None,
                     activation_fn=elu,
                     with_bn=False,
                     with_bias=False,
                     reuse=None,
                     is_training=None):
    """Separable spherical convolution layer (same signature as utils/sph3gcn_util.py:88-163): depthwise conv over the
    graph, pointwise GEMM to `num_output_channels`, then bias / activation / batch norm as the flags say."""
    # inputs may be a PAIR (a, b): the layer of the reference applied to tf.concat((a, b), axis=2) — same variables, same result —
    # with the depthwise kernels reading the two tensors in place (tf_conv3d.depthwise_conv3d_concat)
    pair = inputs if isinstance(inputs, (tuple, list)) else None
    infer_fused = (FUSE_SEPARABLE_INFERENCE and is_training is not None and not bool(is_training) and not torch.is_grad_enabled()
                   and (activation_fn is elu or activation_fn is None))
    if pair is not None and infer_fused and pair[0].is_cuda and getattr(tf_conv3d, "separable_fused_supported_dims", None) is not None \
            and tf_conv3d.separable_fused_supported_dims(pair[0].shape[1], kernel_size, pair[0].shape[-1] + pair[1].shape[-1],
                                                         depth_multiplier, nn_index.shape[2], num_out_channels) \
            and (FUSE_SEPARABLE_INFERENCE is True
                 or _fused_rows_pay(nn_index.shape[0] * nn_index.shape[1], pair[0].shape[-1] + pair[1].shape[-1], depth_multiplier,
                                    num_out_channels)):
        # inference: the one-kernel layer reads ONE input tensor; concatenating the pair (a copy of the layer's input) costs far
        # less than the depthwise tensor the fused kernel never writes
        inputs, pair = torch.cat(tuple(pair), dim=2), None
    if pair is not None:
        supported = getattr(tf_conv3d, "concat_supported", None)       # (the oracle-backed CPU stand-ins of the tests have none)
        if not (FUSE_CONV_CONCAT and pair[0].is_cuda and supported is not None and supported(pair[0], pair[1], torch.empty(
                (kernel_size, pair[0].shape[-1] + pair[1].shape[-1], depth_multiplier), device='meta'))):
            inputs, pair = torch.cat(tuple(pair), dim=2), None
    num_in_channels = inputs.shape[-1] if pair is None else pair[0].shape[-1] + pair[1].shape[-1]
    depthwise_kernel = _variable_with_weight_decay(scope + '/depthwise_weights',
                                                   shape=[kernel_size, num_in_channels, depth_multiplier],
                                                   use_xavier=use_xavier, stddev=stddev,
                                                   with_decay=weight_decay)
    if pair is not None:
        outputs = tf_conv3d.depthwise_conv3d_concat(pair[0], pair[1], depthwise_kernel, nn_index, nn_count, filt_index)
        batch_size = outputs.shape[0]
        num_in_channels = outputs.shape[-1]
        kernel = _variable_with_weight_decay(scope + '/weights', shape=[num_in_channels, num_out_channels],
                                             use_xavier=use_xavier, stddev=stddev, with_decay=weight_decay)
        return _gemm_tail(outputs.reshape(-1, num_in_channels), kernel, (batch_size, -1, num_out_channels), num_out_channels, scope,
                          activation_fn, with_bn, with_bias, reuse, is_training)
    if (infer_fused and tf_conv3d.separable_fused_supported(inputs, depthwise_kernel, nn_index, num_out_channels)
            and (FUSE_SEPARABLE_INFERENCE is True or _fused_layer_pays(inputs, depth_multiplier, nn_index, num_out_channels))):
        # inference: the whole layer in one kernel, the depthwise tensor never written (csrc/sepconv.hip)
        store = get_variable_store()
        kernel = _variable_with_weight_decay(scope + '/weights', shape=[num_in_channels * depth_multiplier, num_out_channels],
                                             use_xavier=use_xavier, stddev=stddev, with_decay=weight_decay)
        biases = store.get_variable(scope + '/biases', [num_out_channels], _constant(0.0)) if with_bias else None
        scale = shift = None
        if with_bn:
            gamma, beta, moving_mean, moving_var = _bn_variables(store, scope + '/bn', num_out_channels)
            scale = gamma * torch.rsqrt(moving_var + 1e-3)
            shift = beta - moving_mean * scale
        return tf_conv3d.separable_conv3d_fused(inputs, depthwise_kernel, kernel, nn_index, nn_count, filt_index, bias=biases,
                                                elu=activation_fn is elu, scale=scale, shift=shift)
    training = True if is_training is None else bool(is_training)
    if (FUSE_SEPARABLE_TRAINING and training and with_bn and activation_fn is elu and FUSE_ELU_BN and inputs.is_cuda
            and getattr(tf_conv3d, "separable_train_supported", None) is not None
            and tf_conv3d.separable_train_supported(inputs, depthwise_kernel, nn_index, num_out_channels)
            and (FUSE_SEPARABLE_TRAINING is True
                 or _fused_train_pays(nn_index.shape[0] * nn_index.shape[1], num_in_channels, depth_multiplier, num_out_channels))):
        # training: depthwise gather + pointwise product + statistics partials in one barrier-free kernel (csrc/sepring.hip);
        # the depthwise tensor is written for the weight gradient but not re-read; same variables, same order of creation
        store = get_variable_store()
        kernel = _variable_with_weight_decay(scope + '/weights', shape=[num_in_channels * depth_multiplier, num_out_channels],
                                             use_xavier=use_xavier, stddev=stddev, with_decay=weight_decay)
        biases = store.get_variable(scope + '/biases', [num_out_channels], _constant(0.0)) if with_bias else None
        gamma, beta, moving_mean, moving_var = _bn_variables(store, scope + '/bn', num_out_channels)
        return tf_conv3d.separable_conv3d_elu_bn_train(inputs, depthwise_kernel, kernel, gamma, beta, moving_mean, moving_var,
                                                       nn_index, nn_count, filt_index, bias=biases)
    outputs = tf_conv3d.depthwise_conv3d(inputs, depthwise_kernel, nn_index, nn_count, filt_index)

    batch_size = outputs.shape[0]
    num_in_channels = outputs.shape[-1]
    kernel = _variable_with_weight_decay(scope + '/weights', shape=[num_in_channels, num_out_channels],
                                         use_xavier=use_xavier, stddev=stddev, with_decay=weight_decay)
    # pointwise convolution as one GEMM over all points
    return _gemm_tail(outputs.reshape(-1, num_in_channels), kernel, (batch_size, -1, num_out_channels), num_out_channels, scope,
                      activation_fn, with_bn, with_bias, reuse, is_training)


def pointwise_conv3d(inputs,
                     num_out_channels,
                     scope,
                     use_xavier=True,
                     stddev=1e-3,
                     weight_decay=None,
                     activation_fn=elu,
                     with_bn=False,
                     with_bias=False,
                     reuse=None,
                     is_training=None):
    """1x1 layer over the points: GEMM + bias / activation / batch norm (same signature as utils/sph3gcn_util.py:166-222)."""
    batch_size = inputs.shape[0]
    num_in_channels = inputs.shape[-1]
    kernel = _variable_with_weight_decay(scope + '/weights', shape=[num_in_channels, num_out_channels],
                                         use_xavier=use_xavier, stddev=stddev, with_decay=weight_decay)
    return _gemm_tail(inputs.reshape(-1, num_in_channels), kernel, (batch_size, -1, num_out_channels), num_out_channels, scope,
                      activation_fn, with_bn, with_bias, reuse, is_training)


def pointwise_conv3d_concat(inputs_a, inputs_b, num_out_channels, scope, use_xavier=True, stddev=1e-3, weight_decay=None,
                            activation_fn=elu, with_bn=False, with_bias=False, reuse=None, is_training=None):
    """pointwise_conv3d(tf.concat((inputs_a, inputs_b), axis=2), ...) — same variables, same result — without the concatenation
    when the layer is a plain product with few outputs (the logits layer of models/SPH3D_s3dis.py:104-108: 256 -> num_cls):
    tf_gemm.linear_concat2 reads the two halves where they are."""
    ca, cb = inputs_a.shape[-1], inputs_b.shape[-1]
    if (FUSE_LOGITS_CONCAT and inputs_a.is_cuda and not with_bn and activation_fn is None
            and tf_gemm.skinny_supported(inputs_a.shape[0] * inputs_a.shape[1], ca, cb, num_out_channels)):
        kernel = _variable_with_weight_decay(scope + '/weights', shape=[ca + cb, num_out_channels],
                                             use_xavier=use_xavier, stddev=stddev, with_decay=weight_decay)
        biases = get_variable_store().get_variable(scope + '/biases', [num_out_channels], _constant(0.0)) if with_bias else None
        out = tf_gemm.linear_concat2(inputs_a.reshape(-1, ca), inputs_b.reshape(-1, cb), kernel, biases)
        return out.reshape(inputs_a.shape[0], -1, num_out_channels)
    return pointwise_conv3d(torch.cat((inputs_a, inputs_b), dim=2), num_out_channels, scope, use_xavier=use_xavier, stddev=stddev,
                            weight_decay=weight_decay, activation_fn=activation_fn, with_bn=with_bn, with_bias=with_bias,
                            reuse=reuse, is_training=is_training)


def fully_connected(inputs,
                    num_out_channels,
                    scope,
                    use_xavier=True,
                    stddev=1e-3,
                    weight_decay=None,
                    activation_fn=elu,
                    with_bn=False,
                    with_bias=False,
                    reuse=None,
                    is_training=None):
    """Dense layer on [B, C] features: GEMM + bias / activation / batch norm (same signature as utils/sph3gcn_util.py:225-273)."""
    num_in_channels = inputs.shape[-1]
    kernel = _variable_with_weight_decay(scope + '/weights', shape=[num_in_channels, num_out_channels],
                                         use_xavier=use_xavier, stddev=stddev, with_decay=weight_decay)
    return _gemm_tail(inputs, kernel, (inputs.shape[0], num_out_channels), num_out_channels, scope, activation_fn, with_bn,
                      with_bias, reuse, is_training)


def pool3d(inputs, nn_index, nn_count, scope, method):
    """Graph pooling onto the sampled points, method 'max' or 'avg' (same signature as utils/sph3gcn_util.py:276-297)."""
    if method == 'max':
        outputs, max_index = tf_pool3d.max_pool3d(inputs, nn_index, nn_count)
    elif method == 'avg':
        outputs = tf_pool3d.avg_pool3d(inputs, nn_index, nn_count)
    else:
        raise ValueError("Unknow pooling method %s." % method)
    return outputs


def pool3d_with_skip(inputs, nn_index, nn_count, scope, method):
    """pool3d(inputs, ...) for an `inputs` that a skip connection uses too: -> (pooled, skip).  `skip` is `inputs`; with max pooling
    on the GPU the two gradients of `inputs` are summed inside the pooling gradient's kernel (tf_pool3d.max_pool3d_with_skip)."""
    fn = getattr(tf_pool3d, "max_pool3d_with_skip", None)        # (the oracle-backed CPU stand-ins of the tests have none)
    if method == 'max' and FUSE_POOL_SKIP and inputs.is_cuda and fn is not None:
        outputs, _max_index, skip = fn(inputs, nn_index, nn_count)
        return outputs, skip
    return pool3d(inputs, nn_index, nn_count, scope, method), inputs


def unpool3d(inputs, nn_index, nn_count, nn_dist, scope, method):
    """Feature interpolation back onto the finer point set, method 'mean' or 'weighted' (same signature as
    utils/sph3gcn_util.py:300-325)."""
    if method == 'mean':
        outputs = tf_unpool3d.mean_interpolate(inputs, nn_index, nn_count)
    elif method == 'weighted':
        sum_nn_dist = torch.sum(nn_dist, dim=-1, keepdim=True)
        epsilon = 1e-7
        weight = (nn_dist + epsilon) / (sum_nn_dist + epsilon)
        outputs = tf_unpool3d.weighted_interpolate(inputs, weight, nn_index, nn_count)
    else:
        raise ValueError("Unknow unpooling method %s." % method)
    return outputs


FUSE_GEMM_BN = True    # the statistics of that tail from the GEMM's epilogue where the shape allows (tf_norm.gemm_elu_batch_norm)
FUSE_POOL_SKIP = True             # pool3d_with_skip: the skip connection's gradient is added inside the max-pool gradient kernel
FUSE_CONV_CONCAT = True           # separable_conv3d((a, b), ...): depthwise kernels over two inputs in place (tf_conv3d.depthwise_conv3d_concat)
FUSE_LOGITS_CONCAT = True         # pointwise_conv3d_concat: few-output layer over two operand halves (tf_gemm.linear_concat2)
# is_training=False under torch.no_grad(): separable_conv3d as ONE kernel (tf_conv3d.separable_conv3d_fused).
#   "auto" (default): where the one-kernel layer is the faster one (measured, tools/exp_sepconv_layers.py: with the pointwise
#                     weights resident in registers always; with W streamed per tile — wider layers — from 16 384 output points up:
#                     at 6144 x 1024 -> 512 the per-tile W reads cost more than the depthwise tensor's round trip saves);
#   True: wherever the kernel covers the shape;  False: never.
FUSE_SEPARABLE_INFERENCE = "auto"


# is_training=True: separable_conv3d + ELU + batch norm with the depthwise gather, the pointwise product and the statistics
# partials in ONE kernel (tf_conv3d.separable_conv3d_elu_bn_train; csrc/sepring.hip) where the kernel covers the shape.
#   False (default): never — measured, tools/exp_sepconv_training.py / profiles/r06_exp_sepconv_training.log: on the four layer
#       shapes of the S3DIS plan the kernel covers the one-kernel forward takes what the gather and the product take one after the
#       other (level 0: 300 vs 297 us at C = 128, 163 vs 146 at C = 64; level 1: 74 vs 77, 52 vs 51): with 16 register-limited
#       waves per CU a wave's MFMA block sits on the same dependent path as its gathers, and in training the depthwise tensor is
#       written anyway (the weight gradient's operand); headline 1911 vs 1918 blocks/s in three alternating pairs;
#   "auto": layers of at least _FUSED_TRAIN_MIN_ROWS output points;  True: wherever the kernel covers the shape.
FUSE_SEPARABLE_TRAINING = {"0": False, "1": True, "auto": "auto"}.get(os.environ.get("SPH3D_FUSE_TRAIN", ""), False)      # (env: A/B runs of bench.py)
_FUSED_TRAIN_MIN_ROWS = int(os.environ.get("SPH3D_FUSE_TRAIN_MIN_ROWS", "16384"))


def _fused_train_pays(rows, C, r, Cout):
    return rows >= _FUSED_TRAIN_MIN_ROWS


def _fused_rows_pay(rows, C, r, Cout):
    small = C <= 128 and C * r <= 256 and Cout <= 128          # sepconv_fused_kernel: W in registers
    return small or rows >= 16384


def _fused_layer_pays(inputs, depth_multiplier, nn_index, num_out_channels):
    return _fused_rows_pay(nn_index.shape[0] * nn_index.shape[1], inputs.shape[-1], depth_multiplier, num_out_channels)
FUSE_ELU_BN = True     # fused sph3d::elu_bn for the ELU -> BN tail (same variables / moving statistics as the unfused ops)


def _bn_variables(store, name, C):
    new = not store.has(name + '/gamma')
    gamma = store.get_variable(name + '/gamma', [C], _constant(1.0))
    beta = store.get_variable(name + '/beta', [C], _constant(0.0))
    if new:
        store._reg.extend([name + '/gamma', name + '/beta'])
    moving_mean = store.get_buffer(name + '/moving_mean', (C,), 0.0)
    moving_var = store.get_buffer(name + '/moving_variance', (C,), 1.0)
    return gamma, beta, moving_mean, moving_var


def _elu_batch_normalization(data, is_training, name):
    store = get_variable_store()
    gamma, beta, moving_mean, moving_var = _bn_variables(store, name, data.shape[-1])
    training = True if is_training is None else bool(is_training)
    return tf_norm.elu_batch_norm(data, gamma, beta, moving_mean, moving_var, training)


def batch_normalization(data, is_training, name, reuse=None):
    """tf.layers.batch_normalization(momentum=0.99, epsilon=1e-3 [TF default], axis=-1) with the
    l2 regularisers on beta/gamma (utils/sph3gcn_util.py:328-332).  Statistics over all but the last axis."""
    store = get_variable_store()
    C = data.shape[-1]
    gamma, beta, moving_mean, moving_var = _bn_variables(store, name, C)
    training = True if is_training is None else bool(is_training)
    flat = data.reshape(-1, C)
    if training:
        # batch statistics for the output; the moving statistics take the BIASED batch variance, as the non-fused
        # tf.layers path (tf.nn.moments) does — F.batch_norm would store the Bessel-corrected one
        out = F.batch_norm(flat, None, None, gamma, beta, training=True, momentum=0.0, eps=1e-3)
        with torch.no_grad():
            var, mean = torch.var_mean(flat, dim=0, unbiased=False)
            moving_mean.mul_(0.99).add_(mean, alpha=1.0 - 0.99)
            moving_var.mul_(0.99).add_(var, alpha=1.0 - 0.99)
    else:
        out = F.batch_norm(flat, moving_mean, moving_var, gamma, beta, training=False, momentum=0.0, eps=1e-3)
    return out.reshape(data.shape)
