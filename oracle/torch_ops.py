"""Oracle-backed torch ops on CPU tensors, with the op-level signatures of the reference's tf_ops/*/tf_*.py.

TEST INFRASTRUCTURE ONLY.  Used by (a) tests that run the s3g_util glue / model call pattern on CPU
("BASELINE config #1: plumbing, no GPU") and (b) the ``cpu_baseline`` leg of bench.py, which times the
same harness with these ops swapped in via ``patched_util()``.  The product package never imports this.
"""
import contextlib

import numpy as np
import torch

import oracle


def _np(t):
    return t.detach().cpu().numpy()


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def build_sphere_neighbor(database, query, radius=0.1, dilation_rate=None, nnsample=100):
    i, c, d = oracle.build_sphere_neighbor(_np(database), _np(query), radius, dilation_rate, nnsample)
    return _t(i), _t(c), _t(d)


def build_cube_neighbor(database, query, length=0.1, dilation_rate=None, nnsample=100, gridsize=3):
    i, c = oracle.build_cube_neighbor(_np(database), _np(query), length, dilation_rate, nnsample, gridsize)
    return _t(i), _t(c)


def spherical_kernel(database, query, nn_index, nn_count, nn_dist, radius, kernel=[8, 2, 3]):
    return _t(oracle.spherical_kernel(_np(database), _np(query), _np(nn_index), _np(nn_count), _np(nn_dist), radius,
                                      kernel))


def farthest_point_sample(neursize, database):
    return _t(oracle.farthest_point_sample(neursize, _np(database)))


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, filter, nn_index, nn_count, bin_index):
        ctx.save_for_backward(input, filter, nn_index, nn_count, bin_index)
        return _t(oracle.depthwise_conv3d(_np(input), _np(filter), _np(nn_index), _np(nn_count), _np(bin_index)))

    @staticmethod
    def backward(ctx, go):
        input, filter, nn_index, nn_count, bin_index = ctx.saved_tensors
        gi, gf = oracle.depthwise_conv3d_grad(_np(input), _np(filter), _np(go), _np(nn_index), _np(nn_count),
                                              _np(bin_index))
        return _t(gi), _t(gf), None, None, None


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, nn_index, nn_count):
        out, mi = oracle.max_pool3d(_np(input), _np(nn_index), _np(nn_count))
        mi = _t(mi)
        ctx.save_for_backward(input, mi)
        ctx.mark_non_differentiable(mi)
        return _t(out), mi

    @staticmethod
    def backward(ctx, go, _gmi):
        input, mi = ctx.saved_tensors
        return _t(oracle.max_pool3d_grad(_np(input), _np(go), _np(mi))), None, None


class _AvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, nn_index, nn_count):
        ctx.save_for_backward(input, nn_index, nn_count)
        return _t(oracle.avg_pool3d(_np(input), _np(nn_index), _np(nn_count)))

    @staticmethod
    def backward(ctx, go):
        input, nn_index, nn_count = ctx.saved_tensors
        return _t(oracle.avg_pool3d_grad(_np(input), _np(go), _np(nn_index), _np(nn_count))), None, None


class _Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, nn_index, nn_count):
        ctx.save_for_backward(input, nn_index, nn_count)
        return _t(oracle.mean_interpolate(_np(input), _np(nn_index), _np(nn_count)))

    @staticmethod
    def backward(ctx, go):
        input, nn_index, nn_count = ctx.saved_tensors
        return _t(oracle.mean_interpolate_grad(_np(input), _np(go), _np(nn_index), _np(nn_count))), None, None


class _Weighted(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, weight, nn_index, nn_count):
        ctx.save_for_backward(input, weight, nn_index, nn_count)
        return _t(oracle.weighted_interpolate(_np(input), _np(weight), _np(nn_index), _np(nn_count)))

    @staticmethod
    def backward(ctx, go):
        input, weight, nn_index, nn_count = ctx.saved_tensors
        return (_t(oracle.weighted_interpolate_grad(_np(input), _np(go), _np(weight), _np(nn_index), _np(nn_count))),
                None, None, None)


class tf_conv3d:          # namespaces mirroring the op modules
    depthwise_conv3d = staticmethod(lambda *a: _Conv.apply(*a))


class tf_pool3d:
    max_pool3d = staticmethod(lambda *a: _MaxPool.apply(*a))
    avg_pool3d = staticmethod(lambda *a: _AvgPool.apply(*a))


class tf_unpool3d:
    mean_interpolate = staticmethod(lambda *a: _Mean.apply(*a))
    weighted_interpolate = staticmethod(lambda *a: _Weighted.apply(*a))


class tf_gemm:
    matmul = staticmethod(torch.matmul)


@contextlib.contextmanager
def patched_util():
    """Temporarily point sph3d_gcn_amd.sph3gcn_util at the oracle ops (CPU tensors).  bench.py's cpu_baseline
    leg and the CPU plumbing tests only."""
    from sph3d_gcn_amd import sph3gcn_util as u
    saved = dict(tf_conv3d=u.tf_conv3d, tf_pool3d=u.tf_pool3d, tf_unpool3d=u.tf_unpool3d, neighbor_fn=u.neighbor_fn,
                 farthest_point_sample=u.farthest_point_sample, spherical_kernel=u.spherical_kernel,
                 tf_gemm=u.tf_gemm)
    u.tf_conv3d, u.tf_pool3d, u.tf_unpool3d = tf_conv3d, tf_pool3d, tf_unpool3d
    u.neighbor_fn = build_sphere_neighbor
    u.farthest_point_sample = farthest_point_sample
    u.spherical_kernel = spherical_kernel
    u.tf_gemm = tf_gemm                    # CPU GEMMs through torch.matmul (MKL/oneDNN): baseline not handicapped
    try:
        yield u
    finally:
        u.tf_conv3d, u.tf_pool3d, u.tf_unpool3d = saved["tf_conv3d"], saved["tf_pool3d"], saved["tf_unpool3d"]
        u.neighbor_fn = saved["neighbor_fn"]
        u.farthest_point_sample = saved["farthest_point_sample"]
        u.spherical_kernel = saved["spherical_kernel"]
        u.tf_gemm = saved["tf_gemm"]
