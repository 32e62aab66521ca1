"""ref_gpu — the reference's OWN CUDA kernels, compiled unmodified for gfx950 and run on the MI355X.

TEST INFRASTRUCTURE ONLY (same rule as the rest of oracle/).

``oracle/Makefile`` target ``ref`` compiles /root/reference/tf_ops/*/tf_*_gpu.cu where they lie with
hipcc into ``oracle/_ref/libsph3d_ref_gfx950.so`` (git-ignored; it travels to the GPU box with the
snapshot).  This module calls the reference's launcher functions in that library — same
``<<<32,1024>>>`` launches, same kernels — on torch device tensors, after zero-filling the outputs as
each reference ``OpKernel::Compute`` did with cudaMemset.

It is how the CPU oracle is pinned to the reference itself: tests compare oracle vs ref_gpu vs
libsph3d on the GPU box, and ``tests/golden/make_golden.py`` stores ref_gpu outputs as golden vectors.

Caveats of running the reference as-is: its FPS kernel has a latent race (tf_sample_gpu.cu:68) that
real hardware does not trigger in practice; its backward kernels use fp32 atomics (order-dependent
sums); its atan2f is ROCm's ocml here (CUDA libdevice originally).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libsph3d_ref_gfx950.so")

_I, _F, _P = ctypes.c_int, ctypes.c_float, ctypes.c_void_p

# Itanium-mangled names of the reference launchers (nm -D on the built library)
_SYMS = {
    "sphere": ("_Z27buildSphereNeighborLauncheriiiifPKfS0_PiS1_Pf", [_I, _I, _I, _I, _F, _P, _P, _P, _P, _P]),
    "cube": ("_Z25buildCubeNeighborLauncheriiiiifPKfS0_PiS1_", [_I, _I, _I, _I, _I, _F, _P, _P, _P, _P]),
    "kernel": ("_Z23sphericalKernelLauncheriiiiiiifPKfS0_PKiS2_S0_Pi", [_I] * 7 + [_F] + [_P] * 6),
    "conv": ("_Z23depthwiseConv3dLauncheriiiiiiPKiS0_S0_PKfS2_Pf", [_I] * 6 + [_P] * 6),
    "conv_grad": ("_Z27depthwiseConv3dGradLauncheriiiiiiiPKiS0_S0_PKfS2_S2_PfS3_", [_I] * 7 + [_P] * 8),
    "maxpool": ("_Z17maxPool3dLauncheriiiiiPKiS0_PKfPfPi", [_I] * 5 + [_P] * 5),
    "maxpool_grad": ("_Z21maxPool3dGradLauncheriiiiPKiPKfPf", [_I] * 4 + [_P] * 3),
    "avgpool": ("_Z17avgPool3dLauncheriiiiiPKiS0_PKfPf", [_I] * 5 + [_P] * 4),
    "avgpool_grad": ("_Z21avgPool3dGradLauncheriiiiiPKiS0_PKfPf", [_I] * 5 + [_P] * 4),
    "mean": ("_Z23meanInterpolateLauncheriiiiiPKiS0_PKfPf", [_I] * 5 + [_P] * 4),
    "mean_grad": ("_Z27meanInterpolateGradLauncheriiiiiPKiS0_PKfPf", [_I] * 5 + [_P] * 4),
    "weighted": ("_Z27weightedInterpolateLauncheriiiiiPKiS0_PKfS2_Pf", [_I] * 5 + [_P] * 5),
    "weighted_grad": ("_Z31weightedInterpolateGradLauncheriiiiiPKiS0_PKfS2_Pf", [_I] * 5 + [_P] * 5),
    "fps": ("_Z27farthestPointSampleLauncheriiiPKfPfPi", [_I] * 3 + [_P] * 3),
}

_lib = None
_fn = {}


def available():
    return os.path.exists(LIB_PATH) and torch.cuda.is_available()


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(LIB_PATH)
        for key, (sym, args) in _SYMS.items():
            f = getattr(_lib, sym)
            f.restype = None
            f.argtypes = args
            _fn[key] = f
    return _fn


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _f(t):
    return t.detach().float().contiguous()


def _i(t):
    return t.detach().int().contiguous()


def _sync():
    torch.cuda.synchronize()


def build_sphere_neighbor(database, query, radius=0.1, dilation_rate=None, nnsample=100):
    database, query = _f(database[:, :, 0:3]), _f(query[:, :, 0:3])
    if dilation_rate is not None:
        radius = dilation_rate * radius
    B, N, _ = database.shape
    M = query.shape[1]
    idx = torch.zeros((B, M, nnsample), dtype=torch.int32, device=database.device)
    cnt = torch.zeros((B, M), dtype=torch.int32, device=database.device)
    dst = torch.zeros((B, M, nnsample), dtype=torch.float32, device=database.device)
    _sync()
    _load()["sphere"](B, N, M, nnsample, radius, _p(database), _p(query), _p(idx), _p(cnt), _p(dst))
    _sync()
    return idx, cnt, dst


def build_cube_neighbor(database, query, length=0.1, dilation_rate=None, nnsample=100, gridsize=3):
    database, query = _f(database[:, :, 0:3]), _f(query[:, :, 0:3])
    if dilation_rate is not None:
        length = dilation_rate * length
    B, N, _ = database.shape
    M = query.shape[1]
    idx = torch.zeros((B, M, nnsample, 2), dtype=torch.int32, device=database.device)
    cnt = torch.zeros((B, M), dtype=torch.int32, device=database.device)
    _sync()
    _load()["cube"](B, N, M, gridsize, nnsample, length, _p(database), _p(query), _p(idx), _p(cnt))
    _sync()
    return idx, cnt


def spherical_kernel(database, query, nn_index, nn_count, nn_dist, radius, kernel=[8, 2, 3]):
    n, p, q = kernel
    database, query = _f(database[:, :, 0:3]), _f(query[:, :, 0:3])
    nn_index, nn_count, nn_dist = _i(nn_index), _i(nn_count), _f(nn_dist)
    B, N, _ = database.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    filt = torch.zeros((B, M, K), dtype=torch.int32, device=database.device)
    _sync()
    _load()["kernel"](B, N, M, K, n, p, q, radius, _p(database), _p(query), _p(nn_index), _p(nn_count),
                      _p(nn_dist), _p(filt))
    _sync()
    return filt


def depthwise_conv3d(input, filter, nn_index, nn_count, bin_index):
    input, filter = _f(input), _f(filter)
    nn_index, nn_count, bin_index = _i(nn_index), _i(nn_count), _i(bin_index)
    B, N, C = input.shape
    F, _, r = filter.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    out = torch.zeros((B, M, C * r), dtype=torch.float32, device=input.device)
    _sync()
    _load()["conv"](B, N, M, C, r, K, _p(nn_index), _p(nn_count), _p(bin_index), _p(input), _p(filter), _p(out))
    _sync()
    return out


def depthwise_conv3d_grad(input, filter, grad_output, nn_index, nn_count, bin_index):
    input, filter, grad_output = _f(input), _f(filter), _f(grad_output)
    nn_index, nn_count, bin_index = _i(nn_index), _i(nn_count), _i(bin_index)
    B, N, C = input.shape
    F, _, r = filter.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    gi = torch.zeros_like(input)
    gf = torch.zeros_like(filter)
    _sync()
    _load()["conv_grad"](B, N, M, F, C, r, K, _p(nn_index), _p(nn_count), _p(bin_index), _p(input), _p(filter),
                         _p(grad_output), _p(gi), _p(gf))
    _sync()
    return gi, gf


def max_pool3d(input, nn_index, nn_count):
    input, nn_index, nn_count = _f(input), _i(nn_index), _i(nn_count)
    B, N, C = input.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    out = torch.zeros((B, M, C), dtype=torch.float32, device=input.device)
    mi = torch.zeros((B, M, C), dtype=torch.int32, device=input.device)
    _sync()
    _load()["maxpool"](B, N, M, C, K, _p(nn_index), _p(nn_count), _p(input), _p(out), _p(mi))
    _sync()
    return out, mi


def max_pool3d_grad(input, grad_output, max_index):
    input, grad_output, max_index = _f(input), _f(grad_output), _i(max_index)
    B, N, C = input.shape
    M = grad_output.shape[1]
    gi = torch.zeros_like(input)
    _sync()
    _load()["maxpool_grad"](B, N, M, C, _p(max_index), _p(grad_output), _p(gi))
    _sync()
    return gi


def avg_pool3d(input, nn_index, nn_count):
    input, nn_index, nn_count = _f(input), _i(nn_index), _i(nn_count)
    B, N, C = input.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    out = torch.zeros((B, M, C), dtype=torch.float32, device=input.device)
    _sync()
    _load()["avgpool"](B, N, M, C, K, _p(nn_index), _p(nn_count), _p(input), _p(out))
    _sync()
    return out


def avg_pool3d_grad(input, grad_output, nn_index, nn_count):
    input, grad_output, nn_index, nn_count = _f(input), _f(grad_output), _i(nn_index), _i(nn_count)
    B, N, C = input.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    gi = torch.zeros_like(input)
    _sync()
    _load()["avgpool_grad"](B, N, M, C, K, _p(nn_index), _p(nn_count), _p(grad_output), _p(gi))
    _sync()
    return gi


def mean_interpolate(input, nn_index, nn_count):
    input, nn_index, nn_count = _f(input), _i(nn_index), _i(nn_count)
    B, M, C = input.shape
    N, K = nn_index.shape[1], nn_index.shape[2]
    out = torch.zeros((B, N, C), dtype=torch.float32, device=input.device)
    _sync()
    _load()["mean"](B, N, M, C, K, _p(nn_index), _p(nn_count), _p(input), _p(out))
    _sync()
    return out


def mean_interpolate_grad(input, grad_output, nn_index, nn_count):
    input, grad_output, nn_index, nn_count = _f(input), _f(grad_output), _i(nn_index), _i(nn_count)
    B, M, C = input.shape
    N, K = nn_index.shape[1], nn_index.shape[2]
    gi = torch.zeros_like(input)
    _sync()
    _load()["mean_grad"](B, N, M, C, K, _p(nn_index), _p(nn_count), _p(grad_output), _p(gi))
    _sync()
    return gi


def weighted_interpolate(input, weight, nn_index, nn_count):
    input, weight, nn_index, nn_count = _f(input), _f(weight), _i(nn_index), _i(nn_count)
    B, M, C = input.shape
    N, K = nn_index.shape[1], nn_index.shape[2]
    out = torch.zeros((B, N, C), dtype=torch.float32, device=input.device)
    _sync()
    _load()["weighted"](B, N, M, C, K, _p(nn_index), _p(nn_count), _p(input), _p(weight), _p(out))
    _sync()
    return out


def weighted_interpolate_grad(input, grad_output, weight, nn_index, nn_count):
    input, grad_output, weight = _f(input), _f(grad_output), _f(weight)
    nn_index, nn_count = _i(nn_index), _i(nn_count)
    B, M, C = input.shape
    N, K = nn_index.shape[1], nn_index.shape[2]
    gi = torch.zeros_like(input)
    _sync()
    _load()["weighted_grad"](B, N, M, C, K, _p(nn_index), _p(nn_count), _p(grad_output), _p(weight), _p(gi))
    _sync()
    return gi


def farthest_point_sample(neursize, database):
    database = _f(database)
    b, n, _ = database.shape
    out = torch.zeros((b, neursize), dtype=torch.int32, device=database.device)
    temp = torch.empty((32, n), dtype=torch.float32, device=database.device)   # tf_sample.cpp:50
    _sync()
    _load()["fps"](b, n, neursize, _p(database), _p(temp), _p(out))
    _sync()
    return out
