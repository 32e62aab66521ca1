"""oracle — CPU restatement of the SPH3D-GCN tf_ops kernels (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py``; never by ``sph3d_gcn_amd``.

The arithmetic lives in ``oracle/sph3d_oracle.c`` (plain C, each function cites
the reference kernel it restates).  This module only marshals numpy arrays to
it through ctypes and mirrors the reference's op-level Python signatures
(``tf_ops/*/tf_*.py``) so tests read like calls of the reference ops.

Pinning status: see the header of ``sph3d_oracle.c`` and DESIGN.md §Oracle.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPH3D_ORACLE_LIB: another build of the same source (the sanitizer build of `make -C oracle sanitize`)
_LIB_PATH = os.environ.get("SPH3D_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")

_c_int = ctypes.c_int
_c_float = ctypes.c_float
_fp = ctypes.POINTER(ctypes.c_float)
_ip = ctypes.POINTER(ctypes.c_int)


def build(force=False, arch_flags=""):
    """Compile liboracle.so with gcc (seconds)."""
    if os.environ.get("SPH3D_ORACLE_LIB"):
        return _LIB_PATH
    src = os.path.join(_HERE, "sph3d_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "sph3d_atan2f.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    cmd = ["make", "-C", _HERE, "-B", "liboracle.so"]
    if arch_flags:
        cmd.append("ARCHF=" + arch_flags)
    subprocess.run(cmd, check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        if _lib.oracle_abi_version() != 1:
            raise RuntimeError("liboracle.so ABI mismatch")
    return _lib


def set_num_threads(n):
    """OpenMP threads used by the oracle (bench.py's cpu_baseline leg picks the best-performing count)."""
    lib().oracle_set_num_threads(_c_int(int(n)))


def get_max_threads():
    return int(lib().oracle_get_max_threads())


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _pf(a):
    return a.ctypes.data_as(_fp)


def _pi(a):
    return a.ctypes.data_as(_ip)


def _check(rc, name):
    if rc != 0:
        raise ValueError("oracle_%s rejected its arguments (status %d)" % (name, rc))


# ---- nnquery (tf_ops/nnquery/tf_nnquery.py) -------------------------------
def build_sphere_neighbor(database, query, radius=0.1, dilation_rate=None, nnsample=100, fixed=False):
    database = _f(np.asarray(database)[:, :, 0:3])
    query = _f(np.asarray(query)[:, :, 0:3])
    if dilation_rate is not None:
        radius = dilation_rate * radius
    B, N, _ = database.shape
    M = query.shape[1]
    idx = np.empty((B, M, nnsample), np.int32)
    cnt = np.empty((B, M), np.int32)
    dst = np.empty((B, M, nnsample), np.float32)
    fn = lib().oracle_build_sphere_neighbor_fixed if fixed else lib().oracle_build_sphere_neighbor
    rc = fn(_c_int(B), _c_int(N), _c_int(M), _c_int(nnsample), _c_float(radius),
            _pf(database), _pf(query), _pi(idx), _pi(cnt), _pf(dst))
    _check(rc, "build_sphere_neighbor")
    return idx, cnt, dst


def build_cube_neighbor(database, query, length=0.1, dilation_rate=None, nnsample=100, gridsize=3):
    database = _f(np.asarray(database)[:, :, 0:3])
    query = _f(np.asarray(query)[:, :, 0:3])
    if dilation_rate is not None:
        length = dilation_rate * length
    B, N, _ = database.shape
    M = query.shape[1]
    idx = np.empty((B, M, nnsample, 2), np.int32)
    cnt = np.empty((B, M), np.int32)
    rc = lib().oracle_build_cube_neighbor(_c_int(B), _c_int(N), _c_int(M), _c_int(gridsize), _c_int(nnsample),
                                          _c_float(length), _pf(database), _pf(query), _pi(idx), _pi(cnt))
    _check(rc, "build_cube_neighbor")
    return idx, cnt


# ---- buildkernel (tf_ops/buildkernel/tf_buildkernel.py) -------------------
def spherical_kernel(database, query, nn_index, nn_count, nn_dist, radius, kernel=[8, 2, 3]):
    n, p, q = kernel
    database = _f(np.asarray(database)[:, :, 0:3])
    query = _f(np.asarray(query)[:, :, 0:3])
    nn_index, nn_count, nn_dist = _i(nn_index), _i(nn_count), _f(nn_dist)
    B, N, _ = database.shape
    M = query.shape[1]
    K = nn_index.shape[2]
    filt = np.empty((B, M, K), np.int32)
    rc = lib().oracle_spherical_kernel(_c_int(B), _c_int(N), _c_int(M), _c_int(K), _c_int(n), _c_int(p), _c_int(q),
                                       _c_float(radius), _pf(database), _pf(query), _pi(nn_index), _pi(nn_count),
                                       _pf(nn_dist), _pi(filt))
    _check(rc, "spherical_kernel")
    return filt


def sphere_bin(dx, dy, dz, dist, radius, n, p, q):
    f = lib().oracle_sphere_bin
    f.restype = _c_int
    return f(_c_float(dx), _c_float(dy), _c_float(dz), _c_float(dist), _c_float(radius), _c_int(n), _c_int(p), _c_int(q))


# ---- convolution (tf_ops/convolution/tf_conv3d.py) ------------------------
def depthwise_conv3d(input, filter, nn_index, nn_count, bin_index):
    input, filter = _f(input), _f(filter)
    nn_index, nn_count, bin_index = _i(nn_index), _i(nn_count), _i(bin_index)
    B, N, C = input.shape
    F, C2, r = filter.shape
    if C2 != C:
        raise ValueError("Input Channel Size error!")
    M, K = nn_index.shape[1], nn_index.shape[2]
    out = np.empty((B, M, C * r), np.float32)
    rc = lib().oracle_depthwise_conv3d(_c_int(B), _c_int(N), _c_int(M), _c_int(F), _c_int(C), _c_int(r), _c_int(K),
                                       _pi(nn_index), _pi(nn_count), _pi(bin_index), _pf(input), _pf(filter), _pf(out))
    _check(rc, "depthwise_conv3d")
    return out


def depthwise_conv3d_grad(input, filter, grad_output, nn_index, nn_count, bin_index):
    input, filter, grad_output = _f(input), _f(filter), _f(grad_output)
    nn_index, nn_count, bin_index = _i(nn_index), _i(nn_count), _i(bin_index)
    B, N, C = input.shape
    F, _, r = filter.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    gi = np.empty((B, N, C), np.float32)
    gf = np.empty((F, C, r), np.float32)
    rc = lib().oracle_depthwise_conv3d_grad(_c_int(B), _c_int(N), _c_int(M), _c_int(F), _c_int(C), _c_int(r), _c_int(K),
                                            _pi(nn_index), _pi(nn_count), _pi(bin_index), _pf(input), _pf(filter),
                                            _pf(grad_output), _pf(gi), _pf(gf))
    _check(rc, "depthwise_conv3d_grad")
    return gi, gf


# ---- pooling (tf_ops/pooling/tf_pool3d.py) --------------------------------
def max_pool3d(input, nn_index, nn_count):
    input, nn_index, nn_count = _f(input), _i(nn_index), _i(nn_count)
    B, N, C = input.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    out = np.empty((B, M, C), np.float32)
    mi = np.empty((B, M, C), np.int32)
    _check(lib().oracle_max_pool3d(_c_int(B), _c_int(N), _c_int(M), _c_int(C), _c_int(K), _pi(nn_index),
                                   _pi(nn_count), _pf(input), _pf(out), _pi(mi)), "max_pool3d")
    return out, mi


def max_pool3d_grad(input, grad_output, max_index):
    input, grad_output, max_index = _f(input), _f(grad_output), _i(max_index)
    B, N, C = input.shape
    M = grad_output.shape[1]
    gi = np.empty((B, N, C), np.float32)
    _check(lib().oracle_max_pool3d_grad(_c_int(B), _c_int(N), _c_int(M), _c_int(C), _pi(max_index),
                                        _pf(grad_output), _pf(gi)), "max_pool3d_grad")
    return gi


def avg_pool3d(input, nn_index, nn_count):
    input, nn_index, nn_count = _f(input), _i(nn_index), _i(nn_count)
    B, N, C = input.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    out = np.empty((B, M, C), np.float32)
    _check(lib().oracle_avg_pool3d(_c_int(B), _c_int(N), _c_int(M), _c_int(C), _c_int(K), _pi(nn_index),
                                   _pi(nn_count), _pf(input), _pf(out)), "avg_pool3d")
    return out


def avg_pool3d_grad(input, grad_output, nn_index, nn_count):
    input, grad_output, nn_index, nn_count = _f(input), _f(grad_output), _i(nn_index), _i(nn_count)
    B, N, C = input.shape
    M, K = nn_index.shape[1], nn_index.shape[2]
    gi = np.empty((B, N, C), np.float32)
    _check(lib().oracle_avg_pool3d_grad(_c_int(B), _c_int(N), _c_int(M), _c_int(C), _c_int(K), _pi(nn_index),
                                        _pi(nn_count), _pf(grad_output), _pf(gi)), "avg_pool3d_grad")
    return gi


# ---- unpooling (tf_ops/unpooling/tf_unpool3d.py) --------------------------
def mean_interpolate(input, nn_index, nn_count):
    input, nn_index, nn_count = _f(input), _i(nn_index), _i(nn_count)
    B, M, C = input.shape
    N, K = nn_index.shape[1], nn_index.shape[2]
    out = np.empty((B, N, C), np.float32)
    _check(lib().oracle_mean_interpolate(_c_int(B), _c_int(N), _c_int(M), _c_int(C), _c_int(K), _pi(nn_index),
                                         _pi(nn_count), _pf(input), _pf(out)), "mean_interpolate")
    return out


def mean_interpolate_grad(input, grad_output, nn_index, nn_count):
    input, grad_output, nn_index, nn_count = _f(input), _f(grad_output), _i(nn_index), _i(nn_count)
    B, M, C = input.shape
    N, K = nn_index.shape[1], nn_index.shape[2]
    gi = np.empty((B, M, C), np.float32)
    _check(lib().oracle_mean_interpolate_grad(_c_int(B), _c_int(N), _c_int(M), _c_int(C), _c_int(K), _pi(nn_index),
                                              _pi(nn_count), _pf(grad_output), _pf(gi)), "mean_interpolate_grad")
    return gi


def weighted_interpolate(input, weight, nn_index, nn_count):
    input, weight, nn_index, nn_count = _f(input), _f(weight), _i(nn_index), _i(nn_count)
    B, M, C = input.shape
    N, K = nn_index.shape[1], nn_index.shape[2]
    out = np.empty((B, N, C), np.float32)
    _check(lib().oracle_weighted_interpolate(_c_int(B), _c_int(N), _c_int(M), _c_int(C), _c_int(K), _pi(nn_index),
                                             _pi(nn_count), _pf(input), _pf(weight), _pf(out)), "weighted_interpolate")
    return out


def weighted_interpolate_grad(input, grad_output, weight, nn_index, nn_count):
    input, grad_output, weight = _f(input), _f(grad_output), _f(weight)
    nn_index, nn_count = _i(nn_index), _i(nn_count)
    B, M, C = input.shape
    N, K = nn_index.shape[1], nn_index.shape[2]
    gi = np.empty((B, M, C), np.float32)
    _check(lib().oracle_weighted_interpolate_grad(_c_int(B), _c_int(N), _c_int(M), _c_int(C), _c_int(K),
                                                  _pi(nn_index), _pi(nn_count), _pf(grad_output), _pf(weight),
                                                  _pf(gi)), "weighted_interpolate_grad")
    return gi


# ---- sampling (tf_ops/sampling/tf_sample.py) ------------------------------
def farthest_point_sample(neursize, database):
    database = _f(database)
    if database.ndim != 3 or database.shape[2] != 3:
        raise ValueError("FarthestPointSample expects (batch_size,num_points,3) inp shape")
    if neursize <= 0:
        raise ValueError("FarthestPointSample expects positive npoint")
    b, n, _ = database.shape
    out = np.empty((b, neursize), np.int32)
    _check(lib().oracle_farthest_point_sample(_c_int(b), _c_int(n), _c_int(neursize), _pf(database), _pi(out)),
           "farthest_point_sample")
    return out
