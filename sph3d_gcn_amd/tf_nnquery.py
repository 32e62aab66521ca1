"""Neighbour search ops — same names and arguments as the reference's
tf_ops/nnquery/tf_nnquery.py:9-60, running libsph3d's HIP kernels.

Registered as PyTorch custom ops ``sph3d::build_sphere_neighbor`` /
``sph3d::build_cube_neighbor`` (no gradient, like ops.NoGradient at :33,:60).
"""
from typing import Tuple

import torch

from . import _lib

_radius_mode = "compat"


def get_radius_mode():
    return _radius_mode


def set_radius_mode(mode):
    """"compat" (default): the reference's semantics, bit-exact — the radius a reference thread grew is carried to the next
    query of its chain (tf_nnquery_gpu.cu:59).  "fixed": every query is searched with the nominal radius (growth only until
    it has a neighbour) — a labelled deviation for clouds far larger than the reference's 8192-point blocks."""
    global _radius_mode
    if mode not in ("compat", "fixed"):
        raise ValueError("radius mode must be 'compat' or 'fixed'")
    _radius_mode = mode


def _build_sphere_neighbor_impl(database: torch.Tensor, query: torch.Tensor, radius: float,
                           nn_sample: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    _lib.require_device(database, query)
    # shape checks of BuildSphereNeighborGpuOp::Compute (tf_nnquery.cpp:76-77)
    if database.dim() != 3 or database.shape[2] != 3:
        raise ValueError("Shape of database points requires to be (batch, npoint, 3)")
    if query.dim() != 3 or query.shape[2] != 3:
        raise ValueError("Shape of query points requires to be (batch, mpoint, 3)")
    database, query = _lib.f32(database), _lib.f32(query)
    B, N, _ = database.shape
    M = query.shape[1]
    nn_index = _lib.empty((B, M, nn_sample), torch.int32, database.device)
    nn_count = _lib.empty((B, M), torch.int32, database.device)
    nn_dist = _lib.empty((B, M, nn_sample), torch.float32, database.device)
    l = _lib.lib()
    # the `_ws` entry points: the search's cell grid lives in memory of OURS (torch's stream-aware allocator), the library
    # allocates nothing (include/sph3d.h)
    sws, swsb = _search_workspace(l, B, N, M, database.device)
    fn = l.sph3d_build_sphere_neighbor_fixed_ws if _radius_mode == "fixed" else l.sph3d_build_sphere_neighbor_ws
    _lib.check(fn(
        B, N, M, nn_sample, radius, _lib.ptr(database), _lib.ptr(query),
        _lib.ptr(nn_index), _lib.ptr(nn_count), _lib.ptr(nn_dist), _lib.ptr(sws), swsb, _lib.stream_ptr()))
    return nn_index, nn_count, nn_dist


def _search_workspace(l, B, N, M, device):
    """-> (tensor or None, bytes): the neighbour search's cell-grid memory for one call (0 bytes: the shape never uses a grid)"""
    nbytes = l.sph3d_build_sphere_neighbor_workspace(B, N, M)
    if not nbytes:
        return None, 0
    return _lib.scratch(nbytes, device, slot=1), nbytes          # (slot 1: the call also holds a transpose workspace)


_build_sphere_neighbor = torch.library.custom_op("sph3d::build_sphere_neighbor", mutates_args=())(_build_sphere_neighbor_impl)


@_build_sphere_neighbor.register_fake
def _(database, query, radius, nn_sample):
    B, M = query.shape[0], query.shape[1]
    return (database.new_empty((B, M, nn_sample), dtype=torch.int32),
            database.new_empty((B, M), dtype=torch.int32),
            database.new_empty((B, M, nn_sample), dtype=torch.float32))


def _build_cube_neighbor_impl(database: torch.Tensor, query: torch.Tensor, length: float, nn_sample: int,
                         grid_size: int) -> Tuple[torch.Tensor, torch.Tensor]:
    _lib.require_device(database, query)
    if database.dim() != 3 or database.shape[2] != 3:
        raise ValueError("Shape of database points requires to be (batch, npoint, 3)")
    if query.dim() != 3 or query.shape[2] != 3:
        raise ValueError("Shape of query points requires to be (batch, mpoint, 3)")
    database, query = _lib.f32(database), _lib.f32(query)
    B, N, _ = database.shape
    M = query.shape[1]
    nn_index = torch.empty((B, M, nn_sample, 2), dtype=torch.int32, device=database.device)
    nn_count = torch.empty((B, M), dtype=torch.int32, device=database.device)
    _lib.check(_lib.lib().sph3d_build_cube_neighbor(
        B, N, M, grid_size, nn_sample, length, _lib.ptr(database), _lib.ptr(query),
        _lib.ptr(nn_index), _lib.ptr(nn_count), _lib.stream_ptr()))
    return nn_index, nn_count


_build_cube_neighbor = torch.library.custom_op("sph3d::build_cube_neighbor", mutates_args=())(_build_cube_neighbor_impl)


@_build_cube_neighbor.register_fake
def _(database, query, length, nn_sample, grid_size):
    B, M = query.shape[0], query.shape[1]
    return (database.new_empty((B, M, nn_sample, 2), dtype=torch.int32),
            database.new_empty((B, M), dtype=torch.int32))


def build_sphere_neighbor(database, query, radius=0.1, dilation_rate=None, nnsample=100):
    """Range search (public signature of tf_nnquery.py:9-31): for every query point the first `nnsample` database points,
    in ascending index order, that lie strictly inside the search sphere.

    database  [B, N, >=3] fp32 (only x, y, z are used)      query  [B, M, >=3] fp32
    radius    search radius; multiplied by `dilation_rate` when that is given
    returns   nn_index [B, M, nnsample] int32 (unused slots 0), nn_count [B, M] int32,
              nn_dist  [B, M, nnsample] fp32 = SQUARE ROOT of the Euclidean distance (the reference's quirk, kept)

    In the default "compat" mode the radius grows along each reference thread's chain of queries exactly as in the
    reference kernel; see set_radius_mode.  No gradient.
    """
    database = database[:, :, 0:3]
    query = query[:, :, 0:3]
    if dilation_rate is not None:
        radius = dilation_rate * radius
    return _build_sphere_neighbor_impl(database, query, float(radius), int(nnsample))


def build_cube_neighbor(database, query, length=0.1, dilation_rate=None, nnsample=100, gridsize=3):
    """Cube search (public signature of tf_nnquery.py:34-52): neighbours inside the axis-aligned cube of edge `length`
    (times `dilation_rate` if given) around each query, with the cell of a gridsize^3 grid each one falls in.

    returns   nn_index [B, M, nnsample, 2] int32 = (neighbour, grid cell), nn_count [B, M] int32.  No gradient.
    """
    database = database[:, :, 0:3]
    query = query[:, :, 0:3]
    if dilation_rate is not None:
        length = dilation_rate * length
    return _build_cube_neighbor_impl(database, query, float(length), int(nnsample), int(gridsize))


def build_sphere_graph(xyz, radius, nnsample, kernel, with_transpose=True):
    """Fused graph construction of one level (not in the reference's API): the intra-level neighbour graph of
    `xyz` AND its spherical-kernel bins from one kernel — the same tensors, bit for bit, as
    ``build_sphere_neighbor(xyz, xyz, radius, None, nnsample)`` followed by ``spherical_kernel(xyz, xyz, ..., radius,
    kernel)`` — and, with_transpose, the counting pass of the transposed graph the convolution gradients gather over (it is
    finished and cached here, so the backward pass finds it ready).
    -> nn_index, nn_count, nn_dist, filt_index"""
    from . import _tgraph
    xyz = _lib.f32(xyz[:, :, 0:3])
    _lib.require_device(xyz)
    if _radius_mode == "fixed":
        raise ValueError("build_sphere_graph implements the reference (compat) radius semantics only")
    n, p, q = (int(v) for v in kernel)
    B, N, _ = xyz.shape
    K = int(nnsample)
    F = n * p * q + 1
    dev = xyz.device
    nn_index = _lib.empty((B, N, K), torch.int32, dev)
    nn_count = _lib.empty((B, N), torch.int32, dev)
    nn_dist = _lib.empty((B, N, K), torch.float32, dev)
    filt = _lib.empty((B, N, K), torch.int32, dev)
    l = _lib.lib()
    ws, wsb = None, 0
    if with_transpose:
        wsb = l.sph3d_graph_transpose_workspace(B, N, N, K, F)
        ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=dev)
    from . import tf_buildkernel
    # the atan2 mode of tf_buildkernel reaches the fused kernel too ("ocml", the default: the bins of the reference's own build)
    sws, swsb = _search_workspace(l, B, N, N, dev)
    _lib.check(l.sph3d_build_sphere_graph_ws(
        B, N, N, K, float(radius), n, p, q, 1 if tf_buildkernel._atan2 == "ocml" else 0, _lib.ptr(xyz), _lib.ptr(xyz),
        _lib.ptr(nn_index), _lib.ptr(nn_count), _lib.ptr(nn_dist), _lib.ptr(filt), _lib.ptr(ws), wsb, _lib.ptr(sws), swsb,
        _lib.stream_ptr()))
    if with_transpose:
        _tgraph.transpose(nn_index, nn_count, N, bin_index=filt, num_bins=F, counted_workspace=ws)
    return nn_index, nn_count, nn_dist, filt


def build_sphere_neighbor_counted(database, query, radius, nnsample):
    """build_sphere_neighbor(database, query, radius, None, nnsample) whose kernel also counts the in-edges of every
    database point, i.e. runs the first pass of the transposed graph that the un-pooling gradient gathers over
    (tf_unpool3d: mean interpolation); the transpose is finished and cached here.  Same three tensors, bit for bit.
    -> nn_index, nn_count, nn_dist"""
    from . import _tgraph
    database = _lib.f32(database[:, :, 0:3])
    query = _lib.f32(query[:, :, 0:3])
    _lib.require_device(database, query)
    if _radius_mode == "fixed":
        raise ValueError("build_sphere_neighbor_counted implements the reference (compat) radius semantics only")
    B, N, _ = database.shape
    M = query.shape[1]
    K = int(nnsample)
    dev = database.device
    nn_index = _lib.empty((B, M, K), torch.int32, dev)
    nn_count = _lib.empty((B, M), torch.int32, dev)
    nn_dist = _lib.empty((B, M, K), torch.float32, dev)
    l = _lib.lib()
    wsb = l.sph3d_graph_transpose_workspace(B, N, M, K, 1)
    ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=dev)
    sws, swsb = _search_workspace(l, B, N, M, dev)
    _lib.check(l.sph3d_build_sphere_graph_ws(B, N, M, K, float(radius), 0, 0, 0, 0, _lib.ptr(database), _lib.ptr(query),
                                             _lib.ptr(nn_index), _lib.ptr(nn_count), _lib.ptr(nn_dist), _lib.ptr(None), _lib.ptr(ws),
                                             wsb, _lib.ptr(sws), swsb, _lib.stream_ptr()))
    _tgraph.transpose(nn_index, nn_count, N, counted_workspace=ws)
    return nn_index, nn_count, nn_dist
