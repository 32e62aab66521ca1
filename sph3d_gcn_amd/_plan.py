"""Per-graph tile plans of the LDS-tiled depthwise convolution (include/sph3d.h: sph3d_tile_plan).

A plan belongs to a neighbour graph, not to a convolution: the two separable convolutions of a level, their
gradients and every later step that reuses the tensors share it.  It needs the coordinates of the graph's points
(to put spatially close points in one tile); the convolution op itself never sees coordinates
(tf_ops/convolution/tf_conv3d.py:10-21), so ``tf_buildkernel.spherical_kernel`` — the op that produced the bin
indices from the coordinates — registers them here, keyed by the identity of its output tensor.  A convolution
called with a ``bin_index`` nobody registered (or a shape the tiled kernels do not cover) runs the gather kernels of
conv3d.hip; results are the same either way.

Entries hold strong references to the tensors they were built from (so a data_ptr cannot be recycled for another
graph while its entry lives) and an event for consumers on other streams, like ``_tgraph``.
"""
import collections
import ctypes

import torch

from . import _lib, _tgraph

UCAP = 236            # rows a tile stages: (UCAP + 4) * 256 B of rows + a 17-KB filter slice fit twice in a CU's 160-KB LDS
MIN_POINTS = 64       # below this a level is a handful of tiles: the gather kernels are used
_MAX_ENTRIES = 24

_mode = "auto"        # "auto" | "direct" (never tile) — a switch for tests and tools/, not a second backend
_variant = 0


def set_mode(mode, variant=0):
    global _mode, _variant
    if mode not in ("auto", "direct"):
        raise ValueError("mode must be 'auto' or 'direct'")
    _mode, _variant = mode, int(variant)


def variant():
    return _variant


def _ident(t):
    return (0, 0) if t is None else (t.data_ptr(), t._version)


_geom = collections.OrderedDict()      # ident(bin_index) -> (database_xyz, query_xyz, bin_index)
_orders = collections.OrderedDict()    # ident(xyz) -> entry
_fwd = collections.OrderedDict()
_bwd = collections.OrderedDict()


def clear():
    for d in (_geom, _orders, _fwd, _bwd):
        d.clear()


def _trim(d, n=_MAX_ENTRIES):
    while len(d) > n:
        d.popitem(last=False)


def register_geometry(bin_index, database, query):
    """called by tf_buildkernel.spherical_kernel: bin_index was computed from these coordinates"""
    _geom[_ident(bin_index)] = (database, query, bin_index)
    _trim(_geom, 2 * _MAX_ENTRIES)


def _entry(table, key, build, keep):
    """cached build with cross-stream ordering: -> tuple of tensors"""
    cur = torch.cuda.current_stream()
    hit = table.get(key)
    if hit is not None:
        table.move_to_end(key)
        out, _keep, ev, built_on = hit
        if built_on != cur.cuda_stream:
            cur.wait_event(ev)
            for t in out:
                if torch.is_tensor(t):
                    t.record_stream(cur)
        return out
    out = build()
    ev = torch.cuda.Event()
    ev.record(cur)
    table[key] = (out, keep, ev, cur.cuda_stream)
    _trim(table)
    return out


def spatial_order(xyz):
    """-> order[B,N] i32 (sph3d_spatial_order), cached per coordinate tensor"""
    def build():
        B, N = xyz.shape[0], xyz.shape[1]
        x = _lib.f32(xyz[:, :, 0:3])
        order = torch.empty((B, N), dtype=torch.int32, device=xyz.device)
        _lib.check(_lib.lib().sph3d_spatial_order(B, N, _lib.ptr(x), _lib.ptr(order), _lib.stream_ptr()))
        return (order,)
    return _entry(_orders, (_ident(xyz), tuple(xyz.shape)), build, (xyz,))[0]


def _sizes(B, T, F, E):
    n_c = ctypes.c_int()
    d, r, p, w = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    _lib.check(_lib.lib().sph3d_tile_plan_sizes(B, T, F, ctypes.c_longlong(E), ctypes.byref(n_c), ctypes.byref(d),
                                                ctypes.byref(r), ctypes.byref(p), ctypes.byref(w)))
    return d.value, r.value, p.value, w.value


def _tile_plan(B, T, NS, F, E, shared, order, bounds, key, key_count, dev, ucap):
    di, ri, pi, wi = _sizes(B, T, F, E)
    i32 = dict(dtype=torch.int32, device=dev)
    desc = torch.empty((di,), **i32)
    rows = torch.empty((ri,), **i32)
    scale = torch.empty((ri,), dtype=torch.float32, device=dev) if key_count is not None else None
    pbounds = torch.empty((pi,), **i32)
    slotw = torch.empty((wi,), **i32)
    pool = torch.empty((2,), **i32)
    _lib.check(_lib.lib().sph3d_tile_plan(B, T, NS, F, 1 if shared else 0, ucap, _lib.ptr(order), _lib.ptr(bounds),
                                          _lib.ptr(key), _lib.ptr(key_count), _lib.ptr(desc), _lib.ptr(rows),
                                          _lib.ptr(scale), _lib.ptr(pbounds), _lib.ptr(slotw), _lib.ptr(pool),
                                          _lib.stream_ptr()))
    return desc, rows, scale, pbounds, slotw


def eligible(N, M, K, F, C, r):
    return (_mode == "auto" and C % 4 == 0 and r in (1, 2) and F <= 63 and K <= 64 and min(N, M) >= MIN_POINTS)


def forward_plan(nn_index, nn_count, bin_index, F, ucap=None):
    """-> (order, desc, rows, pbounds, slotw, bounds, key, ucap) or None when the graph's coordinates are unknown"""
    g = _geom.get(_ident(bin_index))
    if g is None:
        return None
    query = g[1]
    ucap = UCAP if ucap is None else int(ucap)
    B, M, K = nn_index.shape
    if query.shape[1] != M:
        return None

    def build():
        dev = nn_index.device
        order = spatial_order(query)
        bounds = torch.empty((B * M * (F + 1),), dtype=torch.int32, device=dev)
        key = torch.empty((B * M * K + 64,), dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().sph3d_rows_by_bin(B, M, K, F, _lib.ptr(nn_index), _lib.ptr(nn_count), _lib.ptr(bin_index),
                                                _lib.ptr(bounds), _lib.ptr(key), _lib.stream_ptr()))
        N = g[0].shape[1]
        desc, rows, _s, pbounds, slotw = _tile_plan(B, M, N, F, B * M * K, False, order, bounds, key, None, dev, ucap)
        return (order, desc, rows, pbounds, slotw, bounds, key, ucap)

    k = (_ident(nn_index), _ident(nn_count), _ident(bin_index), F, tuple(nn_index.shape), ucap)
    return _entry(_fwd, k, build, (nn_index, nn_count, bin_index))


def backward_plan(nn_index, nn_count, bin_index, F, N, ucap=None):
    """-> (order, desc, rows, row_scale, pbounds, slotw, offsets, ent_key, ent_scale, ucap) or None"""
    g = _geom.get(_ident(bin_index))
    if g is None:
        return None
    database = g[0]
    ucap = UCAP if ucap is None else int(ucap)
    if database.shape[1] != N:
        return None
    B, M, K = nn_index.shape
    offsets, ent_key, ent_scale, _active = _tgraph.transpose(nn_index, nn_count, N, bin_index=bin_index, num_bins=F)

    def build():
        dev = nn_index.device
        order = spatial_order(database)
        desc, rows, scale, pbounds, slotw = _tile_plan(B, N, M, F, B * M * K, True, order, offsets, ent_key, nn_count, dev,
                                                       ucap)
        return (order, desc, rows, scale, pbounds, slotw, offsets, ent_key, ent_scale, ucap)

    k = (_ident(nn_index), _ident(nn_count), _ident(bin_index), F, tuple(nn_index.shape), int(N), ucap)
    return _entry(_bwd, k, build, (nn_index, nn_count, bin_index, offsets, ent_key, ent_scale))
