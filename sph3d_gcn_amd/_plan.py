"""Per-graph tile plans of the LDS-tiled depthwise convolution (include/sph3d.h: sph3d_tile_plan).

A plan belongs to a neighbour graph, not to a convolution: every convolution that reuses the graph's tensors shares
it.  It needs the coordinates of the graph's points (to put spatially close points in one tile); the convolution op
itself never sees coordinates (tf_ops/convolution/tf_conv3d.py:10-21), so ``tf_buildkernel.spherical_kernel`` — the op
that produced the bin indices from the coordinates — registers them here, keyed by the identity of its output tensor.

Measured trade-off (round 2, MI355X, B = 16 x 8192 points): the tiled forward kernel runs a C = 128 layer in 0.207 ms
against 0.275 ms for the gather kernel, but a plan costs 0.32 ms per level-0 graph.  A training step builds new graphs
every step and uses each for two forward convolutions, so the default mode is ``"gather"``; ``set_mode("tiled")`` is for
callers that keep a graph (inference on a fixed cloud, many steps on one batch) and for the tests / tools.  Results are
the same either way up to fp32 summation order.

Entries hold strong references to the tensors they were built from (so a data_ptr cannot be recycled for another
graph while its entry lives) and an event for consumers on other streams, like ``_tgraph``.
"""
import collections
import ctypes

import torch

from . import _lib

UCAP = 236            # rows a tile stages: (UCAP + 2) * 512 B of rows + the 33-KB filter of 33 bins x 256 outputs fit 160 KB of LDS
MIN_POINTS = 64       # below this a level is a handful of tiles: the gather kernels are used
_MAX_ENTRIES = 16

_mode = "gather"      # "gather" | "tiled": which forward kernel a convolution with a registered graph geometry uses


def set_mode(mode):
    global _mode
    if mode not in ("gather", "tiled"):
        raise ValueError("mode must be 'gather' or 'tiled'")
    _mode = mode


def get_mode():
    return _mode


def _ident(t):
    return (0, 0) if t is None else (t.data_ptr(), t._version)


_geom = collections.OrderedDict()      # ident(bin_index) -> (database_xyz, query_xyz, bin_index)
_orders = collections.OrderedDict()    # ident(xyz) -> entry
_fwd = collections.OrderedDict()


def clear():
    for d in (_geom, _orders, _fwd):
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


def applies(N, M, K, F, C, r, ucap=None):
    """does the tiled forward kernel cover this layer (and is the mode on)?"""
    ucap = UCAP if ucap is None else int(ucap)
    return (_mode == "tiled" and K <= 64 and min(N, M) >= MIN_POINTS
            and bool(_lib.lib().sph3d_depthwise_conv3d_tiled_supported(F, C, r, K, ucap)))


def forward_plan(nn_index, nn_count, bin_index, F, ucap=None):
    """-> (hdr, targets, rows, pb, slotw, xsteps, counters, bounds, key, ucap) (see include/sph3d.h: sph3d_tile_plan), or
    None when nobody registered the coordinates of this graph"""
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
        l = _lib.lib()
        order = spatial_order(query)
        bounds = torch.empty((B * M * (F + 1),), dtype=torch.int32, device=dev)
        key = torch.empty((B * M * K + 64,), dtype=torch.int32, device=dev)
        _lib.check(l.sph3d_rows_by_bin(B, M, K, F, _lib.ptr(nn_index), _lib.ptr(nn_count), _lib.ptr(bin_index),
                                       _lib.ptr(bounds), _lib.ptr(key), _lib.stream_ptr()))
        N = g[0].shape[1]
        n_c = ctypes.c_int()
        sz = [ctypes.c_size_t() for _ in range(7)]
        _lib.check(l.sph3d_tile_plan_sizes(B, M, F, ucap, ctypes.c_longlong(B * M * K), ctypes.byref(n_c),
                                           *[ctypes.byref(x) for x in sz]))
        hdr, tgt, rows, pb, slotw, xsteps, counters = (torch.empty((x.value,), dtype=torch.int32, device=dev) for x in sz)
        _lib.check(l.sph3d_tile_plan(B, M, N, F, ucap, _lib.ptr(order), _lib.ptr(bounds), _lib.ptr(key), _lib.ptr(hdr),
                                     _lib.ptr(tgt), _lib.ptr(rows), _lib.ptr(pb), _lib.ptr(slotw), _lib.ptr(xsteps),
                                     _lib.ptr(counters), _lib.stream_ptr()))
        return (hdr, tgt, rows, pb, slotw, xsteps, counters, bounds, key, ucap)

    k = (_ident(nn_index), _ident(nn_count), _ident(bin_index), F, tuple(nn_index.shape), ucap)
    return _entry(_fwd, k, build, (nn_index, nn_count, bin_index))

