"""Per-graph LDS tile plans of the depthwise convolution (include/sph3d.h: sph3d_conv_plan; csrc/convlds.hip).

A plan belongs to a neighbour graph, not to a convolution: every convolution (and every 64-channel slice of it) that
reuses the graph's tensors shares it.  It groups the graph's output points into tiles of <= 32 spatially consecutive
points whose neighbour rows fit the LDS; "spatially consecutive" needs the points' coordinates, which the convolution
op never sees (tf_ops/convolution/tf_conv3d.py:10-21) — so the ops that produce the bin indices from the coordinates
(``tf_buildkernel.spherical_kernel``, ``tf_nnquery.build_sphere_graph``) register them here, keyed by the identity of
their output tensor.  Without coordinates the plan uses index order (correct, less reuse per tile).

Results do not depend on the plan: the kernel sums a point's neighbours in neighbour order with the gather kernel's
arithmetic, so ``"lds"`` and ``"gather"`` modes produce the same bits.

Entries hold strong references to the tensors they were built from (so a data_ptr cannot be recycled for another
graph while its entry lives) and an event for consumers on other streams, like ``_tgraph``.
"""
import collections
import ctypes

import torch

from . import _lib

MIN_POINTS = 64       # below this a level is a handful of tiles: the gather kernels are used
_MAX_ENTRIES = 24

_mode = "gather"      # "gather" | "lds": which forward kernel a covered convolution uses


def set_mode(mode):
    global _mode
    if mode not in ("gather", "lds"):
        raise ValueError("mode must be 'gather' or 'lds'")
    _mode = mode


def get_mode():
    return _mode


def _ident(t):
    return (0, 0) if t is None else (t.data_ptr(), t._version)


_geom = collections.OrderedDict()      # ident(bin_index) -> (query_xyz, bin_index)
_orders = collections.OrderedDict()    # ident(xyz) -> entry
_plans = collections.OrderedDict()


def clear():
    for d in (_geom, _orders, _plans):
        d.clear()


def _trim(d, n=_MAX_ENTRIES):
    while len(d) > n:
        d.popitem(last=False)


def register_geometry(bin_index, database, query):
    """called by the binning ops: bin_index was computed for these query coordinates"""
    _geom[_ident(bin_index)] = (query, bin_index)
    _trim(_geom, 2 * _MAX_ENTRIES)


def _entry(table, key, build, keep):
    """cached build with cross-stream ordering: -> tuple of tensors"""
    cur_raw = _lib.current_raw_stream()
    hit = table.get(key)
    if hit is not None:
        table.move_to_end(key)
        out, _keep, ev, synced = hit
        if cur_raw not in synced:                # built ahead of time on the graph stream: order this stream after it, ONCE
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for t in out:
                if torch.is_tensor(t):
                    t.record_stream(cur)
            synced.add(cur_raw)
        return out
    out = build()
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    table[key] = (out, keep, ev, {cur_raw})
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


def applies(N, M, K, F, C, r):
    """does the LDS kernel cover this layer (and is the mode on)?"""
    return (_mode == "lds" and min(N, M) >= MIN_POINTS and N <= 65536
            and bool(_lib.lib().sph3d_depthwise_conv3d_lds_supported(F, C, r, K)))


def conv_plan(nn_index, nn_count, bin_index, F, n_src):
    """-> (chunk_hdr, records, target_meta, row_lists) of the graph (include/sph3d.h: sph3d_conv_plan)"""
    B, M, K = nn_index.shape

    def build():
        dev = nn_index.device
        l = _lib.lib()
        g = _geom.get(_ident(bin_index))
        order = spatial_order(g[0]) if (g is not None and g[0].shape[1] == M) else None
        sz = [ctypes.c_size_t() for _ in range(4)]
        _lib.check(l.sph3d_conv_plan_sizes(B, M, *[ctypes.byref(x) for x in sz]))
        hdr = torch.empty((sz[0].value,), dtype=torch.int32, device=dev)
        rec = torch.empty((sz[1].value,), dtype=torch.int32, device=dev)
        meta = torch.empty((sz[2].value,), dtype=torch.int32, device=dev)
        rows = torch.empty((sz[3].value,), dtype=torch.int16, device=dev)
        _lib.check(l.sph3d_conv_plan(B, int(n_src), M, K, F, _lib.ptr(order), _lib.ptr(nn_index), _lib.ptr(nn_count),
                                     _lib.ptr(bin_index), _lib.ptr(hdr), _lib.ptr(rec), _lib.ptr(meta), _lib.ptr(rows),
                                     _lib.stream_ptr()))
        return (hdr, rec, meta, rows)

    k = (_ident(nn_index), _ident(nn_count), _ident(bin_index), int(F), int(n_src), tuple(nn_index.shape))
    return _entry(_plans, k, build, (nn_index, nn_count, bin_index))


def prebuild(nn_index, nn_count, bin_index, F, n_src):
    """build the plan now, on the caller's (graph) stream, if a convolution on this graph could use it"""
    B, M, K = nn_index.shape
    if _mode == "lds" and K <= 64 and min(n_src, M) >= MIN_POINTS and n_src <= 65536 and _lib.lib().sph3d_conv_plan_ucap(F) > 0:
        conv_plan(nn_index, nn_count, bin_index, F, n_src)
