"""ctypes binding of libsph3d.so — the C ABI declared in include/sph3d.h.

This is the ONLY compute path of the package.  There is no CPU or eager-PyTorch
fallback: if the library is missing or a tensor is not on a HIP device the ops
raise.  (The CPU oracle lives in /oracle and is test infrastructure.)
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPH3D_LIB: another build of the same library (A/B measurements of kernel variants); there is still no non-HIP path
LIB_PATH = os.environ.get("SPH3D_LIB") or os.path.join(_HERE, "csrc", "libsph3d.so")

_c_int = ctypes.c_int
_c_float = ctypes.c_float
_c_size_t = ctypes.c_size_t
_vp = ctypes.c_void_p

# name -> (restype, argtypes); one entry per declaration in include/sph3d.h
_I, _F, _P, _S = _c_int, _c_float, _vp, _c_size_t
SIGNATURES = {
    "sph3d_abi_version": (_I, []),
    "sph3d_last_error": (ctypes.c_char_p, []),
    "sph3d_build_info": (ctypes.c_char_p, []),
    "sph3d_build_sphere_neighbor": (_I, [_I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P]),
    "sph3d_build_sphere_neighbor_fixed": (_I, [_I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P]),
    "sph3d_build_sphere_neighbor_workspace": (_S, [_I, _I, _I]),
    "sph3d_build_sphere_neighbor_ws": (_I, [_I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _S, _P]),
    "sph3d_build_sphere_neighbor_fixed_ws": (_I, [_I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _S, _P]),
    "sph3d_release_stream_scratch": (_I, [_P]),
    "sph3d_release_all_scratch": (_I, []),
    "sph3d_build_sphere_graph_ws": (_I, [_I, _I, _I, _I, _F, _I, _I, _I, _I] + [_P] * 6 + [_P, _S, _P, _S, _P]),
    "sph3d_build_cube_neighbor": (_I, [_I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P]),
    "sph3d_spherical_kernel": (_I, [_I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P]),
    "sph3d_spherical_kernel_ocml": (_I, [_I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P]),
    "sph3d_depthwise_conv3d": (_I, [_I] * 7 + [_P] * 7),
    "sph3d_depthwise_conv3d_grad_workspace": (_S, [_I] * 7),
    "sph3d_depthwise_conv3d_grad": (_I, [_I] * 7 + [_P] * 8 + [_P, _S, _P]),
    "sph3d_max_pool3d": (_I, [_I] * 5 + [_P] * 6),
    "sph3d_max_pool3d_grad": (_I, [_I] * 4 + [_P] * 4),
    "sph3d_max_pool3d_grad_t": (_I, [_I] * 4 + [_P] * 8),
    "sph3d_avg_pool3d": (_I, [_I] * 5 + [_P] * 5),
    "sph3d_avg_pool3d_grad": (_I, [_I] * 5 + [_P] * 4 + [_P, _S, _P]),
    "sph3d_graph_balanced_order": (_I, [_I, _I, _I, _P, _P, _P]),
    "sph3d_gather_nd": (_I, [_I, _I, ctypes.c_longlong, _I, _P, _P, _P, _P]),
    "sph3d_gather_rows_count": (_I, [_I] * 4 + [_P] * 5 + [_P, _S, _P]),
    "sph3d_graph_transpose_workspace": (_S, [_I] * 5),
    "sph3d_graph_transpose": (_I, [_I] * 5 + [_P] * 8 + [_P, _S, _P]),
    "sph3d_graph_transpose_count": (_I, [_I] * 5 + [_P] * 3 + [_I, _P, _S, _P]),
    "sph3d_graph_transpose_finish": (_I, [_I] * 5 + [_P] * 8 + [_P, _S, _P]),
    "sph3d_graph_transpose_finish_ordered": (_I, [_I] * 5 + [_P] * 9 + [_P, _S, _P]),
    "sph3d_build_sphere_graph": (_I, [_I, _I, _I, _I, _F, _I, _I, _I] + [_P] * 6 + [_P, _S, _P]),
    "sph3d_build_sphere_graph_ocml": (_I, [_I, _I, _I, _I, _F, _I, _I, _I] + [_P] * 6 + [_P, _S, _P]),
    "sph3d_depthwise_conv3d_grad_t_workspace": (_S, [_I] * 5),
    "sph3d_depthwise_conv3d_grad_t": (_I, [_I] * 6 + [_P] * 10 + [_P, _S, _P]),
    "sph3d_depthwise_conv3d_cat_supported": (_I, [_I] * 4),
    "sph3d_depthwise_conv3d_cat": (_I, [_I] * 8 + [_P] * 8),
    "sph3d_depthwise_conv3d_grad_t_cat": (_I, [_I] * 7 + [_P] * 12 + [_P, _S, _P]),
    "sph3d_spatial_order": (_I, [_I, _I, _P, _P, _P]),
    "sph3d_scatter_grad_t": (_I, [_I] * 4 + [_P] * 6),
    "sph3d_scatter_grad_workspace": (_S, [_I] * 4),
    "sph3d_mean_interpolate": (_I, [_I] * 5 + [_P] * 5),
    "sph3d_mean_interpolate_grad": (_I, [_I] * 5 + [_P] * 4 + [_P, _S, _P]),
    "sph3d_weighted_interpolate": (_I, [_I] * 5 + [_P] * 6),
    "sph3d_weighted_interpolate_grad": (_I, [_I] * 5 + [_P] * 5 + [_P, _S, _P]),
    "sph3d_farthest_point_sample_workspace": (_S, [_I] * 3),
    "sph3d_farthest_point_sample": (_I, [_I] * 3 + [_P, _P, _P, _S, _P]),
    "sph3d_pointwise_gemm": (_I, [_I] * 3 + [_P, _P, _P, _I, _I, _P, _P]),
    "sph3d_pointwise_gemm_mode": (_I, [_I]),
    "sph3d_pointwise_gemm_tn_workspace": (_S, [_I] * 3),
    "sph3d_pointwise_gemm_tn": (_I, [_I] * 3 + [_P, _P, _P, _P, _S, _P]),
    "sph3d_elu_bn_workspace": (_S, [_I] * 2),
    "sph3d_pointwise_gemm_bnstats_blocks": (_I, [_I] * 3),
    "sph3d_pointwise_gemm_bnstats": (_I, [_I] * 3 + [_P] * 6),
    "sph3d_nngrid_launches": (ctypes.c_longlong, []),
    "sph3d_masked_softmax_xent_parts": (_I, [_I]),
    "sph3d_masked_softmax_xent": (_I, [_I, _I, _I] + [_P] * 5 + [_P]),
    "sph3d_adam_step": (_I, [ctypes.c_longlong] + [_P] * 4 + [_F] * 4 + [_I, _P]),
    "sph3d_pointwise_gemm_skinny_supported": (_I, [_I] * 4),
    "sph3d_pointwise_gemm_skinny": (_I, [_I] * 4 + [_P] * 6),
    "sph3d_pointwise_gemm_skinny_tn_workspace": (_S, [_I] * 4),
    "sph3d_pointwise_gemm_skinny_tn": (_I, [_I] * 4 + [_P] * 5 + [_S, _P]),
    "sph3d_separable_conv3d_fused_supported": (_I, [_I] * 6),
    "sph3d_separable_conv3d_fused": (_I, [_I] * 9 + [_P] * 11),
    "sph3d_separable_conv3d_train_supported": (_I, [_I] * 6),
    "sph3d_separable_conv3d_train_blocks": (_I, [_I]),
    "sph3d_separable_conv3d_train": (_I, [_I] * 8 + [_P] * 11),
    "sph3d_separable_conv3d_ring_failures": (_I, []),
    "sph3d_pointwise_gemm_exchange_failures": (_I, []),
    "sph3d_elu_bn_forward_partials": (_I, [_I] * 3 + [_P] * 6 + [_F, _F] + [_P] * 4),
    "sph3d_elu_bn_forward": (_I, [_I, _I, _P, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _S, _P]),
    "sph3d_elu_bn_backward": (_I, [_I, _I] + [_P] * 5 + [_I] + [_P] * 3 + [_P, _S, _P]),
}
# test-only hook exported by the library but not part of the reference surface
_EXTRA = {
    "sph3d_selftest_math": (_I, [_I, _P, _P, _P, _P, _P, _P, _P]),
}

ABI_VERSION = 2          # include/sph3d.h: SPH3D_ABI_VERSION
_lib = None


class Sph3dError(RuntimeError):
    pass


def lib():
    """Load libsph3d.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Sph3dError(
            "libsph3d.so is not built (%s).  Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C sph3d_gcn_amd/csrc`.  There is no fallback path." % LIB_PATH)
    l = ctypes.CDLL(LIB_PATH)
    for table in (SIGNATURES, _EXTRA):
        for name, (res, args) in table.items():
            try:
                fn = getattr(l, name)
            except AttributeError:
                if table is SIGNATURES:
                    raise Sph3dError("libsph3d.so does not export %s" % name)
                continue
            fn.restype = res
            fn.argtypes = args
    if l.sph3d_abi_version() != ABI_VERSION:
        raise Sph3dError("libsph3d.so ABI version %d, this package binds version %d" % (l.sph3d_abi_version(), ABI_VERSION))
    _lib = _Proxy(l)
    return _lib


# ---- optional per-call HIP-event timing (bench.py's roofline leg) ---------------------------------
# When enabled, every kernel-launching C-ABI call is bracketed by two events recorded on the stream the
# kernel is launched on (torch's current stream), so elapsed_time() is that call's device time.
_timing = None


def timing_start():
    global _timing
    _timing = []


def timing_stop():
    """-> list of (name, int_args, start_event, end_event); events are resolved by the caller after a sync"""
    global _timing
    out, _timing = _timing, None
    return out


_PURE = ("_workspace", "_blocks", "_supported", "_parts")
_NO_TIME = ("sph3d_abi_version", "sph3d_last_error", "sph3d_build_info", "workspace", "_release_", "_blocks", "_supported", "_launches", "_parts", "_failures", "_mode")


class _Proxy:
    """Attribute access returns the ctypes function, wrapped with event timing while timing is on."""

    def __init__(self, cdll):
        self._cdll = cdll
        self._cache = {}

    def __getattr__(self, name):
        # (reached once per symbol: the result is stored on the instance, so later look-ups never come here — 180 look-ups per
        # step at 2 us each otherwise)
        fn = getattr(self._cdll, name)     # AttributeError if not exported
        if any(k in name for k in _PURE) and name != "sph3d_pointwise_gemm_bnstats_blocks":
            # pure functions of their integer arguments (workspace sizes, shape predicates): a step asks ~75 of them, always
            # for the same shapes — answered from a dict instead of a foreign call each time
            memo = {}

            def cached(*args, _fn=fn, _memo=memo):
                r = _memo.get(args)
                if r is None:
                    r = _memo[args] = _fn(*args)
                return r
            self.__dict__[name] = cached
            return cached
        if any(k in name for k in _NO_TIME):
            self.__dict__[name] = fn
            return fn
        w = self._cache.get(name)
        if w is None:
            # the call's integer arguments (its dimensions): by declared type — device addresses are plain ints too
            sig = SIGNATURES.get(name) or _EXTRA.get(name)
            int_pos = tuple(i for i, t in enumerate(sig[1]) if t is _I) if sig else ()

            def w(*args, _fn=fn, _name=name, _pos=int_pos):
                if _timing is None:
                    return _fn(*args)
                st = torch.cuda.current_stream()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(st)
                e0.raw_stream = st.cuda_stream          # (for timelines: tools/exp_step_timeline.py)
                rc = _fn(*args)
                e1.record(st)
                _timing.append((_name, tuple(args[i] for i in _pos if i < len(args)), e0, e1))
                return rc
            self._cache[name] = w
        self.__dict__[name] = w
        return w


def check(rc):
    """Map a C-ABI status to the exception the reference's OP_REQUIRES would have raised."""
    if rc == 0:
        return
    msg = lib().sph3d_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg)          # errors::InvalidArgument
    raise Sph3dError("libsph3d status %d: %s" % (rc, msg))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def current_raw_stream():
    """hipStream_t of torch's current stream as an int.  (torch.cuda.current_stream() builds a Stream object through three
    layers of device-index helpers: 9 us per call, 130 calls per step = 1.2 ms of the 8 ms the host needs to issue a step.)"""
    if _raw_stream is not None and _get_device is not None:
        return _raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


def stream_ptr():
    return current_raw_stream()          # ctypes turns the int into the void* argument


def require_device(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise Sph3dError("sph3d ops run on the HIP device only (got a %s tensor); "
                             "the CPU oracle is test infrastructure, not a fallback" % t.device)


def f32(t):
    """contiguous float32 view/copy on the same device"""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def i32(t):
    if t.dtype != torch.int32:
        t = t.int()
    return t.contiguous()


def ptr(t):
    """device address for a void* argument (ctypes converts the int; None = NULL)"""
    return t.data_ptr() if t is not None else None


# ---- per-stream scratch for workspaces that are dead when the call returns (gradient slabs, split-K slabs, statistics
# partials): one grow-only buffer per (stream, slot) instead of a torch.empty per call (the step makes ~80 such allocations;
# work on one stream is ordered, so consecutive calls may share the bytes).  NOT for memory a later call reads: every call
# site passes it as the `workspace` of ONE C-ABI call, dead when that call's kernels are done.  release_scratch() frees.
_scratch = {}


def release_scratch(stream=None):
    """drop the call-local workspaces of one raw stream handle (None: of every stream) — for a stream that is being destroyed (its
    handle value may be recycled for a new stream of another pool) or to give the memory back; the next call allocates again"""
    for key in [k for k in _scratch if stream is None or k[1] == stream]:
        del _scratch[key]


def scratch(nbytes, device, slot=0):
    """-> a uint8 tensor of at least nbytes on `device` for the current stream, or None for nbytes == 0"""
    if not nbytes:
        return None
    key = (device.index, current_raw_stream(), slot)      # (the null stream of two devices is one handle value)
    t = _scratch.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty((nbytes + nbytes // 4,), dtype=torch.uint8, device=device)
        _scratch[key] = t
    return t


# ---- one allocation per graph plan and producing stream ----------------------------------------------------------------------
# The tensors of a graph plan (neighbour lists, bins, samples, transposed graphs: ~110 per S3DIS step) are produced on side streams
# and read by the main stream's kernels, so each is `record_stream`-ed for the main stream — and the caching allocator then records
# ONE EVENT PER BLOCK AND STREAM when a block is freed.  A step frees them in a burst (when the transposed-graph cache evicts the
# step's entries), and ~100 event records are ~100 marker packets in the main stream's queue: 0.47 ms in which no kernel of the
# feature path runs (tools/exp_step_timeline.py: the gap at the start of every step; gone when nothing is freed).  With all
# outputs of a plan's stream carved from ONE block, the burst is one event per stream.  An arena is sized by the plan that came
# before it (same shapes): the first plan of a shape allocates tensor by tensor and only measures.
class Arena:
    __slots__ = ("device", "cap", "off", "need", "_i32", "_f32", "buf")

    def __init__(self, nbytes, device):
        """allocates on the CURRENT stream (the stream whose kernels will write the tensors)"""
        self.device, self.cap, self.off, self.need = device, int(nbytes), 0, 0
        self.buf = torch.empty((self.cap,), dtype=torch.uint8, device=device) if self.cap > 0 else None
        self._i32 = self.buf.view(torch.int32) if self.buf is not None else None
        self._f32 = self.buf.view(torch.float32) if self.buf is not None else None

    def take(self, shape, dtype):
        n = 4
        for d in shape:
            n *= d
        n = (n + 255) & ~255
        self.need += n
        if n == 0 or self.buf is None or self.off + n > self.cap:
            return None          # (zero-element shapes and the measuring arena of a first plan: the caller allocates)
        base = self._i32 if dtype is torch.int32 else self._f32
        strides, s = [], 1
        for d in reversed(shape):
            strides.append(s)
            s *= d
        strides.reverse()
        t = torch.as_strided(base, shape, strides, self.off >> 2)
        self.off += n
        return t


import threading

_arena = threading.local()       # (per thread: an arena belongs to the stream its creator was issuing on)


class arena_scope:
    """`with arena_scope(arena):` — _lib.empty() carves from `arena` inside, on this thread"""

    def __init__(self, arena):
        self.arena = arena

    def __enter__(self):
        self.prev = getattr(_arena, "cur", None)
        _arena.cur = self.arena
        return self.arena

    def __exit__(self, *exc):
        _arena.cur = self.prev
        return False


def empty(shape, dtype, device):
    """torch.empty for the OUTPUTS of graph-building ops (int32 / float32): from the active arena when there is one"""
    a = getattr(_arena, "cur", None)
    if a is not None and a.device == device and (dtype is torch.int32 or dtype is torch.float32):
        t = a.take(shape, dtype)
        if t is not None:
            return t
    return torch.empty(shape, dtype=dtype, device=device)
