"""Spherical-kernel bin assignment — mirrors tf_ops/buildkernel/tf_buildkernel.py:10-34.

PyTorch custom op ``sph3d::spherical_kernel`` (no gradient, :34).

The angle function of the binning has two modes (``set_atan2``):

* ``"ocml"`` — THE DEFAULT: ROCm's device-library atan2f, the function the reference's own kernel calls when it is built
  for this GPU.  The bins equal the reference build's bit for bit (tests/test_gpu_parity.py, tests/test_gpu_round3.py: all
  8.4 M level-0 slots of the bench batch); a CPU cannot reproduce them.
* ``"shared"`` — the correctly rounded atan2f of include/sph3d_atan2f.h, the same bits on the GPU and in the CPU oracle.
  It differs from the reference build only for neighbours within an ulp of an angular bin boundary (about 0.07 % of the
  level-0 slots; the exact list on the golden clouds is pinned in tests/golden/ref_gfx950.json).  The oracle-comparison
  tests select it explicitly (tests/conftest.py).
"""
import torch

from . import _lib

DEFAULT_ATAN2 = "ocml"
_atan2 = DEFAULT_ATAN2


def set_atan2(which):
    global _atan2
    if which not in ("shared", "ocml"):
        raise ValueError("atan2 must be 'shared' or 'ocml'")
    _atan2 = which


def _spherical_kernel_impl(database: torch.Tensor, query: torch.Tensor, nn_index: torch.Tensor,
                      nn_count: torch.Tensor, nn_dist: torch.Tensor, radius: float,
                      n_azim: int, p_elev: int, q_radi: int) -> torch.Tensor:
    _lib.require_device(database, query, nn_index, nn_count, nn_dist)
    # SphericalKernelGpuOp::Compute checks (tf_buildkernel.cpp:66-68)
    if database.dim() != 3 or database.shape[2] != 3:
        raise ValueError("Shape of database points requires to be (batch, npoint, 3)")
    if query.dim() != 3 or query.shape[2] != 3:
        raise ValueError("Shape of query points requires to be (batch, mpoint, 3)")
    if nn_index.dim() != 3:
        raise ValueError("Shape of nn_index requires to be of rank 3")
    database, query = _lib.f32(database), _lib.f32(query)
    nn_index, nn_count, nn_dist = _lib.i32(nn_index), _lib.i32(nn_count), _lib.f32(nn_dist)
    B, N, _ = database.shape
    M = query.shape[1]
    K = nn_index.shape[2]
    filt_index = _lib.empty((B, M, K), torch.int32, database.device)
    l = _lib.lib()
    fn = l.sph3d_spherical_kernel_ocml if _atan2 == "ocml" else l.sph3d_spherical_kernel
    _lib.check(fn(
        B, N, M, K, n_azim, p_elev, q_radi, radius, _lib.ptr(database), _lib.ptr(query),
        _lib.ptr(nn_index), _lib.ptr(nn_count), _lib.ptr(nn_dist), _lib.ptr(filt_index), _lib.stream_ptr()))
    return filt_index


_spherical_kernel = torch.library.custom_op("sph3d::spherical_kernel", mutates_args=())(_spherical_kernel_impl)


@_spherical_kernel.register_fake
def _(database, query, nn_index, nn_count, nn_dist, radius, n_azim, p_elev, q_radi):
    return torch.empty_like(nn_index, dtype=torch.int32)


def spherical_kernel(database, query, nn_index, nn_count, nn_dist, radius, kernel=[8, 2, 3]):
    """Kernel bin of every graph edge (public signature of tf_buildkernel.py:10-30).

    database [B, N, >=3], query [B, M, >=3] fp32; nn_index / nn_count / nn_dist as returned by build_sphere_neighbor
    (nn_dist is the square-rooted distance and is compared with `radius` as it is); kernel = [n azimuth sectors,
    p elevation bands, q radial shells].
    returns  filt_index [B, M, K] int32 in [0, n*p*q]: 0 for the query point itself, otherwise 1 + a cell index built
             from (shell, band, sector) as in csrc/sphere_bin.hpp.  No gradient.
    """
    n, p, q = kernel
    database = database[:, :, 0:3]
    query = query[:, :, 0:3]
    return _spherical_kernel_impl(database, query, nn_index, nn_count, nn_dist, float(radius), int(n), int(p), int(q))
