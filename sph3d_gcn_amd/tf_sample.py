"""Point sampling — mirrors tf_ops/sampling/tf_sample.py:15-49.

``farthest_point_sample`` is the HIP kernel (custom op ``sph3d::farthest_point_sample``,
no gradient :24).  ``inverse_density_sample`` and ``random_sample`` were stock-TF one-liners in
the reference (:27-49) and are stock-torch one-liners here (RNG-dependent: parity is statistical).
"""
import torch

from . import _lib


def _farthest_point_sample_impl(database: torch.Tensor, npoint: int) -> torch.Tensor:
    _lib.require_device(database)
    if npoint <= 0:
        raise ValueError("FarthestPointSample expects positive npoint")                     # tf_sample.cpp:35
    if database.dim() != 3 or database.shape[2] != 3:
        raise ValueError("FarthestPointSample expects (batch_size,num_points,3) inp shape")  # :40
    database = _lib.f32(database)
    b, n, _ = database.shape
    out = _lib.empty((b, npoint), torch.int32, database.device)
    l = _lib.lib()
    wsb = l.sph3d_farthest_point_sample_workspace(b, n, npoint)
    ws = _lib.scratch(wsb, database.device)
    _lib.check(l.sph3d_farthest_point_sample(b, n, npoint, _lib.ptr(database), _lib.ptr(out), _lib.ptr(ws), wsb,
                                             _lib.stream_ptr()))
    return out


_farthest_point_sample = torch.library.custom_op("sph3d::farthest_point_sample", mutates_args=())(_farthest_point_sample_impl)


@_farthest_point_sample.register_fake
def _(database, npoint):
    return database.new_empty((database.shape[0], npoint), dtype=torch.int32)


def farthest_point_sample(neursize, database):
    """Farthest-point sampling (public signature of tf_sample.py:15-23; note the argument order: count first).

    database [B, N, 3] fp32 -> [B, neursize] int32: index 0 first, then repeatedly the point farthest from the chosen set
    (ties: lower index within the reference's 1024-thread stride, then lower index).  No gradient.
    """
    return _farthest_point_sample_impl(database, int(neursize))


def inverse_density_sample(neursize, probability):
    '''Gumbel-max top-k over log(probability) (tf_sample.py:27-41): `neursize` DISTINCT points per cloud, point i drawn
    with weight probability[i] (build_graph passes the mean sqrt-distance to the neighbours, i.e. sparse regions are
    favoured).  Entries that are not positive and finite (a row whose neighbour count is 0 gives 0/0) get weight zero.'''
    probability = torch.where(torch.isfinite(probability) & (probability > 0), probability, torch.zeros_like(probability))
    logits = torch.log(probability)
    u = torch.rand_like(logits)
    z = -torch.log(-torch.log(u))
    _, neuron_index = torch.topk(logits + z, int(neursize), dim=-1)
    return neuron_index.int()


def random_sample(neursize, database):
    '''uniform sampling with replacement (tf_sample.py:44-49)'''
    batch_size, num_points = database.shape[0], database.shape[1]
    return torch.randint(0, num_points, (batch_size, int(neursize)), dtype=torch.int32, device=database.device)
