"""world_size-2 gloo worker for test_gloo_world2_flat_grad_allreduce (CPU, oracle ops)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import torch_ops  # noqa: E402
from sph3d_gcn_amd.harness import dist as hdist  # noqa: E402
from sph3d_gcn_amd.harness import s3dis_net, synth  # noqa: E402


def _loss(blocks, cfg, wrap):
    xyz, label, inner = synth.s3dis_batch(blocks[0], len(blocks), 512, extent=(0.8, 0.8, 1.0))
    pts = torch.from_numpy(xyz)
    model = s3dis_net.SPH3DS3DIS(cfg, device=torch.device("cpu"), seed=7)
    pred, _ = model(pts, is_training=False)       # inference-mode BN: per-cloud results independent of the shard
    flat = wrap(model)
    pred, _ = model(pts, is_training=False)
    return model, flat, model.loss(pred, torch.from_numpy(label), torch.from_numpy(inner))


def grads_for(blocks, cfg):
    """bench.py's path: bucketed flat gradient, each bucket all-reduced as the backward pass produces it"""
    with torch_ops.patched_util():
        _m, flat, loss = _loss(blocks, cfg, lambda m: hdist.FlatGradAllReduce(m.parameters(), bucket_bytes=64 << 10))
        assert len(flat.buckets) >= 3
        flat.backward(loss)
    return flat


def whole_grad(blocks, cfg):
    """plain autograd over the whole batch in one process: the expected sum"""
    with torch_ops.patched_util():
        model, _f, loss = _loss(blocks, cfg, lambda m: None)
        params = [p for p in model.parameters() if p.requires_grad]
        grads = torch.autograd.grad(loss, params, allow_unused=True)
    return torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(grads, params)])


def main():
    rank, world, _ = hdist.init_from_env(backend="gloo")
    assert world == 2
    cfg = s3dis_net.small_config(512)
    cfg.num_sample = [128, 32]
    b, e = hdist.shard_range(4, rank, world)
    flat = grads_for(list(range(b, e)), cfg)
    flat.all_reduce()
    whole = whole_grad([0, 1, 2, 3], cfg)
    torch.testing.assert_close(flat.flat, whole, rtol=2e-4, atol=2e-5)
    other = flat.flat.clone()
    dist.broadcast(other, src=0)
    assert torch.equal(other, flat.flat)
    dist.barrier()
    if rank == 0:
        print("GLOO_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
