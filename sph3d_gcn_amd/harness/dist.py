"""Data-parallel plumbing: one process per GPU, batch sharded by cloud, ONE flat gradient all-reduce per
step (the reference is single-GPU).  Every op of the path is independent per cloud, so the
forward/backward data path has no collective; only the parameter gradient is summed (RCCL over xGMI:
backend "nccl" on ROCm; "gloo" in the CPU tests)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_range(total, rank, world):
    """Contiguous [begin, end) of `total` clouds owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class FlatGradAllReduce:
    """All parameters and all gradients live in two flat fp32 buffers (the nn.Parameters are views), so
    the step's collective is a single all-reduce of ~15 MiB for the S3DIS net — one large message per step
    suits xGMI's per-link-bound ring — and the optimiser update is one elementwise kernel over
    ``flat_param`` (pass ``[self.flat_param]`` to the optimiser)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_param = torch.nn.Parameter(torch.empty(n, dtype=torch.float32, device=dev))
        self.flat_param.grad = self.flat
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                self.flat_param.data[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat_param.data[off:off + k].view_as(p)
                p.grad = self.flat[off:off + k].view_as(p)
                off += k

    def broadcast_params(self, src=0):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(self.flat_param.data, src=src)

    def zero(self):
        self.flat.zero_()

    def backward(self, loss):
        """loss.backward() for the flat layout: the parameter gradients are taken with torch.autograd.grad and written
        into the flat buffer by ONE concatenation.  (Through the .grad views every parameter costs its own in-place `add`
        launch — 69 of them, 0.35 ms of 5-us kernels per S3DIS step — and the buffer must be zeroed first.)"""
        grads = torch.autograd.grad(loss, self.params, allow_unused=True)
        parts = [(g if g is not None else torch.zeros_like(p)).reshape(-1) for g, p in zip(grads, self.params)]
        torch.cat(parts, out=self.flat)
        return self.flat

    def all_reduce(self, average=False):
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if average:
                self.flat.div_(dist.get_world_size())
        return self.flat
