"""Adam over ONE flat fp32 parameter buffer (harness plumbing: train_s3dis.py:224 uses tf.train.AdamOptimizer).

On the GPU the update is one streaming kernel (csrc/optim.hip: sph3d_adam_step, torch.optim.Adam's arithmetic to 2e-6);
on CPU tensors (oracle-backed tests) it wraps torch.optim.Adam.  The surface a training loop needs from an optimiser is
kept: a settable learning rate (``lr`` / ``param_groups[0]['lr']``: the reference decays it every step), ``state_dict`` /
``load_state_dict`` for checkpoint / resume, ``zero_grad``.
"""
import torch

from .. import _lib


class FlatAdam:
    def __init__(self, flat_param, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if flat_param.dtype != torch.float32 or not flat_param.is_contiguous():
            raise ValueError("FlatAdam needs one contiguous float32 parameter buffer")
        self.p = flat_param
        self.b1, self.b2, self.eps = float(betas[0]), float(betas[1]), float(eps)
        self.t = 0
        # one group, torch-style, so that `for g in opt.param_groups: g['lr'] = ...` works on both branches
        self._groups = [{"params": [flat_param], "lr": float(lr), "betas": (self.b1, self.b2), "eps": self.eps}]
        if flat_param.is_cuda:
            self.m = torch.zeros_like(flat_param.data)
            self.v = torch.zeros_like(flat_param.data)
            self._torch = None
        else:
            self._torch = torch.optim.Adam([flat_param], lr=lr, betas=betas, eps=eps)

    @property
    def param_groups(self):
        # the wrapped optimiser's list is looked up on every access: torch.optim.Optimizer.load_state_dict REPLACES it, an alias
        # taken at construction would go stale after a resume and a per-step learning-rate decay would be ignored
        return self._torch.param_groups if self._torch is not None else self._groups

    @property
    def lr(self):
        return float(self.param_groups[0]["lr"])

    @lr.setter
    def lr(self, value):
        self.param_groups[0]["lr"] = float(value)

    def step(self):
        if self._torch is not None:
            return self._torch.step()
        g = self.p.grad
        if g is None:                      # nothing was back-propagated: torch.optim.Adam skips such parameters too
            return
        if g.dtype != torch.float32 or not g.is_contiguous() or g.device != self.p.device or g.numel() != self.p.numel():
            raise ValueError("FlatAdam: the gradient must be a contiguous float32 buffer on the parameter's device")
        self.t += 1
        _lib.check(_lib.lib().sph3d_adam_step(self.p.numel(), _lib.ptr(self.p.data), _lib.ptr(g), _lib.ptr(self.m), _lib.ptr(self.v),
                                              self.lr, self.b1, self.b2, self.eps, self.t, _lib.stream_ptr()))

    def zero_grad(self, set_to_none=False):
        if self._torch is not None:
            return self._torch.zero_grad(set_to_none=set_to_none)
        if self.p.grad is not None:
            if set_to_none:
                self.p.grad = None
            else:
                self.p.grad.zero_()

    def state_dict(self):
        if self._torch is not None:
            return self._torch.state_dict()
        return {"step": self.t, "exp_avg": self.m.clone(), "exp_avg_sq": self.v.clone(), "lr": self.lr,
                "betas": (self.b1, self.b2), "eps": self.eps}

    def load_state_dict(self, state):
        if self._torch is not None:
            return self._torch.load_state_dict(state)
        self.t = int(state["step"])
        self.m.copy_(state["exp_avg"])
        self.v.copy_(state["exp_avg_sq"])
        self.lr = state.get("lr", self.lr)
