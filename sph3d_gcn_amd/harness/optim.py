"""The training loop's optimiser (train_s3dis.py:224: tf.train.AdamOptimizer(learning_rate, epsilon=1e-4)) over the flat parameter /
gradient buffers of harness.dist.FlatGradAllReduce: one streaming HIP kernel per step (csrc/optim.hip) with torch.optim.Adam's
arithmetic; on CPU tensors it IS torch.optim.Adam."""
import torch

from .. import _lib


class FlatAdam:
    def __init__(self, flat_param, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.p = flat_param
        self.lr, self.b1, self.b2, self.eps = float(lr), float(betas[0]), float(betas[1]), float(eps)
        self.t = 0
        if flat_param.is_cuda:
            self.m = torch.zeros_like(flat_param.data)
            self.v = torch.zeros_like(flat_param.data)
            self._torch = None
        else:
            self._torch = torch.optim.Adam([flat_param], lr=lr, betas=betas, eps=eps)

    def step(self):
        if self._torch is not None:
            return self._torch.step()
        self.t += 1
        g = self.p.grad
        _lib.check(_lib.lib().sph3d_adam_step(self.p.numel(), _lib.ptr(self.p.data), _lib.ptr(g), _lib.ptr(self.m), _lib.ptr(self.v),
                                              self.lr, self.b1, self.b2, self.eps, self.t, _lib.stream_ptr()))

    def zero_grad(self, set_to_none=False):
        if self.p.grad is not None:
            self.p.grad.zero_()
