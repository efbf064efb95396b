"""Flat-buffer Adam on the HIP kernel `ppk_adam_step`.

Replaces `OPTIMIZER_FACTORY['adam'](params, lr=lr)` = `torch.optim.Adam`
(reference reactions_vae.py:179-181, step at :96).  All parameters are views
into ONE flat fp32 buffer and all gradients views into ONE flat gradient
buffer, so the optimizer is a single launch and the data-parallel gradient
all-reduce is a single RCCL collective over the flat gradient.  Every
parameter starts on a 256-byte boundary of the flat buffer (the LDS-DMA GEMM
kernels want 16-byte aligned operands); the gaps hold zeros and stay zero.
"""
import torch

from . import _C


class FlatAdam:

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('optimizer got an empty parameter list')
        dev = self.params[0].device
        _C.require_gpu(*self.params)
        self.offsets = []
        n = 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 63) // 64 * 64
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)
                p.grad = self.flat_grad[off:off + k].view(p.shape)
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.step_count = 0

    def zero_grad(self, set_to_none: bool = False):
        # gradients stay views of the flat buffer so autograd accumulates in place
        self.flat_grad.zero_()
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + k].view(p.shape)

    def step(self):
        g = self.param_groups[0]
        self.step_count += 1
        _C.check(_C.lib.ppk_adam_step(
            _C.p(self.flat), _C.p(self.flat_grad), _C.p(self.exp_avg), _C.p(self.exp_avg_sq), self.flat.numel(),
            g['lr'], g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], self.step_count, _C.stream()))

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq,
                    param_groups=[{k: v for k, v in self.param_groups[0].items() if k != 'params'}])

    def load_state_dict(self, sd):
        self.step_count = int(sd['step'])
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        self.param_groups[0].update(sd['param_groups'][0])
