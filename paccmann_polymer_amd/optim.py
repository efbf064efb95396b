"""Flat-buffer Adam on the HIP kernel `ppk_adam_step`.

Replaces `OPTIMIZER_FACTORY['adam'](params, lr=lr)` = `torch.optim.Adam`
(reference reactions_vae.py:179-181, step at :96).  All parameters are views
into ONE flat fp32 buffer and all gradients views into ONE flat gradient
buffer, so the optimizer is a single launch per parameter group and the
data-parallel gradient all-reduce is a single RCCL collective over the flat
gradient.  Every parameter starts on a 256-byte boundary of the flat buffer
(the LDS-DMA GEMM kernels want 16-byte aligned operands); the gaps hold zeros
and stay zero.

`FlatAdam` IS a `torch.optim.Optimizer`: the reference hands its optimizer to
`optim.lr_scheduler.StepLR(optimizer, 12, gamma=0.1)` (reactions_vae.py:186-188,
stepped at :111), which only accepts Optimizer instances and drives the
learning rate through `param_groups[i]['lr']` — read here at every step.
"""
import torch

from . import _C


class FlatAdam(torch.optim.Optimizer):

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError('invalid Adam hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        for g in self.param_groups:
            g['params'] = [p for p in g['params'] if p.requires_grad]
        self.params = [p for g in self.param_groups for p in g['params']]
        if not self.params:
            raise ValueError('optimizer got an empty parameter list')
        dev = self.params[0].device
        _C.require_gpu(*self.params)
        self.offsets = []
        self.group_ranges = []                      # flat [start, end) of every parameter group
        n = 0
        for g in self.param_groups:
            start = n
            for p in g['params']:
                self.offsets.append(n)
                n += (p.numel() + 63) // 64 * 64
            self.group_ranges.append((start, n))
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
        self.step_count = 0

    def zero_grad(self, set_to_none: bool = False):
        # gradients stay views of the flat buffer so autograd accumulates in place
        self.flat_grad.zero_()
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + k].view(p.shape)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.step_count += 1
        st = _C.stream()
        for g, (a, b) in zip(self.param_groups, self.group_ranges):
            if b == a:
                continue
            _C.check(_C.lib.ppk_adam_step(
                _C.p(self.flat[a:b]), _C.p(self.flat_grad[a:b]), _C.p(self.exp_avg[a:b]), _C.p(self.exp_avg_sq[a:b]), b - a,
                float(g['lr']), g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], self.step_count, st))
        return loss

    def state_dict(self):
        return dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq,
                    param_groups=[{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups])

    def load_state_dict(self, sd):
        self.step_count = int(sd['step'])
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        for g, s in zip(self.param_groups, sd['param_groups']):
            g.update(s)
