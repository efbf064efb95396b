"""Single-node data parallelism: one process per GPU, RCCL over xGMI through
`torch.distributed` (backend "nccl" on ROCm; "gloo" for the CPU tests).

The reference has no distributed code; the contract is
"N-GPU result == 1-GPU result on the concatenated batch" (SURVEY §8e):
bundles are never split across ranks (faithful bundle mode), so the only
collective on the data path is ONE all-reduce(SUM) of the flat gradient.
  global loss = (sum_r n_r * dec_r) / (sum_r n_r) + gamma * sum_r gr_r
so each rank back-propagates  dec_r * (n_r / n_total) + gamma * gr_r  and the
gradients are summed (not averaged).
"""
from typing import List, Sequence

import torch
import torch.distributed as dist


def shard_bundles(n_bundles: int, rank: int, world: int) -> List[int]:
    """Round-robin bundle ownership: bundle b belongs to rank b % world."""
    return list(range(rank, n_bundles, world))


def global_token_count(local_valid_tokens: int, group=None) -> int:
    """sum over ranks of the number of non-pad decoder targets (host integers)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return int(local_valid_tokens)
    t = torch.tensor([int(local_valid_tokens)], dtype=torch.int64)
    if dist.get_backend(group) == 'nccl':
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


class GradSync:
    """all-reduce(SUM) of one flat gradient buffer (C1).  xGMI is point-to-point
    (7 links x ~153 GB/s per GPU): one ~23 MB collective per step lets RCCL pick
    its direct algorithm instead of many small ring launches."""

    def __init__(self, flat_grad: torch.Tensor, group=None):
        self.flat_grad = flat_grad
        self.group = group
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1

    def __call__(self):
        if not self.active:
            return
        if self.flat_grad.is_cuda and dist.get_backend(self.group) == 'gloo':
            # rehearsals with several ranks on one GPU: gloo's device-tensor all-reduce hangs on buffers of this
            # size (both ranks inside all_reduce, 23 MB, PyTorch 2.10 ROCm), so the sum goes through host memory
            host = self.flat_grad.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            self.flat_grad.copy_(host)
            return
        dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)


def dp_train_step(step_fn, optimizer, local_valid_tokens: int, total_valid_tokens: int, graph_gamma: float,
                  sync: GradSync):
    """Runs `step_fn() -> (decoder_loss, gr_loss)` (autograd tensors) on the local
    shard and applies the data-parallel weighting described in the module docstring."""
    optimizer.zero_grad()
    decoder_loss, gr_loss = step_fn()
    w = float(local_valid_tokens) / float(total_valid_tokens)
    loss = decoder_loss * w + graph_gamma * gr_loss
    loss.backward()
    sync()
    optimizer.step()
    return loss.detach(), decoder_loss.detach(), gr_loss.detach()
