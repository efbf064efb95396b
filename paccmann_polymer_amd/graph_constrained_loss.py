"""Graph-regularisation loss on latent codes — host side of kernel K4.

Mirror of the reference module
`paccmann_polymer/topologically_regularized_models/graph_constrained_loss.py`
(`graph_regularizer_function` :4-10, `compute_graph_dists` :13-22,
`graph_loss` :25-67): same names, argument meaning, return type and error
behaviour.  The arithmetic is one fused HIP forward + one hand-written HIP
backward (csrc/ppk_graphreg.hip) instead of an O(N^3) Python loop.

`graph_loss_segmented` is the batched form used for scale-up:
sum over independent bundles of `graph_loss(z_b, G_b)`.
"""
from typing import Sequence, Union

import numpy as np
import torch

from . import _C


def graph_regularizer_function(delta_ij, di, dj):
    """Reference :4-10.  Scalar helper kept for API compatibility (the fused
    kernel evaluates it in registers; this host version is plain tensor ops)."""
    if delta_ij == 0:
        return torch.pow(di - dj, 2)
    elif delta_ij > 0:
        return torch.relu(dj - di)
    else:
        return torch.relu(di - dj)


class _DenseGraphLoss(torch.autograd.Function):

    @staticmethod
    def forward(ctx, z, g, seg_ptr, mat_ptr, n_seg, n_total, sum_n2, max_n, g_dtype):
        _C.require_gpu(z, g, seg_ptr, mat_ptr)
        z = z.contiguous()
        zdim = z.shape[1]
        ws_bytes = _C.lib.ppk_graphreg_dense_workspace_bytes(n_seg, n_total, sum_n2)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=z.device)
        loss = torch.empty(1, dtype=torch.float32, device=z.device)
        seg_loss = torch.empty(n_seg, dtype=torch.float32, device=z.device)
        _C.check(_C.lib.ppk_graphreg_dense_fwd(
            _C.p(z), z.stride(0), _C.p(g), g_dtype, _C.p(seg_ptr), _C.p(mat_ptr),
            n_seg, n_total, sum_n2, max_n, zdim,
            _C.p(loss), _C.p(seg_loss), _C.p(ws), ws_bytes, _C.stream()))
        ctx.save_for_backward(z, seg_ptr, mat_ptr, ws)
        ctx.sizes = (n_seg, n_total, sum_n2, max_n, zdim, ws_bytes)
        ctx.mark_non_differentiable(seg_loss)
        return loss.reshape(()), seg_loss

    @staticmethod
    def backward(ctx, dloss, _dseg):
        z, seg_ptr, mat_ptr, ws = ctx.saved_tensors
        n_seg, n_total, sum_n2, max_n, zdim, ws_bytes = ctx.sizes
        dloss = dloss.to(torch.float32).reshape(1).contiguous()
        dz = torch.empty_like(z)
        _C.check(_C.lib.ppk_graphreg_dense_bwd(
            _C.p(z), z.stride(0), _C.p(seg_ptr), _C.p(mat_ptr),
            n_seg, n_total, sum_n2, max_n, zdim,
            _C.p(dloss), _C.p(dz), dz.stride(0), _C.p(ws), ws_bytes, _C.stream()))
        return dz, None, None, None, None, None, None, None, None


def _g_to_device(g, device):
    """Keep the reference's comparison semantics: fp32 G compares in fp32,
    anything else (fp64 tensors, numpy arrays, ints) in fp64."""
    if isinstance(g, np.ndarray):
        g = torch.from_numpy(np.ascontiguousarray(g))
    if not torch.is_tensor(g):
        g = torch.as_tensor(g)
    if g.dtype == torch.float32:
        return g.to(device).contiguous(), 0
    return g.to(device=device, dtype=torch.float64).contiguous(), 1


def compute_graph_dists(locations: torch.Tensor) -> torch.Tensor:
    """Reference :13-22 — N-by-N Euclidean distance matrix (fp32), computed by
    the K4 distance kernel (direct differences, exact zero diagonal)."""
    _C.require_gpu(locations)
    z = locations.detach().to(torch.float32).contiguous()
    n = z.shape[0]
    dev = z.device
    seg_ptr = torch.tensor([0, n], dtype=torch.int64, device=dev)
    mat_ptr = torch.tensor([0, n * n], dtype=torch.int64, device=dev)
    g = torch.zeros(n * n, dtype=torch.float32, device=dev)
    ws_bytes = _C.lib.ppk_graphreg_dense_workspace_bytes(1, n, n * n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    seg_loss = torch.empty(1, dtype=torch.float32, device=dev)
    _C.check(_C.lib.ppk_graphreg_dense_fwd(
        _C.p(z), z.stride(0), _C.p(g), 0, _C.p(seg_ptr), _C.p(mat_ptr),
        1, n, n * n, n, z.shape[1], _C.p(loss), _C.p(seg_loss), _C.p(ws), ws_bytes, _C.stream()))
    # D is the first carve of the workspace
    return ws[:4 * n * n].view(torch.float32).reshape(n, n).clone().squeeze()


def graph_loss(z_locations: torch.Tensor, g_dists: Union[torch.Tensor, np.ndarray]):
    """Reference :25-67.  `z_locations` [N, Z] (HIP tensor), `g_dists` [N, N]
    tensor or numpy array.  Returns a 0-dim fp32 tensor on z's device; raises a
    bare `ValueError` on a shape mismatch (:42-43)."""
    if z_locations.dim() != 2:
        raise ValueError()
    n = z_locations.shape[0]
    if tuple(np.shape(g_dists)) != (n, n):
        raise ValueError()
    if n < 2:
        # reference: len() of the squeezed 0-d distance tensor raises TypeError
        raise TypeError('len() of a 0-d tensor')
    _C.require_gpu(z_locations)
    dev = z_locations.device
    g, g_dtype = _g_to_device(g_dists, dev)
    seg_ptr = torch.tensor([0, n], dtype=torch.int64, device=dev)
    mat_ptr = torch.tensor([0, n * n], dtype=torch.int64, device=dev)
    loss, _ = _DenseGraphLoss.apply(
        z_locations.to(torch.float32), g.reshape(-1), seg_ptr, mat_ptr, 1, n, n * n, n, g_dtype)
    return loss


def graph_loss_segmented(
    z_locations: torch.Tensor,
    g_dists: Union[torch.Tensor, np.ndarray, Sequence],
    seg_sizes: Sequence[int],
    return_per_segment: bool = False,
):
    """sum_b graph_loss(z_b, G_b) over consecutive bundles of `seg_sizes` nodes
    (what running the reference step once per bundle with gradient accumulation
    computes: reactions_vae.py:52-96).  `g_dists` is either a list of per-bundle
    [n_b, n_b] matrices or their flat concatenation."""
    sizes = [int(s) for s in seg_sizes]
    n_total = sum(sizes)
    if z_locations.dim() != 2 or z_locations.shape[0] != n_total or not sizes:
        raise ValueError()
    _C.require_gpu(z_locations)
    dev = z_locations.device
    if isinstance(g_dists, (list, tuple)):
        if len(g_dists) != len(sizes):
            raise ValueError()
        for gm, n in zip(g_dists, sizes):
            if tuple(np.shape(gm)) != (n, n):
                raise ValueError()
        if all(torch.is_tensor(x) and x.dtype == torch.float32 for x in g_dists):
            g = torch.cat([x.reshape(-1) for x in g_dists])
        else:
            g = torch.cat([torch.as_tensor(np.asarray(x.cpu() if torch.is_tensor(x) else x),
                                           dtype=torch.float64).reshape(-1) for x in g_dists])
    else:
        g = g_dists
    sum_n2 = sum(n * n for n in sizes)
    if int(np.prod(np.shape(g))) != sum_n2:
        raise ValueError()
    g, g_dtype = _g_to_device(g, dev)
    seg = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum(sizes, out=seg[1:])
    mat = np.zeros(len(sizes) + 1, dtype=np.int64)
    np.cumsum([n * n for n in sizes], out=mat[1:])
    seg_ptr = torch.from_numpy(seg).to(dev)
    mat_ptr = torch.from_numpy(mat).to(dev)
    loss, seg_loss = _DenseGraphLoss.apply(
        z_locations.to(torch.float32), g.reshape(-1), seg_ptr, mat_ptr,
        len(sizes), n_total, sum_n2, max(sizes), g_dtype)
    return (loss, seg_loss) if return_per_segment else loss
