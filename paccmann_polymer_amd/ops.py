"""Thin host wrappers (no autograd) over the generic libppk.so entry points."""
import torch

from . import _C


def gemm(A, B, transA=False, transB=False, alpha=1.0, beta=0.0, C=None, bias=None, ws=None):
    """C = alpha * op(A) @ op(B) + beta * C + bias  via ppk_gemm_f32.
    `ws=None` allocates the split-K workspace the library asks for (a few output tiles with a very long K run on
    a handful of workgroups without it: 153 ms against 5 ms for a [256, 512] weight gradient over 2.4 M rows);
    `ws=False` runs without one."""
    _C.require_gpu(A, B, C, bias)
    M = A.shape[1] if transA else A.shape[0]
    K = A.shape[0] if transA else A.shape[1]
    N = B.shape[0] if transB else B.shape[1]
    Kb = B.shape[1] if transB else B.shape[0]
    if K != Kb:
        raise ValueError(f'gemm: inner dimensions differ ({K} vs {Kb})')
    if C is None:
        C = torch.empty(M, N, dtype=torch.float32, device=A.device)
        beta = 0.0
    assert A.stride(1) == 1 and B.stride(1) == 1 and C.stride(1) == 1
    if ws is None:
        need = _C.lib.ppk_gemm_f32_workspace_bytes(M, N, K)
        ws = torch.empty(need, dtype=torch.uint8, device=A.device) if need else None
    elif ws is False:
        ws = None
    ws_bytes = 0 if ws is None else ws.numel() * ws.element_size()
    _C.check(_C.lib.ppk_gemm_f32(
        int(transA), int(transB), M, N, K, alpha, _C.p(A), A.stride(0), _C.p(B), B.stride(0), beta,
        _C.p(C), C.stride(0), _C.p(bias), _C.p(ws), ws_bytes, _C.stream()))
    return C
