"""Micro-benchmark of the NT GEMM path (k_nt64 / k_nt32 behind ppk_gemm_f32) at the GRU step shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paccmann_polymer_amd import ops


def bench(M, N, K, iters=20, check=True):
    A = torch.randn(M, K, device='cuda')
    B = torch.randn(N, K, device='cuda') * 0.05
    C = torch.empty(M, N, device='cuda')
    for _ in range(3):
        ops.gemm(A, B, False, True, C=C)
    torch.cuda.synchronize()
    if check:
        ref = A.double() @ B.double().T
        err = (C.double() - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
        assert err < 1e-5, err
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm(A, B, False, True, C=C)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    print(f'M={M:6d} N={N:5d} K={K:5d}  tiles64={(-(-M // 64)) * (-(-N // 64)):6d}  {t * 1e6:8.1f} us  {2.0 * M * N * K / t / 1e12:6.1f} TFLOP/s', flush=True)


if __name__ == '__main__':
    for (N, K) in ((512, 1536), (1536, 512), (512, 512)):
        for M in (1024, 2048, 4096, 8192, 32768, 131072):
            bench(M, N, K)
