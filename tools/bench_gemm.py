"""Micro-benchmark of ppk_gemm_f32 at the GRU step shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paccmann_polymer_amd import ops

def bench(M, N, K, tA, tB, iters=50):
    A = torch.randn(K if tA else M, M if tA else K, device='cuda')
    B = torch.randn(N if tB else K, K if tB else N, device='cuda')
    C = torch.empty(M, N, device='cuda')
    ws = torch.empty(ops._C.lib.ppk_gemm_f32_workspace_bytes(M, N, K), dtype=torch.uint8, device='cuda')
    for _ in range(5):
        ops.gemm(A, B, tA, tB, C=C, ws=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm(A, B, tA, tB, C=C, ws=ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / ms / 1e9
    # torch (rocBLAS/hipBLASLt) for orientation
    opA = A.T if tA else A; opB = B.T if tB else B
    for _ in range(5): torch.matmul(opA, opB, out=C)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): torch.matmul(opA, opB, out=C)
    e1.record(); torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / iters
    print(f'M={M} N={N} K={K} tA={tA} tB={tB}: ppk {ms*1e3:.1f} us {tf:.1f} TF | torch {ms2*1e3:.1f} us {2.0*M*N*K/ms2/1e9:.1f} TF', flush=True)

for shp in [(2048, 1536, 512, 0, 1), (2048, 1536, 180, 0, 1), (2048, 512, 1536, 0, 0), (1024, 1536, 512, 0, 1),
            (512, 1536, 512, 0, 1), (4096, 4096, 4096, 0, 1), (1536, 512, 123000, 1, 0), (1536, 180, 123000, 1, 0),
            (123000, 64, 512, 0, 1), (2048, 56, 512, 0, 1)]:
    bench(*shp)
