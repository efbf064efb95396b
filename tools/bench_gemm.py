"""Micro-benchmark of ppk_gemm_f32 at the GRU step shapes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paccmann_polymer_amd import ops

def bench(M, N, K, tA, tB, iters=30):
    A = torch.randn(K if tA else M, M if tA else K, device='cuda')
    B = torch.randn(N if tB else K, K if tB else N, device='cuda')
    C = torch.empty(M, N, device='cuda')
    ws = torch.empty(ops._C.lib.ppk_gemm_f32_workspace_bytes(M, N, K), dtype=torch.uint8, device='cuda')
    res = []
    for tile in (1, 3, 2):
        os.environ['PPK_GEMM_TILE'] = str(tile)
        for _ in range(3):
            ops.gemm(A, B, tA, tB, C=C, ws=ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.gemm(A, B, tA, tB, C=C, ws=ws)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        res.append(f'{ms*1e3:7.1f}us {2.0*M*N*K/ms/1e9:6.1f}TF')
    print(f'M={M:6d} N={N:5d} K={K:6d} tA={tA} tB={tB}: 64x64 {res[0]} | 128x64 {res[1]} | 128x128 {res[2]}', flush=True)

for shp in [(2048, 1536, 512, 0, 1), (2048, 1536, 180, 0, 1), (2048, 512, 1536, 0, 1), (2048, 512, 1536, 0, 0),
            (1024, 1536, 512, 0, 1), (1024, 512, 1536, 0, 1), (512, 1536, 512, 0, 1), (256, 1536, 512, 0, 1),
            (1536, 512, 123000, 1, 0), (1536, 180, 123000, 1, 0), (123000, 64, 512, 0, 1), (2048, 56, 512, 0, 1)]:
    bench(*shp)
