"""GCNConv's dense product [N_nodes, 256] x [256, 512] at configs[2] size through the NN path (weight as stored)
and the NT path (transposed weight copy), forward and the two backward products."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paccmann_polymer_amd import ops


def timeit(fn, iters=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


M, K, N = 2_400_000, 256, 512
x = torch.randn(M, K, device='cuda')
w = torch.randn(K, N, device='cuda') * 0.05
wt = w.t().contiguous()
y = torch.empty(M, N, device='cuda')
dy = torch.randn(M, N, device='cuda')
dx = torch.empty(M, K, device='cuda')
fl = 2.0 * M * N * K
for name, fn in (('fwd NN  x @ w', lambda: ops.gemm(x, w, False, False, C=y)),
                 ('fwd NT  x @ wt^T', lambda: ops.gemm(x, wt, False, True, C=y)),
                 ('bwd dx NT  dy @ w^T', lambda: ops.gemm(dy, w, False, True, C=dx)),
                 ('bwd dx NN  dy @ wt', lambda: ops.gemm(dy, wt, False, False, C=dx)),
                 ('bwd dw TN  x^T @ dy', lambda: ops.gemm(x, dy, True, False))):
    t = timeit(fn)
    print(f'{name:24s} {t * 1e3:7.3f} ms  {fl / t / 1e12:6.1f} TFLOP/s', flush=True)
ref = (x[:1000].double() @ w.double())
ops.gemm(x, wt, False, True, C=y)
print('NT max rel err', float((y[:1000].double() - ref).abs().max() / ref.abs().max()))
