"""Where does the time go in the full-size K6/K7 check (tests/test_gpu_gcn.py)?"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paccmann_polymer_amd import gcn

def tick(msg, t0):
    torch.cuda.synchronize(); print(f'{msg}: {time.time() - t0:.3f} s', flush=True); return time.time()

hubs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.default_rng(9)
N, E, F = 2_400_000, 5_000_000, 256
src = rng.integers(0, N, size=E); dst = rng.integers(0, N, size=E)
if hubs: dst[: E // 3] = rng.integers(0, hubs, size=E // 3)
ei = torch.tensor(np.stack([src, dst])).cuda()
t = tick('start', time.time())
g = gcn.CSRGraph(ei, N, check=True); t = tick('CSRGraph check=True', t)
g = gcn.CSRGraph(ei, N); t = tick('CSRGraph', t)
x = torch.randn(N, F, device='cuda'); t = tick('randn', t)
y = gcn._spmm(g.fwd, x, g.nnz_cap); t = tick('spmm fwd', t)
y = gcn._spmm(g.fwd, x, g.nnz_cap); t = tick('spmm fwd again', t)
y = gcn._spmm(g.rev, x, g.nnz_cap); t = tick('spmm rev', t)
keep = ei[0] != ei[1]; loops = torch.arange(N, device='cuda')
s = torch.cat([ei[0][keep], loops]); d = torch.cat([ei[1][keep], loops]); t = tick('aug', t)
_, order = torch.sort(d, stable=True); t = tick('sort', t)
deg = torch.bincount(d, minlength=N); t = tick('bincount', t)
dinv = deg.double().pow(-0.5); norm = dinv[s] * dinv[d]
v = norm[:, None] * x.double()[s]; t = tick('gather fp64', t)
ref = torch.zeros(N, F, dtype=torch.float64, device='cuda').index_add_(0, d, v); t = tick('index_add fp64', t)
