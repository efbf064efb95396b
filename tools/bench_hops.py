"""Hop-distance matrices of one step's bundles (128 bundles of 16..21 nodes): host BFS loop vs one HIP launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from paccmann_polymer_amd import data_prep as DP
rng = np.random.default_rng(0)
eis, ns = [], []
for _ in range(128):
    n = int(rng.integers(16, 22))
    eis.append(rng.integers(0, n, size=(2, 20)))
    ns.append(n)
t0 = time.perf_counter()
for _ in range(5):
    host = [DP.hop_distance_matrix(e, n) for e, n in zip(eis, ns)]
    g = torch.tensor(np.concatenate([h.reshape(-1) for h in host])).cuda()
torch.cuda.synchronize()
th = (time.perf_counter() - t0) / 5
DP.hop_distance_matrices_device(eis, ns)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    DP.hop_distance_matrices_device(eis, ns)
torch.cuda.synchronize()
td = (time.perf_counter() - t0) / 20
print(f'host loop {th * 1e3:.2f} ms   device call (packing + copy + kernel + status read) {td * 1e3:.2f} ms')
