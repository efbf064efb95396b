import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import bench_graph as BG
from paccmann_polymer_amd import gcn
rng = np.random.default_rng(0)
N = 2_400_000
bonds = BG.molecule_batch(rng, N)
nb = bonds.shape[1]
a, b = rng.integers(0, N, size=nb), rng.integers(0, N, size=nb)
keep = a != b
bonds = np.stack([a[keep], b[keep]]).astype(np.int64)
ei_np = np.concatenate([bonds, bonds[::-1]], axis=1)
ei_np[1, : ei_np.shape[1] // 3] = rng.integers(0, 3, size=ei_np.shape[1] // 3)
ei = torch.tensor(ei_np).cuda()
for _ in range(3):
    g = gcn.CSRGraph(ei, N)
torch.cuda.synchronize()
