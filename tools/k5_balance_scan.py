import sys, json
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np, torch
import bench_graph as bg
from paccmann_polymer_amd.graph_loss_sparse import SparseRelationGraph, _SparseGraphLoss
rng = np.random.default_rng(1)
N, Z = 1_000_000, 256
und = bg.relation_graph(rng, N, 10)
E = und.shape[1]
attr = torch.tensor(rng.integers(1, 4, size=E).astype(np.float32)).cuda()
graph = SparseRelationGraph(torch.tensor(und).cuda(), attr, N)
z = torch.randn(N, Z, device='cuda', requires_grad=True)
deg = (graph.rowptr[1:] - graph.rowptr[:-1])
print('max deg', int(deg.max()), 'sum deg^2', float((deg.double() ** 2).sum()))
for c in (0.0, 0.5, 1.0, 2.0, 4.0, 8.0, 16.0):
    st = graph.anchor_ranges(8, c)
    ts = []
    for r in range(8):
        a0, a1 = st[r], st[r + 1]
        def fb():
            z.grad = None
            _SparseGraphLoss.apply(z, graph, a0, a1, False).backward()
        ts.append(bg.timeit(fb, iters=3) * 1e3)
    print(c, st[:3], ['%.2f' % t for t in ts], flush=True)
