"""Kernel-level cost of ONE rank's share of the sharded sparse graph loss (run under rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import bench_graph as bg
from paccmann_polymer_amd.graph_loss_sparse import SparseRelationGraph, _SparseGraphLoss
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.default_rng(1)
N, Z = 1_000_000, 256
und = bg.relation_graph(rng, N, 10)
E = und.shape[1]
attr = torch.tensor(rng.integers(1, 4, size=E).astype(np.float32)).cuda()
graph = SparseRelationGraph(torch.tensor(und).cuda(), attr, N)
z = torch.randn(N, Z, device='cuda', requires_grad=True)
st = graph.anchor_ranges(8)
a0, a1 = st[rank], st[rank + 1]
for _ in range(5):
    z.grad = None
    _SparseGraphLoss.apply(z, graph, a0, a1, False).backward()
torch.cuda.synchronize()
