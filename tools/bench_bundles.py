"""Bundle sampling for every node of a relation graph: the host statements (data_prep.sample_bundle + relabel_bundle,
numpy global RNG like the reference's load_data, utils.py:155-196) against one launch of ppk_sample_bundles."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paccmann_polymer_amd import data_prep  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rng = np.random.default_rng(0)
# preferential-attachment-like edge list, ~10 edges per node
tgt = (rng.random(10 * N) ** 2 * N).astype(np.int64)
src = rng.integers(0, N, 10 * N)
edges = np.concatenate([np.stack([np.arange(N), (np.arange(N) + 1) % N], 1), np.stack([src, tgt], 1)]).tolist()
t0 = time.perf_counter()
adj = data_prep.adjacency_lists(edges)
rowptr, col = data_prep.adjacency_csr(adj)
t_adj = time.perf_counter() - t0
sub = min(N, 20000)
t0 = time.perf_counter()
for root in range(sub):
    data_prep.relabel_bundle(data_prep.sample_bundle(lambda x: adj[x], root))
t_host = (time.perf_counter() - t0) * N / sub
rp, cl = torch.as_tensor(rowptr).cuda(), torch.as_tensor(col).cuda()
for _ in range(2):
    out = data_prep.sample_bundles_device(rp, cl)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = data_prep.sample_bundles_device(rp, cl)
torch.cuda.synchronize()
t_dev = time.perf_counter() - t0
t0 = time.perf_counter()
host = {k: v.cpu() for k, v in out.items()}
t_copy = time.perf_counter() - t0
print(f'N = {N} nodes, {int(rowptr[-1])} adjacency entries (max degree {int(np.diff(rowptr).max())}); adjacency lists on the host {t_adj:.2f} s')
print(f'host sampler + relabelling (extrapolated from {sub} roots): {t_host:.2f} s = {N / t_host:.0f} bundles/s')
print(f'ppk_sample_bundles, all roots in one launch: {t_dev * 1e3:.2f} ms = {N / t_dev:.3g} bundles/s (+ {t_copy * 1e3:.1f} ms to copy the results to the host)')
print(f'mean bundle: {float(out["n_nodes"].float().mean()):.1f} nodes, {float(out["n_edges"].float().mean()):.1f} edges')
