"""The reference's own regime (reactions_vae.py:52-59: one bundle of <= 21 SMILES per optimizer step) timed through
`trainer.train()`: samples/s with the per-layer-step launch schedule (ppk_seq_persist_mode 0), with the sequence kernels forced to
one row-block persistent launch per pass (mode 1), with the small-batch weights-stationary kernels (mode 2) and with the default (-1), same bundles, same model.
Usage: python tools/small_batch_bench.py [bundles=48] [epochs=3] [modes=0,1,2,-1]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from paccmann_polymer_amd import _C  # noqa: E402
from paccmann_polymer_amd.data_prep import Data  # noqa: E402
from paccmann_polymer_amd.optim import FlatAdam  # noqa: E402
from paccmann_polymer_amd.paccmann_encoder import Encoder  # noqa: E402
from paccmann_polymer_amd.trainer import train  # noqa: E402

n_bundles = int(sys.argv[1]) if len(sys.argv) > 1 else 48
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
modes = [int(m) for m in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0, 1, 2, -1]
rng = np.random.default_rng(5)
loader = []
for b in range(n_bundles):
    n = int(rng.integers(6, 22))                                   # bundle sizes of load_data: anchor + neighbours, <= 21
    seqs = [torch.tensor(np.concatenate([[2], rng.integers(4, 64, size=int(rng.integers(18, 99))), [3]]), dtype=torch.int64) for _ in range(n)]
    edges = [(i, int(rng.integers(0, i))) for i in range(1, n)] + [(int(rng.integers(0, n)), int(rng.integers(0, n))) for _ in range(n // 2)]
    edges = [(u, v) for u, v in edges if u != v]
    loader.append(Data(x=seqs, edge_index=torch.tensor(edges, dtype=torch.int64).T.contiguous(), num_nodes=n))
n_samples = sum(len(d.x) for d in loader)
results = {}
for mode in modes:
    assert _C.lib.ppk_seq_persist_mode(mode) == mode
    torch.manual_seed(0)
    model = Encoder(bench.PARAMS, None, None, batch_size=1, batch_mode='Packed')
    opt = FlatAdam(model.gru_vae.parameters(), lr=1e-4)
    train(0, model, loader[:4], opt, None, graph_gamma=1.0)         # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = [train(e, model, loader, opt, None, graph_gamma=1.0) for e in range(epochs)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    results[mode] = (n_samples * epochs / dt, dt / (epochs * n_bundles) * 1e3, losses[-1])
_C.lib.ppk_seq_persist_mode(-1)
for mode, (sps, ms, loss) in results.items():
    print(f'ppk_seq_persist_mode {mode:2d}: {sps:8.1f} samples/s, {ms:7.3f} ms per bundle step, last epoch loss {loss:.6f}')
if all(m in results for m in (0, 1, 2)):
    print(f'bundles {n_bundles} (6..21 SMILES of 20..100 tokens), epochs {epochs}: forced persistent (1) / per-step launches (0) = {results[1][0] / results[0][0]:.3f}x; weights-stationary (2) / per-step launches (0) = {results[2][0] / results[0][0]:.3f}x')
