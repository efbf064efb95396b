"""samples/s of the epoch driver `trainer.train_bundled` FROM PYTHON LISTS of token tensors (the reference loader's
form: a list of Data(x = list of 1-D LongTensors, edge_index), reactions_vae.py:190) at bundles_per_step = 128
(batch 2048, BASELINE configs[1]), beside the resident-batch step that bench.py times.  The driver packs group t + 1
on a worker thread while the GPU runs group t; the two numbers should agree to a few percent when the overlap works."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from paccmann_polymer_amd.data_prep import Data  # noqa: E402
from paccmann_polymer_amd.optim import FlatAdam  # noqa: E402
from paccmann_polymer_amd.paccmann_encoder import Encoder  # noqa: E402
from paccmann_polymer_amd.trainer import train_bundled, train_step_bundles  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(0)
loader = []
first_gs = None
for _ in range(STEPS):
    seqs, gs, sizes, edges = bench.synth_batch(rng, K, 16, return_edges=True)
    first_gs = gs if first_gs is None else first_gs
    for b in range(K):
        loader.append(Data(x=seqs[16 * b:16 * (b + 1)], edge_index=torch.tensor(edges[b]), num_nodes=16))
torch.manual_seed(0)
model = Encoder(bench.PARAMS, None, None, batch_size=1, batch_mode='Packed')
opt = FlatAdam(model.gru_vae.parameters(), lr=1e-4)
n_samples = sum(len(d.x) for d in loader)

train_bundled(0, model, loader[:2 * K], opt, None, graph_gamma=1.0, bundles_per_step=K)      # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
train_bundled(1, model, loader, opt, None, graph_gamma=1.0, bundles_per_step=K)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'train_bundled from Python lists: {STEPS} steps x {K} bundles = {n_samples} SMILES in {dt * 1e3:.1f} ms -> {n_samples / dt:.0f} samples/s ({dt / STEPS * 1e3:.2f} ms per step)')

# the resident-batch step (what bench.py times): one group prepared once, stepped STEPS times
seqs = [s for d in loader[:K] for s in d.x]
enc, dec, tgt, g_flat = bench.to_device_batch(seqs, first_gs, 'cuda')
sizes = [16] * K
model.update_batch_size(len(seqs))
for _ in range(2):
    train_step_bundles(model, opt, enc, dec, tgt, g_flat, sizes, 1.0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    train_step_bundles(model, opt, enc, dec, tgt, g_flat, sizes, 1.0)
torch.cuda.synchronize()
dr = time.perf_counter() - t0
print(f'resident batch: {dr / STEPS * 1e3:.2f} ms per step -> {len(seqs) * STEPS / dr:.0f} samples/s; driver / resident = {(dt / STEPS) / (dr / STEPS):.3f}')
