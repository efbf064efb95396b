"""Per-step durations of the sequence kernels inside one process (library profiling table): is the spread between
processes (k_seq_fwd 15.6 vs 16.8 ms per step in same-box runs) also there from step to step?"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from paccmann_polymer_amd import _C  # noqa: E402
from paccmann_polymer_amd.graph_constrained_loss import graph_loss_segmented  # noqa: E402
from paccmann_polymer_amd.optim import FlatAdam  # noqa: E402
from paccmann_polymer_amd.paccmann_encoder import Encoder  # noqa: E402

lib = _C.lib
torch.manual_seed(0)
model = Encoder(bench.PARAMS, None, None, batch_size=1, batch_mode='Packed')
model.gru_vae.train()
opt = FlatAdam(model.gru_vae.parameters(), lr=1e-4)
rng = np.random.default_rng(1234)
seqs, gs, sizes = bench.synth_batch(rng, 128, 16)
enc, dec, tgt, g_flat = bench.to_device_batch(seqs, gs, 'cuda')
B = len(seqs)
model.update_batch_size(B)
eps = torch.randn(B, bench.PARAMS['latent_dim'], device='cuda')
nk = lib.ppk_prof_kind_count()
names = [lib.ppk_prof_kind_name(k).decode() for k in range(nk)]
rows = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    lib.ppk_prof_enable(1)
    opt.zero_grad()
    dl, mu, lv, z = model.train_step(enc, dec, tgt, eps)
    (dl + graph_loss_segmented(z.squeeze(0), g_flat, sizes)).backward()
    opt.step()
    torch.cuda.synchronize()
    lib.ppk_prof_enable(0)
    ms, work, launches = (ctypes.c_double * nk)(), (ctypes.c_double * nk)(), (ctypes.c_int64 * nk)()
    lib.ppk_prof_collect(ms, work, launches)
    d = {names[k]: ms[k] for k in range(nk) if launches[k]}
    rows.append((d.get('k_seq_fwd', 0.0), d.get('k_seq_bwd', 0.0), d.get('k_tn128', 0.0)))
    print(f'step {it:3d}: k_seq_fwd {rows[-1][0]:7.3f}  k_seq_bwd {rows[-1][1]:7.3f}  k_tn128 {rows[-1][2]:7.3f} ms', flush=True)
a = np.array(rows[3:])
print('mean', a.mean(0).round(3), 'min', a.min(0).round(3), 'max', a.max(0).round(3))
