"""The persistent / per-step crossover at other hidden sizes than the bench architecture (H = 512): a group is H / 32
workgroups, so a narrower model fills the machine later.  Prints one training step (ms) under both schedules."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from paccmann_polymer_amd import _C  # noqa: E402
from paccmann_polymer_amd.graph_constrained_loss import graph_loss_segmented  # noqa: E402
from paccmann_polymer_amd.paccmann_encoder import Encoder  # noqa: E402

for H, L in [(128, 2), (256, 2), (384, 2), (512, 2), (768, 2), (1024, 2), (512, 3), (256, 1)]:
    params = dict(bench.PARAMS, rnn_cell_size=H, n_layers=L)
    torch.manual_seed(0)
    model = Encoder(params, None, None, batch_size=1, batch_mode='Packed')
    model.gru_vae.train()
    for nb in [32, 64, 128, 256]:
        rng = np.random.default_rng(nb)
        seqs, gs, sizes = bench.synth_batch(rng, nb, 16)
        enc, dec, tgt, g_flat = bench.to_device_batch(seqs, gs, 'cuda')
        B = len(seqs)
        model.update_batch_size(B)
        eps = torch.randn(B, params['latent_dim'], device='cuda')
        out = []
        for mode in (0, 1):
            _C.lib.ppk_seq_persist_mode(mode)

            def step():
                for p in model.gru_vae.parameters():
                    p.grad = None
                dl, mu, lv, z = model.train_step(enc, dec, tgt, eps)
                (dl + graph_loss_segmented(z.squeeze(0), g_flat, sizes)).backward()
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                step()
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 3)
        print(f'H {H:5d} L {L} B {B:5d}: per-step launches {out[0]:8.3f} ms, persistent {out[1]:8.3f} ms, ratio {out[1] / out[0]:.3f}', flush=True)
_C.lib.ppk_seq_persist_mode(-1)
