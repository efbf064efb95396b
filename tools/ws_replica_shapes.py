"""Training step (encode, decode, graph loss, backward) against batch size for the per-step launches (mode 0), the
row-block persistent kernels (mode 1) and the weights-stationary kernels with replicas (mode 2): where auto mode should
hand over.  Usage: python tools/ws_replica_shapes.py [H=512] [bundles of 16, comma separated]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from paccmann_polymer_amd import _C  # noqa: E402
from paccmann_polymer_amd.graph_constrained_loss import graph_loss_segmented  # noqa: E402
from paccmann_polymer_amd.paccmann_encoder import Encoder  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
params = dict(bench.PARAMS, rnn_cell_size=H)
torch.manual_seed(0)
model = Encoder(params, None, None, batch_size=1, batch_mode='Packed')
model.gru_vae.train()
print(f'# H {H}: ms per training step; bundles of 16 SMILES of 20..100 tokens')
for nb in ([1, 2, 3, 4, 6, 8, 12, 16, 24, 32] if len(sys.argv) < 3 else [int(x) for x in sys.argv[2].split(',')]):
    rng = np.random.default_rng(nb)
    seqs, gs, sizes = bench.synth_batch(rng, nb, 16)
    enc, dec, tgt, g_flat = bench.to_device_batch(seqs, gs, 'cuda')
    B = len(seqs)
    model.update_batch_size(B)
    eps = torch.randn(B, params['latent_dim'], device='cuda')
    out = []
    for mode in (0, 1, 2, 3):
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
        for _ in range(5):
            step()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 5)
    print(f'B {B:5d}: per-step launches {out[0]:8.3f}, row-block persistent {out[1]:8.3f}, weights-stationary {out[2]:8.3f} (looped forward beyond 32 rows: {out[3]:8.3f})   ws / best other {out[2] / min(out[0], out[1]):.3f}')
_C.lib.ppk_seq_persist_mode(-1)
