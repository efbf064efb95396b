"""The data-parallel training step with the HIP kernels: two ranks on the one GPU (gloo carries the flat
gradient; on a node the same code runs over RCCL) must leave the same parameters as a single process on
the concatenated batch.  Same weighting code path as bench.py (`dp_train_step`)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PARAMS = dict(vocab_size=24, embedding_size=12, rnn_cell_size=40, n_layers=2, stack_width=6, stack_depth=5,
              latent_dim=8, dropout=0.0, padding_index=0, batch_size=1, batch_mode='packed')


def make_bundles(seed, n_bundles):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_bundles):
        n = int(rng.integers(3, 9))
        seqs = [torch.tensor(np.concatenate([[2], rng.integers(4, 24, size=int(rng.integers(2, 9))), [3]])) for _ in range(n)]
        ei = rng.integers(0, n, size=(2, n + 2))
        out.append((seqs, ei, torch.tensor(rng.standard_normal((n, PARAMS['latent_dim'])), dtype=torch.float32)))
    return out


def run_step(bundles, sync_group_active):
    """One optimizer step on `bundles`; returns (synchronised flat gradient, flat parameter vector afterwards)."""
    from paccmann_polymer_amd import data_prep as DP
    from paccmann_polymer_amd.paccmann_encoder import Encoder
    from paccmann_polymer_amd.optim import FlatAdam
    from paccmann_polymer_amd.distributed import GradSync, global_token_count, dp_train_step
    from paccmann_polymer_amd.graph_constrained_loss import graph_loss_segmented
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = Encoder(PARAMS, None, None, batch_size=1, batch_mode='Packed')
    model.gru_vae.train()
    opt = FlatAdam(model.gru_vae.parameters(), lr=1e-2)
    seqs = [s for b in bundles for s in b[0]]
    enc, dec, tgt = DP.packed_sequential_data_preparation(seqs, input_keep=1., device=dev)
    g_flat, sizes, _ = DP.hop_distance_matrices_device([b[1] for b in bundles], [len(b[0]) for b in bundles], device=dev)
    eps = torch.cat([b[2] for b in bundles]).to(dev)
    model.update_batch_size(len(seqs))
    n_valid = sum(len(s) for s in seqs) - len(seqs)
    total = global_token_count(n_valid) if sync_group_active else n_valid
    sync = GradSync(opt.flat_grad)

    def fwd():
        dec_loss, mu, logvar, z = model.train_step(enc, dec, tgt, eps)
        return dec_loss, graph_loss_segmented(z.squeeze(0), g_flat, sizes)
    dp_train_step(fwd, opt, n_valid, total, 0.7, sync)
    torch.cuda.synchronize()
    params = torch.cat([p.detach().reshape(-1) for p in model.gru_vae.parameters()]).cpu().numpy()
    return opt.flat_grad.detach().cpu().numpy().copy(), params


def _rank_main(rank, world, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from paccmann_polymer_amd.distributed import shard_bundles
        bundles = make_bundles(3, 6)
        mine = [bundles[i] for i in shard_bundles(len(bundles), rank, world)]
        q.put((rank, run_step(mine, True)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_hip_step_equals_single_process():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want_g, want_p = run_step(make_bundles(3, 6), False)
    (g0, p0), (g1, p1) = res[0][1], res[1][1]
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)       # both ranks hold the same gradient / parameters
    err = np.linalg.norm(g0.astype(np.float64) - want_g) / np.linalg.norm(want_g.astype(np.float64))
    assert err <= 1e-4, err                                         # N-rank gradient == 1-rank gradient on the concatenated batch
    assert np.allclose(p0, want_p, rtol=0, atol=2e-4)               # Adam's first step is lr * sign(g): signs agree
