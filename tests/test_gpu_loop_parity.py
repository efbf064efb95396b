"""Round-2 parity cases around the kernels, all against the CPU oracle (oracle/stackgru_ref.py fp64 +
oracle/graph_loss_ref.c + oracle/gcn_ref.py) — never against the HIP path itself:

  * the optimizer-step LOOP of reactions_vae.py:52-111 (`trainer.train`, one bundle per step, StepLR on the fused
    Adam) and its batched form `train_step_bundles`: per-step losses and the parameters after K steps;
  * the FULL model (H 512, stack 50x50, Z 256) at BASELINE configs[0] shape: batch 64 = 4 bundles x 16, len 20..100,
    whole step (encode, reparameterise, decode, segmented graph loss, backward);
  * one DeepGraphInfomax step on packed token input (reactions_infomax.py:116-134) incl. `corruption`;
  * StepLR(FlatAdam) == StepLR(torch.optim.Adam); token id == vocab_size raises IndexError.

Upstream paccmann_chemistry / torch_geometric are absent offline: parity vs those packages is UNPINNED;
the oracle composes torch's own nn.GRU / CrossEntropyLoss / pack_sequence / index_add_.
Bars (north_star): losses 1e-5 rel, gradients 1e-4 rel."""
import numpy as np
import pytest
import torch
from torch.nn.utils.rnn import pack_sequence

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('seq_schedule')]

SMALL = dict(vocab_size=30, embedding_size=16, rnn_cell_size=32, n_layers=2, stack_width=6, stack_depth=5,
             latent_dim=8, dropout=0.0, padding_index=0, batch_size=1, batch_mode='packed')
FULL = dict(vocab_size=64, embedding_size=128, rnn_cell_size=512, n_layers=2, stack_width=50, stack_depth=50,
            latent_dim=256, dropout=0.0, padding_index=0, batch_size=1, batch_mode='packed')


def rel(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def build_pair(params, seed):
    from oracle import stackgru_ref as R
    from paccmann_polymer_amd.paccmann_encoder import Encoder
    torch.manual_seed(seed)
    ref = R.TeacherVAERef(R.StackGRUEncoderRef(params), R.StackGRUDecoderRef(params)).double()
    model = Encoder(params, None, None, batch_size=1, batch_mode='Packed')
    model.gru_vae.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    model.gru_vae.train()
    return ref, model


def make_bundle(rng, n, vocab, lo, hi):
    from paccmann_polymer_amd.data_prep import Data
    seqs = [torch.tensor(np.concatenate([[2], rng.integers(4, vocab, size=int(rng.integers(lo, hi + 1)) - 2), [3]]), dtype=torch.int64)
            for _ in range(n)]
    edges = [[int(rng.integers(0, v)), v] for v in range(1, n)]
    for _ in range(3):
        a, b = rng.integers(0, n, size=2)
        if a != b:
            edges.append([int(a), int(b)])
    return Data(x=seqs, edge_index=torch.tensor(np.asarray(edges, dtype=np.int64).T), num_nodes=n)


def oracle_step(ref, opt, bundles, eps, gamma):
    """One optimizer step of the oracle over `bundles` (K = 1: the reference's step, reactions_vae.py:58-96)."""
    from oracle import graph_loss_oracle as O
    from oracle import stackgru_ref as R
    from paccmann_polymer_amd.data_prep import hop_distance_matrix
    seqs = [s for b in bundles for s in b.x]
    enc = pack_sequence(seqs, enforce_sorted=False)
    tgt = pack_sequence([torch.cat([s[1:], torch.zeros(1, dtype=torch.int64)]) for s in seqs], enforce_sorted=False)
    opt.zero_grad()
    dec_loss, mu, lv, z = R.train_step_ref(ref, enc, enc, tgt, eps)
    z2 = z.squeeze(0)
    gs = [hop_distance_matrix(b.edge_index, len(b.x), order='networkx') for b in bundles]
    gl, dz = O.graph_loss_segmented_c(z2.detach().numpy(), gs, [len(b.x) for b in bundles], acc_fp32=False)
    (dec_loss + gamma * (z2 * torch.tensor(dz)).sum()).backward()       # the C oracle's analytic d graph_loss / dz
    opt.step()
    return float(dec_loss), float(gl)


def assert_params_track(model, ref, start, what):
    """After K Adam steps: parameter DELTAS agree (L2, per tensor).  Adam normalises every element's update to
    ~lr, so the few elements whose gradient sits at rounding level differ by O(lr) whatever the kernel accuracy:
    the bar is on the L2 norm of the delta, 2e-2."""
    refp = dict(ref.named_parameters())
    for name, p in model.gru_vae.named_parameters():
        d_ref = refp[name].detach() - start[name]
        d_gpu = p.detach().double().cpu() - start[name]
        if float(d_ref.norm()) < 1e-12:
            assert float(d_gpu.norm()) < 1e-9, (what, name)
            continue
        assert float((d_gpu - d_ref).norm() / d_ref.norm()) <= 2e-2, (what, name, float((d_gpu - d_ref).norm() / d_ref.norm()))


def test_train_loop_matches_oracle_loop_with_steplr():
    """trainer.train (reference signature, one bundle per step, scheduler stepped per epoch) over 3 epochs x 3 bundles
    against the oracle loop with torch.optim.Adam + StepLR."""
    from paccmann_polymer_amd.optim import FlatAdam
    from paccmann_polymer_amd.trainer import train
    rng = np.random.default_rng(21)
    ref, model = build_pair(SMALL, 21)
    start = {k: v.detach().clone() for k, v in ref.named_parameters()}
    loader = [make_bundle(rng, int(n), SMALL['vocab_size'], 4, 12) for n in (5, 9, 3)]
    gamma, lr, epochs = 0.3, 2e-3, 3
    eps_all = [[torch.tensor(rng.standard_normal((len(b.x), SMALL['latent_dim']))) for b in loader] for _ in range(epochs)]

    opt_r = torch.optim.Adam(ref.parameters(), lr=lr)
    sch_r = torch.optim.lr_scheduler.StepLR(opt_r, 1, gamma=0.5)
    want = []
    for e in range(epochs):
        for b, eps in zip(loader, eps_all[e]):
            want.append(oracle_step(ref, opt_r, [b], eps, gamma))
        sch_r.step()

    opt = FlatAdam(model.gru_vae.parameters(), lr=lr)
    sch = torch.optim.lr_scheduler.StepLR(opt, 1, gamma=0.5)            # reactions_vae.py:186-188 on the fused optimizer
    feed = iter([e for ep in eps_all for e in ep])
    orig = model.gru_vae.reparameterize
    model.gru_vae.reparameterize = lambda mu, lv, eps=None: orig(mu, lv, next(feed).float().cuda())

    class Writer:
        def __init__(self):
            self.rows = {}

        def add_scalar(self, tag, value, step):
            self.rows.setdefault(tag, {})[step] = value
    w = Writer()
    for e in range(epochs):
        train(e, model, loader, opt, sch, graph_gamma=gamma, writer=w)
    assert opt.param_groups[0]['lr'] == pytest.approx(lr * 0.5 ** epochs) == opt_r.param_groups[0]['lr']
    for i, (dl, gl) in enumerate(want):
        assert abs(w.rows['recon_loss'][i] - dl) <= 1e-4 * abs(dl), (i, w.rows['recon_loss'][i], dl)
        assert abs(w.rows['graph_loss'][i] - gl) <= 1e-4 * abs(gl), (i, w.rows['graph_loss'][i], gl)
    assert_params_track(model, ref, start, 'train')


def test_train_step_bundles_matches_oracle_over_steps():
    """The batched step (K bundles per optimizer step) over 4 steps."""
    from paccmann_polymer_amd.data_prep import hop_distance_matrices_device, packed_sequential_data_preparation
    from paccmann_polymer_amd.optim import FlatAdam
    from paccmann_polymer_amd.trainer import train_step_bundles
    rng = np.random.default_rng(22)
    ref, model = build_pair(SMALL, 22)
    start = {k: v.detach().clone() for k, v in ref.named_parameters()}
    gamma, lr, steps = 0.5, 1e-3, 4
    opt_r = torch.optim.Adam(ref.parameters(), lr=lr)
    opt = FlatAdam(model.gru_vae.parameters(), lr=lr)
    for s in range(steps):
        bundles = [make_bundle(rng, int(rng.integers(3, 17)), SMALL['vocab_size'], 3, 14) for _ in range(4)]
        B = sum(len(b.x) for b in bundles)
        eps = torch.tensor(rng.standard_normal((B, SMALL['latent_dim'])))
        dl_r, gl_r = oracle_step(ref, opt_r, bundles, eps, gamma)
        seqs = [x for b in bundles for x in b.x]
        enc, dec, tgt = packed_sequential_data_preparation(seqs, device='cuda')
        g_flat, sizes, _ = hop_distance_matrices_device([b.edge_index for b in bundles], [len(b.x) for b in bundles], device='cuda')
        loss, dl, gl = train_step_bundles(model, opt, enc, dec, tgt, g_flat, sizes, gamma, eps=eps.float().cuda())
        assert abs(float(dl) - dl_r) <= 1e-4 * abs(dl_r), (s, float(dl), dl_r)
        assert abs(float(gl) - gl_r) <= 1e-4 * abs(gl_r), (s, float(gl), gl_r)
    assert_params_track(model, ref, start, 'bundles')


def test_full_model_configs0_step_vs_oracle():
    """BASELINE configs[0] on the HIP path: FULL model, batch 64 = 4 bundles x 16 SMILES of len 20..100, whole step
    vs the fp64 oracle.  (100-step fp32 recurrences through MFMA accumulation order: the case the short-sequence
    tests do not reach.)"""
    from oracle import graph_loss_oracle as O
    from oracle import stackgru_ref as R
    from paccmann_polymer_amd.data_prep import hop_distance_matrix, packed_sequential_data_preparation
    from paccmann_polymer_amd.graph_constrained_loss import graph_loss_segmented
    rng = np.random.default_rng(23)
    ref, model = build_pair(FULL, 23)
    bundles = [make_bundle(rng, 16, FULL['vocab_size'], 20, 100) for _ in range(4)]
    seqs = [x for b in bundles for x in b.x]
    assert max(len(s) for s in seqs) > 90 and min(len(s) for s in seqs) < 30
    B = len(seqs)
    eps = torch.tensor(rng.standard_normal((B, FULL['latent_dim'])))
    gs = [hop_distance_matrix(b.edge_index, 16, order='networkx') for b in bundles]
    sizes = [16] * 4
    gamma = 1.0

    enc = pack_sequence(seqs, enforce_sorted=False)
    tgt = pack_sequence([torch.cat([s[1:], torch.zeros(1, dtype=torch.int64)]) for s in seqs], enforce_sorted=False)
    dl_r, mu_r, lv_r, z_r = R.train_step_ref(ref, enc, enc, tgt, eps)
    mu_r.retain_grad()
    z2 = z_r.squeeze(0)
    gl_r, dz = O.graph_loss_segmented_c(z2.detach().numpy(), gs, sizes, acc_fp32=False)
    (dl_r + gamma * (z2 * torch.tensor(dz)).sum()).backward()

    e, d, t = packed_sequential_data_preparation(seqs, device='cuda')
    dl, mu, lv, z = model.train_step(e, d, t, eps.float().cuda())
    gl = graph_loss_segmented(z.squeeze(0), gs, sizes)
    (dl + gamma * gl).backward()

    assert abs(float(dl) - float(dl_r)) <= 1e-5 * abs(float(dl_r)), (float(dl), float(dl_r))
    assert abs(float(gl) - gl_r) <= 1e-5 * abs(gl_r), (float(gl), gl_r)
    assert rel(mu, mu_r) <= 1e-5 and rel(lv, lv_r) <= 1e-5 and rel(z, z_r) <= 1e-5
    refg = dict(ref.named_parameters())
    bad = []
    for name, p in model.gru_vae.named_parameters():
        g_r = refg[name].grad
        if g_r is None or float(g_r.norm()) < 1e-12:
            continue
        err = rel(p.grad, g_r)
        if name == 'encoder.hidden_to_mu.bias':
            # = column sums of dL/dmu.  graph_loss depends on the latents through differences only, so its share of
            # those sums is EXACTLY zero in exact arithmetic: what is left is the decoder's share (1e-5) beside the
            # rounding residue of summands of size 1e+1..1e+2 — an ill-conditioned sum in any fp32 evaluation, the
            # reference's included.  It is held to 1e-6 of the un-cancelled column sums sum_b |dL/dmu_b| instead.
            scale = float(mu_r.grad.abs().sum(0).norm())
            ok = float((p.grad.detach().double().cpu() - g_r).norm()) <= 1e-6 * scale
        else:
            ok = err <= 1e-4
        if not ok:
            bad.append((name, err))
    assert not bad, bad


def test_infomax_step_on_packed_tokens_vs_oracle():
    """reactions_infomax.py:116-134: packed tokens -> frozen StackGRU encoder (mu) -> GCNConv -> PReLU inside
    DeepGraphInfomax with the reference's PackedSequence corruption; loss and gradients vs stackgru_ref + gcn_ref."""
    from oracle import gcn_ref as G
    from paccmann_polymer_amd import infomax
    from paccmann_polymer_amd.data_prep import packed_sequential_data_preparation
    from paccmann_polymer_amd.gcn import corruption
    rng = np.random.default_rng(24)
    ref, vae = build_pair(SMALL, 24)
    hid = 24
    torch.manual_seed(5)
    enc = infomax.Encoder(paccmann_vae=vae, in_channels=SMALL['latent_dim'], hidden_channels=hid)
    model = infomax.build_infomax(enc, hidden_channels=hid).cuda()
    b = make_bundle(rng, 13, SMALL['vocab_size'], 3, 15)
    model.encoder.update_batch_size(len(b.x))
    seq, _, _ = packed_sequential_data_preparation(b.x, device='cuda')
    ei = b.edge_index

    torch.manual_seed(77)
    pos, neg, s = model(seq, ei.cuda())
    loss = model.loss(pos, neg, s)
    loss.backward()
    assert all(p.grad is None for p in vae.gru_vae.parameters())          # the sequence encoder is frozen

    # oracle: same randperm draw through the same corruption on the CPU copy
    packed_cpu = pack_sequence(b.x, enforce_sorted=False)
    torch.manual_seed(77)
    cor_cpu, _ = corruption(packed_cpu, ei)
    with torch.no_grad():
        x_pos, _ = ref.encode(packed_cpu)
        x_neg, _ = ref.encode(cor_cpu)
    W = enc.conv.weight.detach().cpu().double().requires_grad_(True)
    bb = enc.conv.bias.detach().cpu().double().requires_grad_(True)
    a = enc.prelu.weight.detach().cpu().double().requires_grad_(True)
    Wd = model.weight.detach().cpu().double().requires_grad_(True)

    def f(xx):
        return torch.nn.functional.prelu(G.gcn_conv_ref(xx, ei, W, bb), a)
    pz, nz = f(x_pos), f(x_neg)
    lr = G.dgi_loss_ref(pz, nz, torch.sigmoid(pz.mean(0)), Wd)
    lr.backward()
    assert rel(pos, pz) <= 1e-5 and rel(neg, nz) <= 1e-5
    assert abs(float(loss) - float(lr)) <= 1e-5 * abs(float(lr))
    assert rel(enc.conv.weight.grad, W.grad) <= 1e-4
    assert rel(enc.conv.bias.grad, bb.grad) <= 1e-4
    assert rel(enc.prelu.weight.grad, a.grad) <= 1e-4
    assert rel(model.weight.grad, Wd.grad) <= 1e-4

    # and the reference-shaped loop runs on it
    from paccmann_polymer_amd.optim import FlatAdam
    opt = FlatAdam([p for p in model.parameters()], lr=1e-3)
    out = infomax.train(0, model, [b, make_bundle(rng, 7, SMALL['vocab_size'], 3, 9)], opt, None)
    assert np.isfinite(out)


def test_steplr_on_flat_adam_equals_torch_adam_steplr():
    from paccmann_polymer_amd.optim import FlatAdam
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(s, device='cuda')) for s in ((7, 5), (33,), (4, 4, 3))]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    mine = FlatAdam(ps, lr=1e-2)
    ref = torch.optim.Adam(qs, lr=1e-2)
    assert isinstance(mine, torch.optim.Optimizer)
    s1 = torch.optim.lr_scheduler.StepLR(mine, 2, gamma=0.1)
    s2 = torch.optim.lr_scheduler.StepLR(ref, 2, gamma=0.1)
    for epoch in range(5):
        for _ in range(3):
            mine.zero_grad()
            ref.zero_grad()
            for p, q in zip(ps, qs):
                g = torch.randn_like(q)
                (p * g).sum().backward()
                (q * g).sum().backward()
            mine.step()
            ref.step()
        s1.step()
        s2.step()
        assert mine.param_groups[0]['lr'] == pytest.approx(ref.param_groups[0]['lr'])
    for p, q in zip(ps, qs):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6)
    # two parameter groups with their own learning rates
    ps = [torch.nn.Parameter(torch.randn(9, device='cuda')), torch.nn.Parameter(torch.randn(5, 3, device='cuda'))]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    mine = FlatAdam([dict(params=[ps[0]], lr=1e-2), dict(params=[ps[1]], lr=1e-3)])
    ref = torch.optim.Adam([dict(params=[qs[0]], lr=1e-2), dict(params=[qs[1]], lr=1e-3)])
    for _ in range(3):
        mine.zero_grad()
        ref.zero_grad()
        for p, q in zip(ps, qs):
            g = torch.randn_like(q)
            (p * g).sum().backward()
            (q * g).sum().backward()
        mine.step()
        ref.step()
    for p, q in zip(ps, qs):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6)


def test_out_of_vocabulary_token_raises_index_error():
    """nn.Embedding raises on an id outside [0, V); the kernels index a [V, 3H] table with it."""
    from paccmann_polymer_amd.data_prep import packed_sequential_data_preparation
    _, model = build_pair(SMALL, 3)
    V = SMALL['vocab_size']
    seqs = [torch.tensor([2, 5, V, 3]), torch.tensor([2, 4, 3])]
    e, d, t = packed_sequential_data_preparation(seqs, device='cuda')
    with pytest.raises(IndexError):
        model.train_step(e, d, t)
    # a device token stream of unknown provenance is checked too (once, with one sync)
    raw = pack_sequence([s.cuda() for s in seqs], enforce_sorted=False)
    with pytest.raises(IndexError):
        model.gru_vae.encode(raw)
    neg = pack_sequence([torch.tensor([2, -1, 3]).cuda()], enforce_sorted=False)
    with pytest.raises(IndexError):
        model.gru_vae.encode(neg)


def test_hop_matrices_bad_label_raises_without_sync():
    from paccmann_polymer_amd.data_prep import hop_distance_matrices_device
    with pytest.raises(IndexError):
        hop_distance_matrices_device([np.array([[0, 1], [1, 5]])], [3], device='cuda')


@pytest.mark.parametrize('copies', [2, 3])
def test_max_normaliser_ties_split_evenly_dense_and_sparse(copies):
    """`z_dists.max()` (graph_constrained_loss.py:40) with TIED maxima: torch's backward of a full `max()` splits the
    gradient evenly over the tied entries.  Duplicate latent rows give exact ties (and zero distances between the
    copies, where cdist's backward is 0).  K4 (dense) and K5 (sparse, complete edge set) against torch autograd in
    fp64 through the vectorised restatement of the same loss."""
    from oracle import graph_loss_oracle as O
    from paccmann_polymer_amd.graph_loss_sparse import SparseRelationGraph, graph_loss_sparse
    import paccmann_polymer_amd as P
    rng = np.random.default_rng(11 + copies)
    far = np.array([[0., 0., 0., 0.], [6., 8., 0., 0.]], dtype=np.float32)           # distance 10: the maximum
    near = (rng.integers(-2, 3, size=(5, 4)) + np.array([3., 4., 0., 0.])).astype(np.float32)
    z = np.concatenate([np.tile(far, (copies, 1)), near])
    n = len(z)
    G = rng.integers(1, 4, size=(n, n)).astype(np.float64)
    G = np.triu(G, 1) + np.triu(G, 1).T
    z64 = torch.tensor(z, dtype=torch.float64, requires_grad=True)
    D = torch.cdist(z64, z64)
    assert int((D == D.max()).sum()) == 2 * copies * copies                         # the tie is real
    ref = O.graph_loss_vectorised_torch(z64, torch.tensor(G))
    ref.backward()
    dref = z64.grad.numpy()
    # the C oracle shares the gradient the same way (dense and sparse restatements)
    lo, dzo = O.graph_loss_c(z, G)
    assert abs(lo - float(ref.detach())) <= 1e-5 * abs(float(ref.detach())) and rel(torch.tensor(dzo), torch.tensor(dref)) <= 1e-4
    iu = np.triu_indices(len(z), 1)
    lso, dzso = O.graph_loss_sparse_c(z, np.stack(iu).astype(np.int64), G[iu])
    assert abs(lso - float(ref.detach())) <= 1e-5 * abs(float(ref.detach())) and rel(torch.tensor(dzso), torch.tensor(dref)) <= 1e-4
    # dense
    zt = torch.tensor(z, device='cuda', requires_grad=True)
    loss = P.graph_loss(zt, torch.tensor(G))
    loss.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-5 * abs(float(ref.detach()))
    assert rel(zt.grad, torch.tensor(dref)) <= 1e-4
    # sparse, every pair listed once
    iu = np.triu_indices(n, 1)
    g = SparseRelationGraph(torch.tensor(np.stack(iu).astype(np.int64)).cuda(), torch.tensor(G[iu], dtype=torch.float32).cuda(), n)
    zs = torch.tensor(z, device='cuda', requires_grad=True)
    ls = graph_loss_sparse(zs, g)
    ls.backward()
    assert abs(float(ls.detach()) - float(ref.detach())) <= 1e-5 * abs(float(ref.detach()))
    assert rel(zs.grad, torch.tensor(dref)) <= 1e-4


def _feed_eps(model, eps_list):
    feed = iter(eps_list)
    orig = model.gru_vae.reparameterize
    model.gru_vae.reparameterize = lambda mu, lv, eps=None: orig(mu, lv, next(feed).float().cuda())


def test_train_bundled_one_bundle_per_step_reproduces_train():
    """trainer.train_bundled with bundles_per_step = 1 IS the reference loop (reactions_vae.py:52-96): same
    parameters after two epochs as trainer.train on the same bundles and noise, same scalars to the writer."""
    from paccmann_polymer_amd.optim import FlatAdam
    from paccmann_polymer_amd.trainer import train, train_bundled
    rng = np.random.default_rng(31)
    loader = [make_bundle(rng, int(n), SMALL['vocab_size'], 4, 12) for n in (5, 9, 3, 16)]
    eps = [torch.tensor(rng.standard_normal((len(b.x), SMALL['latent_dim']))) for _ in range(2) for b in loader]

    class Writer:
        def __init__(self):
            self.rows = {}

        def add_scalar(self, tag, value, step):
            self.rows.setdefault(tag, {})[step] = value
    out = []
    for fn in (train, train_bundled):
        _, model = build_pair(SMALL, 31)
        opt = FlatAdam(model.gru_vae.parameters(), lr=2e-3)
        _feed_eps(model, eps)
        w = Writer()
        for e in range(2):
            if fn is train:
                train(e, model, loader, opt, None, graph_gamma=0.3, writer=w)
            else:
                train_bundled(e, model, loader, opt, None, graph_gamma=0.3, bundles_per_step=1, writer=w)
        out.append((w.rows, {k: p.detach().clone() for k, p in model.gru_vae.named_parameters()}))
    (rows_a, pa), (rows_b, pb) = out
    for tag in ('recon_loss', 'graph_loss', 'loss'):
        assert sorted(rows_a[tag]) == sorted(rows_b[tag]) == list(range(8))
        for i in range(8):
            assert abs(rows_a[tag][i] - rows_b[tag][i]) <= 1e-6 * abs(rows_a[tag][i]), (tag, i)
    for k in pa:
        assert float((pa[k] - pb[k]).abs().max()) <= 1e-6 * max(float(pa[k].abs().max()), 1e-12), k


def test_train_bundled_groups_of_four_match_oracle_loop():
    """bundles_per_step = 4 over 10 bundles (groups of 4, 4 and a ragged last group of 2), two epochs with StepLR, against
    the oracle's batched step + torch.optim.Adam; the worker thread prepares group t + 1 while group t runs."""
    from paccmann_polymer_amd.optim import FlatAdam
    from paccmann_polymer_amd.trainer import train_bundled
    rng = np.random.default_rng(32)
    ref, model = build_pair(SMALL, 32)
    start = {k: v.detach().clone() for k, v in ref.named_parameters()}
    loader = [make_bundle(rng, int(rng.integers(3, 17)), SMALL['vocab_size'], 3, 14) for _ in range(10)]
    groups = [loader[0:4], loader[4:8], loader[8:10]]
    gamma, lr, epochs = 0.5, 1e-3, 2
    eps = [torch.tensor(rng.standard_normal((sum(len(b.x) for b in g), SMALL['latent_dim']))) for _ in range(epochs) for g in groups]
    opt_r = torch.optim.Adam(ref.parameters(), lr=lr)
    sch_r = torch.optim.lr_scheduler.StepLR(opt_r, 1, gamma=0.5)
    want = []
    it = iter(eps)
    for e in range(epochs):
        for g in groups:
            want.append(oracle_step(ref, opt_r, g, next(it), gamma))
        sch_r.step()
    opt = FlatAdam(model.gru_vae.parameters(), lr=lr)
    sch = torch.optim.lr_scheduler.StepLR(opt, 1, gamma=0.5)
    _feed_eps(model, eps)

    class Writer:
        rows = {}

        def add_scalar(self, tag, value, step):
            self.rows.setdefault(tag, {})[step] = value
    w = Writer()
    for e in range(epochs):
        train_bundled(e, model, loader, opt, sch, graph_gamma=gamma, bundles_per_step=4, writer=w)
    assert opt.param_groups[0]['lr'] == pytest.approx(lr * 0.25)
    for i, (dl, gl) in enumerate(want):
        assert abs(w.rows['recon_loss'][i] - dl) <= 1e-4 * abs(dl), (i, w.rows['recon_loss'][i], dl)
        assert abs(w.rows['graph_loss'][i] - gl) <= 1e-4 * abs(gl), (i, w.rows['graph_loss'][i], gl)
    assert_params_track(model, ref, start, 'train_bundled')
