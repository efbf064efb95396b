"""GPU parity of the StackGRU VAE step (K1 sequence kernels + fp32-MFMA GEMMs,
K2 cross-entropy, K3 reparameterise + KL) through the C-ABI against the CPU
restatement oracle/stackgru_ref.py (torch nn.GRU / CrossEntropyLoss /
pack_sequence primitives) on identical weights, tokens and injected eps.

Parity versus upstream paccmann_chemistry itself is UNPINNED (package absent
offline); the GRU cell / CE arithmetic is pinned to torch's.
Bars: loss 1e-5 rel, gradients 1e-4 rel (north_star)."""
import numpy as np
import pytest
import torch
from torch.nn.utils.rnn import pack_sequence

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('seq_schedule')]

SMALL = dict(vocab_size=20, embedding_size=16, rnn_cell_size=32, n_layers=2, stack_width=6, stack_depth=5,
             latent_dim=8, dropout=0.0, padding_index=0, batch_size=1, batch_mode='packed')
ODD = dict(vocab_size=33, embedding_size=10, rnn_cell_size=20, n_layers=3, stack_width=7, stack_depth=4,
           latent_dim=5, dropout=0.0, padding_index=0, batch_size=1, batch_mode='packed')
ONE = dict(vocab_size=12, embedding_size=8, rnn_cell_size=96, n_layers=1, stack_width=5, stack_depth=3,
           latent_dim=6, dropout=0.0, padding_index=0, batch_size=1, batch_mode='packed')
FULL = dict(vocab_size=64, embedding_size=128, rnn_cell_size=512, n_layers=2, stack_width=50, stack_depth=50,
            latent_dim=256, dropout=0.0, padding_index=0, batch_size=1, batch_mode='packed')


def make_batch(rng, B, vocab, lo, hi):
    seqs = []
    for _ in range(B):
        n = int(rng.integers(lo, hi + 1))
        body = rng.integers(4, vocab, size=n - 2)
        seqs.append(torch.tensor(np.concatenate([[2], body, [3]]), dtype=torch.int64))
    return seqs


def prep(seqs):
    enc = pack_sequence(seqs, enforce_sorted=False)
    tgt = pack_sequence([torch.cat([s[1:], torch.tensor([0])]) for s in seqs], enforce_sorted=False)
    return enc, enc, tgt


def rel(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def build_pair(params, seed, dtype):
    from oracle import stackgru_ref as R
    from paccmann_polymer_amd import stack_gru_vae as M
    torch.manual_seed(seed)
    ref = R.TeacherVAERef(R.StackGRUEncoderRef(params), R.StackGRUDecoderRef(params)).to(dtype)
    gpu = M.TeacherVAE(M.StackGRUEncoder(params), M.StackGRUDecoder(params))
    gpu.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    return ref, gpu.cuda()


@pytest.mark.parametrize('params,B,lo,hi,seed', [
    (SMALL, 5, 3, 9, 0), (SMALL, 1, 4, 4, 1), (ODD, 7, 3, 12, 2), (SMALL, 9, 5, 5, 3), (FULL, 16, 5, 30, 4),
    (ONE, 6, 3, 8, 7), (ONE, 140, 2, 6, 8),          # one layer: the recurrent product of the only cell rides with the stack heads
    (FULL, 21, 4, 24, 9), (FULL, 32, 3, 12, 10), (FULL, 17, 2, 9, 11), (FULL, 1, 6, 6, 12),      # the reference's bundle sizes: one and two 16-row tiles of the
                                                                                              # weights-stationary kernels (ppk_seq_ws.hip), a second tile that empties early, a single row
])
def test_train_step_matches_oracle_fp64(params, B, lo, hi, seed):
    check_train_step(params, B, lo, hi, seed)


@pytest.mark.parametrize('limit', [0, 1 << 30])
@pytest.mark.parametrize('params,B,lo,hi,seed', [(ODD, 150, 3, 12, 5), (SMALL, 70, 2, 7, 6)])
def test_train_step_both_tile_families(params, B, lo, hi, seed, limit):
    """The NT GEMM kernels come in a big-tile family (k_nt64 / k_ntF: the batch-2048 steps) and a
    small-tile family (k_nt32 / k_ntFs: short tail steps); force each in turn on a ragged batch that
    spans several row tiles."""
    from paccmann_polymer_amd import _C
    old = _C.lib.ppk_nt_small_grid_limit(-2)
    try:
        assert _C.lib.ppk_nt_small_grid_limit(limit) == limit
        check_train_step(params, B, lo, hi, seed)
    finally:
        _C.lib.ppk_nt_small_grid_limit(old)


def check_train_step(params, B, lo, hi, seed):
    from oracle import stackgru_ref as R
    rng = np.random.default_rng(seed)
    ref, gpu = build_pair(params, seed, torch.float64)
    seqs = make_batch(rng, B, params['vocab_size'], lo, hi)
    enc, dec, tgt = prep(seqs)
    eps = torch.tensor(rng.standard_normal((B, params['latent_dim'])), dtype=torch.float64)

    loss_r, mu_r, lv_r, z_r = R.train_step_ref(ref, enc, dec, tgt, eps)
    kl_r = R.kl_divergence(mu_r, lv_r)
    (loss_r + 0.3 * kl_r + 0.1 * z_r.pow(2).sum()).backward()

    mu, lv = gpu.encode(enc)
    z = gpu.reparameterize(mu, lv, eps.float().cuda()).unsqueeze(0)
    loss = gpu.decode(z, dec, tgt)
    (loss + 0.3 * gpu.last_kl + 0.1 * z.pow(2).sum()).backward()

    assert abs(float(loss.detach()) - float(loss_r.detach())) <= 1e-5 * abs(float(loss_r.detach()))
    assert abs(float(gpu.last_kl.detach()) - float(kl_r.detach())) <= 1e-5 * abs(float(kl_r.detach()))
    assert rel(mu, mu_r) <= 1e-5 and rel(lv, lv_r) <= 1e-5 and rel(z, z_r) <= 1e-5
    refg = dict(ref.named_parameters())
    worst = 0.0
    for name, p in gpu.named_parameters():
        g_r = refg[name].grad
        assert p.grad is not None, name
        if float(g_r.norm()) < 1e-12:
            assert float(p.grad.norm()) < 1e-9, name
            continue
        worst = max(worst, rel(p.grad, g_r))
        assert rel(p.grad, g_r) <= 1e-4, (name, rel(p.grad, g_r))
    # padding row of the embedding has no gradient
    assert float(gpu.encoder.embedding.weight.grad[0].abs().max()) == 0.0


def test_bitwise_reproducible():
    rng = np.random.default_rng(11)
    _, gpu = build_pair(SMALL, 11, torch.float64)
    seqs = make_batch(rng, 6, SMALL['vocab_size'], 3, 10)
    enc, dec, tgt = prep(seqs)
    eps = torch.randn(6, SMALL['latent_dim']).cuda()
    outs = []
    for _ in range(2):
        gpu.zero_grad(set_to_none=True)
        mu, lv = gpu.encode(enc)
        z = gpu.reparameterize(mu, lv, eps).unsqueeze(0)
        loss = gpu.decode(z, dec, tgt)
        loss.backward()
        outs.append([loss.detach().clone()] + [p.grad.clone() for p in gpu.parameters()])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_kernels_vs_torch_fp32_small_ops():
    """K2 / K3 in isolation against plain torch fp32 on the GPU."""
    from paccmann_polymer_amd import stack_gru_vae as M
    torch.manual_seed(0)
    logits = torch.randn(301, 37, device='cuda', requires_grad=True)
    tgt = torch.randint(0, 37, (301,), device='cuda')
    loss = M._CrossEntropy.apply(logits, tgt, 0)
    loss.backward()
    l2 = logits.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(l2, tgt, ignore_index=0)
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-6 * abs(float(ref))
    assert rel(logits.grad, l2.grad) <= 1e-5
    mu = torch.randn(40, 9, device='cuda', requires_grad=True)
    lv = torch.randn(40, 9, device='cuda', requires_grad=True)
    eps = torch.randn(40, 9, device='cuda')
    z, kl = M.reparameterize_kl(mu, lv, eps)
    (z.sum() * 0.7 + 2.0 * kl).backward()
    mu2, lv2 = mu.detach().clone().requires_grad_(True), lv.detach().clone().requires_grad_(True)
    z2 = eps * torch.exp(0.5 * lv2) + mu2
    kl2 = -0.5 * torch.sum(1 + lv2 - mu2.pow(2) - lv2.exp())
    (z2.sum() * 0.7 + 2.0 * kl2).backward()
    assert rel(z, z2) <= 1e-6 and abs(float(kl) - float(kl2)) <= 1e-5 * abs(float(kl2))
    assert rel(mu.grad, mu2.grad) <= 1e-5 and rel(lv.grad, lv2.grad) <= 1e-5


def test_rejects_cpu_and_unpacked():
    from paccmann_polymer_amd import stack_gru_vae as M
    enc = M.StackGRUEncoder(SMALL).cuda()
    with pytest.raises(TypeError):
        enc.encoder_train_step(torch.zeros(4, 2, dtype=torch.int64))
    with pytest.raises(IndexError):
        enc.encoder_train_step(pack_sequence([torch.tensor([2, 99, 3])]))


@pytest.mark.parametrize('seed', range(10))
def test_train_step_random_configurations(seed):
    """Random model / batch shapes against the fp64 oracle: hidden sizes off the 32-unit grid, 1..4 layers,
    stack widths across the 32-column padding boundary, batches across the row-tile boundaries."""
    rng = np.random.default_rng(1000 + seed)
    params = dict(vocab_size=int(rng.integers(6, 70)), embedding_size=int(rng.integers(3, 40)),
                  rnn_cell_size=int(rng.choice([8, 20, 32, 33, 64, 72, 96, 130])), n_layers=int(rng.integers(1, 5)),
                  stack_width=int(rng.choice([3, 7, 29, 32, 33, 50])), stack_depth=int(rng.integers(2, 9)),
                  latent_dim=int(rng.integers(2, 20)), dropout=0.0, padding_index=0, batch_size=1, batch_mode='packed')
    B = int(rng.choice([1, 2, 31, 33, 65, 129, 200]))
    lo = int(rng.integers(2, 5))
    hi = lo + int(rng.integers(0, 9))
    check_train_step(params, B, lo, hi, 2000 + seed)


def test_reparam_kl_kernel_matches_reference_twin_golden():
    """K3 against the golden vectors of the in-tree reference twin (cora/vae.py:31-34,50)."""
    import glob
    import os
    from paccmann_polymer_amd import stack_gru_vae as M
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reparam_kl_*.npz')))
    assert len(files) == 3
    for path in files:
        g = np.load(path)
        mu = torch.tensor(g['mu']).cuda().requires_grad_(True)
        logvar = torch.tensor(g['logvar']).cuda().requires_grad_(True)
        z, kl = M.reparameterize_kl(mu, logvar, torch.tensor(g["eps"]).cuda())
        ((z * torch.tensor(g['w']).cuda()).sum() + 0.3 * kl).backward()
        assert rel(z, torch.tensor(g['z'])) <= 1e-6
        assert abs(float(kl) - float(g['kld'])) <= 1e-5 * abs(float(g['kld']))
        assert rel(mu.grad, torch.tensor(g['dmu'])) <= 1e-5 and rel(logvar.grad, torch.tensor(g['dlogvar'])) <= 1e-5


def test_full_size_batch_equals_weighted_halves():
    """Size-independent property at BASELINE config 1b (2048 sequences, ragged 20..100 tokens, the full
    model): the token-mean decoder loss and the KL of the whole batch, and every parameter gradient, equal
    the token-weighted combination of the two half batches.  The halves run through different launch
    shapes (1024-row tiles, other split-K slab counts, other tile families on the tails) than the whole."""
    rng = np.random.default_rng(77)
    _, gpu = build_pair(FULL, 77, torch.float32)
    B = 2048
    seqs = make_batch(rng, B, FULL['vocab_size'], 20, 100)
    eps = torch.tensor(rng.standard_normal((B, FULL['latent_dim'])), dtype=torch.float32).cuda()
    n_tok = lambda idx: sum(len(seqs[i]) for i in idx) - len(idx)        # targets that are not padding
    c = 1e-3

    def run(idx, kl_weight):
        gpu.zero_grad(set_to_none=True)
        enc, dec, tgt = prep([seqs[i] for i in idx])
        mu, lv = gpu.encode(enc)
        z = gpu.reparameterize(mu, lv, eps[idx]).unsqueeze(0)
        loss = gpu.decode(z, dec, tgt)
        (loss + kl_weight * gpu.last_kl).backward()
        return float(loss.detach()), float(gpu.last_kl.detach()), {k: p.grad.detach().double().clone() for k, p in gpu.named_parameters()}

    everything = list(range(B))
    parts = [list(range(0, B // 2)), list(range(B // 2, B))]
    w = [n_tok(p) / n_tok(everything) for p in parts]
    whole = run(everything, c)
    # w_h * (loss_h + (c / w_h) * kl_h) summed over the halves is the whole objective
    halves = [run(p, c / wh) for p, wh in zip(parts, w)]
    assert abs(w[0] * halves[0][0] + w[1] * halves[1][0] - whole[0]) <= 1e-5 * abs(whole[0])
    assert abs(halves[0][1] + halves[1][1] - whole[1]) <= 1e-5 * abs(whole[1])
    for name, g in whole[2].items():
        want = w[0] * halves[0][2][name] + w[1] * halves[1][2][name]
        if float(g.norm()) < 1e-12:
            assert float(want.norm()) < 1e-9, name
            continue
        assert rel(want, g) <= 1e-4, (name, rel(want, g))


@pytest.mark.parametrize('vocab,B', [(575, 40), (3000, 12)])
def test_train_step_large_vocabulary(vocab, B):
    """Vocabulary sizes of real SMILES / SELFIES languages (the pretrained PaccMann language has 575 tokens):
    the cross-entropy rows, the output projection and the token table are far wider than in the other cases."""
    params = dict(SMALL, vocab_size=vocab, embedding_size=24, rnn_cell_size=64)
    check_train_step(params, B, 3, 14, 4000 + vocab)


def test_train_step_long_sequences():
    check_train_step(dict(SMALL, stack_depth=12), 3, 300, 420, 5001)


def test_stack_beyond_the_lds_image_is_refused():
    """The stack kernels hold one sequence's whole stack in LDS; a stack that does not fit is refused with an
    error before anything is launched (no silent fallback)."""
    from paccmann_polymer_amd import stack_gru_vae as M
    params = dict(SMALL, stack_depth=400, stack_width=50)
    enc = M.StackGRUEncoder(params).cuda()
    seqs = make_batch(np.random.default_rng(0), 3, params['vocab_size'], 3, 6)
    with pytest.raises((RuntimeError, ValueError)):
        enc.encoder_train_step(pack_sequence(seqs, enforce_sorted=False))


def test_device_packer_equals_pack_sequence_and_times():
    """The packer whose index arithmetic runs on the device (data_prep._packed_preparation_device) and the default host
    packer: integers bit-equal to pack_sequence at batch 2048 and at a reference-size bundle; prints the time of both
    (no gain from the device: both spend their time turning the Python list into one tensor)."""
    import time
    from torch.nn.utils.rnn import pack_sequence
    from paccmann_polymer_amd import data_prep as DP
    for B, lo, hi in ((2048, 20, 100), (21, 3, 60)):
        rng = np.random.default_rng(B)
        seqs = [torch.tensor(rng.integers(2, 60, size=int(rng.integers(lo, hi + 1)))) for _ in range(B)]
        ref = pack_sequence(seqs, enforce_sorted=False)
        ref_t = pack_sequence([torch.cat([s[1:], torch.tensor([0])]) for s in seqs], enforce_sorted=False)
        for packer in (DP._packed_preparation_device, DP._packed_preparation_vectorised):
            enc, dec, tgt = packer(seqs, 'cuda')
            for got, want in ((enc, ref), (dec, ref), (tgt, ref_t)):
                assert got.data.is_cuda and torch.equal(got.data.cpu(), want.data)
                assert torch.equal(got.batch_sizes, want.batch_sizes) and got.batch_sizes.device.type == 'cpu'
                assert torch.equal(got.sorted_indices.cpu(), want.sorted_indices)
                assert torch.equal(got.unsorted_indices.cpu(), want.unsorted_indices)
        times = {}
        for name, fn in (('device', lambda: DP._packed_preparation_device(seqs, 'cuda')), ('host', lambda: DP._packed_preparation_vectorised(seqs, 'cuda'))):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            times[name] = (time.perf_counter() - t0) / 5 * 1e3
        print(f'packing B={B}: device-side {times["device"]:.2f} ms, host-side {times["host"]:.2f} ms per call')


def test_persistent_launch_in_several_machine_fulls_equals_per_step_launches():
    """More groups than fit the device at once (FULL architecture: 16 workgroups per group, 32 groups resident; batch
    2304 = 36 groups): the persistent path launches a machine-full of groups at a time.  Same loss and gradients as
    the per-step launch schedule on the same batch (short sequences keep it cheap)."""
    from paccmann_polymer_amd import _C
    from paccmann_polymer_amd import stack_gru_vae as M
    lib = _C.lib
    rng = np.random.default_rng(77)
    seqs = make_batch(rng, 2304, FULL['vocab_size'], 3, 9)
    enc, dec, tgt = prep(seqs)

    def dev(p):
        from torch.nn.utils.rnn import PackedSequence
        return PackedSequence(p.data.cuda(), p.batch_sizes, p.sorted_indices.cuda(), p.unsorted_indices.cuda())
    enc, dec, tgt = dev(enc), dev(dec), dev(tgt)
    eps = torch.randn(len(seqs), FULL['latent_dim'], device='cuda', generator=torch.Generator('cuda').manual_seed(5))
    out = {}
    try:
        for mode in (0, 1):
            assert lib.ppk_seq_persist_mode(mode) == mode
            torch.manual_seed(3)
            vae = M.TeacherVAE(M.StackGRUEncoder(FULL), M.StackGRUDecoder(FULL)).cuda()
            vae.train()
            vae.encoder._update_batch_size(len(seqs))
            vae.decoder._update_batch_size(len(seqs))
            mu, logvar = vae.encode(enc)
            z = vae.reparameterize(mu, logvar, eps).unsqueeze(0)
            loss = vae.decode(z, dec, tgt)
            loss.backward()
            torch.cuda.synchronize()
            out[mode] = (float(loss.detach()), {n: p.grad.detach().clone() for n, p in vae.named_parameters() if p.grad is not None})
    finally:
        lib.ppk_seq_persist_mode(-1)
    (l0, g0), (l1, g1) = out[0], out[1]
    assert abs(l0 - l1) <= 1e-5 * abs(l0), (l0, l1)
    for n in g0:
        scale = float(g0[n].abs().max())
        assert float((g0[n] - g1[n]).abs().max()) <= 2e-4 * max(scale, 1e-12), n
