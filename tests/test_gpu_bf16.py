"""bf16-MFMA variant of K1 (BASELINE.json configs[4]; ppk_mfma_mode(1), include/ppk.h): the products of the persistent
sequence kernels take bf16-rounded operands and accumulate in fp32.  Stated tolerance against the fp32 path
(mode 0, the one that is parity-checked against the fp64 oracle) on a whole training step at the bench
architecture: loss within 2e-3 relative, every gradient tensor within 5e-2 of its largest entry (the
worst is encoder.hidden_to_mu.bias, a sum that cancels almost completely: 3.4e-2; every other tensor below 2e-2) and with
cosine similarity >= 0.999 (0.995 for that one bias).  The fp32 path must come back bit-identical after the mode is switched off again."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu

LOSS_RTOL = 2e-3
GRAD_TOL = 5e-2
GRAD_COS = 0.999


def one_step(lib, mode, batch):
    import bench
    from paccmann_polymer_amd.graph_constrained_loss import graph_loss_segmented
    from paccmann_polymer_amd.paccmann_encoder import Encoder
    enc, dec, tgt, g_flat, sizes, eps, B = batch
    torch.manual_seed(0)
    model = Encoder(bench.PARAMS, None, None, batch_size=1, batch_mode='Packed')
    model.gru_vae.train()
    model.update_batch_size(B)
    assert lib.ppk_mfma_mode(mode) == mode
    try:
        dec_loss, mu, logvar, z = model.train_step(enc, dec, tgt, eps)
        loss = dec_loss + graph_loss_segmented(z.squeeze(0), g_flat, sizes)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        lib.ppk_mfma_mode(0)
    grads = {n: p.grad.detach().double().cpu() for n, p in model.gru_vae.named_parameters() if p.grad is not None}
    return float(loss.detach()), grads


@pytest.mark.parametrize('bundles', [4, 40])
def test_bf16_operand_step_within_stated_tolerance(bundles):
    import bench
    from paccmann_polymer_amd import _C
    lib = _C.lib
    rng = np.random.default_rng(7 + bundles)
    seqs, gs, sizes = bench.synth_batch(rng, bundles, 16)
    enc, dec, tgt, g_flat = bench.to_device_batch(seqs, gs, 'cuda')
    B = len(seqs)
    eps = torch.randn(B, bench.PARAMS['latent_dim'], device='cuda', generator=torch.Generator('cuda').manual_seed(3))
    batch = (enc, dec, tgt, g_flat, sizes, eps, B)
    assert lib.ppk_seq_persist_mode(1) == 1                   # the variant lives in the persistent kernels (auto: batch >= 1024 only)
    try:
        l32, g32 = one_step(lib, 0, batch)
        l16, g16 = one_step(lib, 1, batch)
        l32b, g32b = one_step(lib, 0, batch)
    finally:
        lib.ppk_seq_persist_mode(-1)
    assert l32b == l32 and all(torch.equal(g32[n], g32b[n]) for n in g32), 'fp32 path changed after a bf16 step'
    assert l16 != l32, 'bf16 mode did not take effect'
    assert abs(l16 - l32) <= LOSS_RTOL * abs(l32), (l16, l32)
    worst = 0.0
    for n in g32:
        a, b = g32[n].flatten(), g16[n].flatten()
        scale = float(a.abs().max())
        if scale == 0.0:
            assert float(b.abs().max()) == 0.0, n
            continue
        err = float((a - b).abs().max()) / scale
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        worst = max(worst, err)
        assert err <= GRAD_TOL, (n, err)
        # encoder.hidden_to_mu.bias is a sum that cancels to ~3e-3 of its summands (the graph loss adds exactly zero to
        # it): its direction is the least certain of all tensors
        assert cos >= (0.995 if n == 'encoder.hidden_to_mu.bias' else GRAD_COS), (n, cos)
    print(f'bundles {bundles}: loss fp32 {l32:.6f} bf16 {l16:.6f} (rel {abs(l16 - l32) / abs(l32):.2e}), worst gradient error {worst:.2e} of max')


@pytest.mark.parametrize('M,N,K', [(384, 256, 9001), (1536, 512, 20011), (200, 196, 8200)])
def test_tn_weight_gradient_gemm_is_the_product_of_bf16_rounded_operands(M, N, K):
    """Sharp statement of what mode 1 computes in k_tn128: C = bf16(A)^T . bf16(B) accumulated in fp32 (checked
    against the fp64 product of the rounded operands to fp32-accumulation accuracy); the column sums (the bias
    gradient) stay fp32 sums of the unrounded A."""
    from paccmann_polymer_amd import _C
    lib = _C.lib
    torch.manual_seed(M + N + K)
    A = torch.randn(K, M, device='cuda')
    B = torch.randn(K, N, device='cuda')
    C = torch.full((M, N), float('nan'), device='cuda')
    cs = torch.full((M,), float('nan'), device='cuda')
    wsb = lib.ppk_gemm_f32_workspace_bytes(M, N, K)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device='cuda')
    assert lib.ppk_mfma_mode(1) == 1
    try:
        _C.check(lib.ppk_gemm_tn_colsum_f32(M, N, K, _C.p(A), M, _C.p(B), N, _C.p(C), N, _C.p(cs), _C.p(ws), wsb, _C.stream()))
        torch.cuda.synchronize()
    finally:
        lib.ppk_mfma_mode(0)
    Ab, Bb = A.bfloat16().double(), B.bfloat16().double()
    # the K tail (< 32 rows of the last slab) is multiplied unrounded: allow it its rounding difference
    ref = Ab.T @ Bb
    bound = 2e-6 * ((Ab.abs().T @ Bb.abs()).max().item() + 1.0) + 2.0 ** -7 * 32 * float(A.abs().max() * B.abs().max())
    assert (C.double() - ref).abs().max().item() <= bound
    exact = A.double().T @ B.double()
    assert (C.double() - exact).abs().max().item() > 0.0
    assert (cs.double() - A.double().sum(0)).abs().max().item() <= 2e-6 * A.abs().double().sum(0).max().item()


# ---- BASELINE configs[4] on what one GPU can show (8 x MI355X, 2048 sequences per GPU, bf16 operands) -------------------
def test_bf16_full_per_gpu_workload_equals_weighted_halves_and_tracks_fp32():
    """configs[4]'s PER-GPU share at full size: 2048 ragged sequences (20..100 tokens), bench architecture,
    ppk_mfma_mode(1), default (auto) launch schedule.  (i) Size-independent property: loss, KL and every gradient of
    the whole batch equal the sum over its two halves when each half back-propagates its token-weighted loss
    (w_h * loss_h + c * kl_h).  The weights go INTO the backward pass, not onto its results: then every operand of
    every product has the same value in the whole and in the half (up to fp32 rounding of the seed), so the bf16
    rounding of the operands is the same too — scaling gradients afterwards would compare bf16(c x) with c bf16(x).
    (ii) the stated tolerance against the fp32 step of the same batch (include/ppk.h, ppk_mfma_mode)."""
    from paccmann_polymer_amd import _C
    from test_gpu_stackgru import FULL, build_pair, make_batch, prep, rel
    lib = _C.lib
    rng = np.random.default_rng(177)
    _, gpu = build_pair(FULL, 177, torch.float32)
    B = 2048
    seqs = make_batch(rng, B, FULL['vocab_size'], 20, 100)
    eps = torch.tensor(rng.standard_normal((B, FULL['latent_dim'])), dtype=torch.float32).cuda()
    n_tok = lambda idx: sum(len(seqs[i]) for i in idx) - len(idx)
    c = 1e-3

    def run(idx, loss_weight):
        gpu.zero_grad(set_to_none=True)
        enc, dec, tgt = prep([seqs[i] for i in idx])
        mu, lv = gpu.encode(enc)
        z = gpu.reparameterize(mu, lv, eps[idx]).unsqueeze(0)
        loss = gpu.decode(z, dec, tgt)
        (loss_weight * loss + c * gpu.last_kl).backward()
        return float(loss.detach()), float(gpu.last_kl.detach()), {k: p.grad.detach().double().clone() for k, p in gpu.named_parameters()}

    everything = list(range(B))
    parts = [list(range(0, B // 2)), list(range(B // 2, B))]
    w = [n_tok(p) / n_tok(everything) for p in parts]
    whole32 = run(everything, 1.0)
    assert lib.ppk_mfma_mode(1) == 1
    try:
        whole = run(everything, 1.0)
        halves = [run(p, wh) for p, wh in zip(parts, w)]
    finally:
        lib.ppk_mfma_mode(0)
    assert whole[0] != whole32[0], 'bf16 mode did not take effect at batch 2048 / 1024 (auto schedule)'
    # (i) the halves add up, under bf16 operands
    assert abs(w[0] * halves[0][0] + w[1] * halves[1][0] - whole[0]) <= 1e-5 * abs(whole[0])
    assert abs(halves[0][1] + halves[1][1] - whole[1]) <= 1e-5 * abs(whole[1])
    for name, g in whole[2].items():
        want = halves[0][2][name] + halves[1][2][name]
        if float(g.norm()) < 1e-12:
            assert float(want.norm()) < 1e-9, name
            continue
        # the seeds w_h / n_h and 1 / n agree to fp32 rounding only, which moves a few operands across a bf16 rounding
        # boundary (one part in 2^8 each): measured 2e-4 on the worst tensor, held to 1e-3 — fifty times below the
        # bf16 tolerance of (ii), i.e. the halves add up to far better than the rounding itself
        assert rel(want, g) <= 1e-3, (name, rel(want, g))
    # (ii) stated tolerance against the fp32 step
    assert abs(whole[0] - whole32[0]) <= LOSS_RTOL * abs(whole32[0])
    for name, g in whole32[2].items():
        a, b = g.flatten(), whole[2][name].flatten()
        scale = float(a.abs().max())
        if scale == 0.0:
            continue
        assert float((a - b).abs().max()) / scale <= GRAD_TOL, name
        assert float(torch.dot(a, b) / (a.norm() * b.norm())) >= (0.995 if name == 'encoder.hidden_to_mu.bias' else GRAD_COS), name


# Tolerance of the bf16-operand step against the fp64 ORACLE, stated before it was measured (include/ppk.h): an
# operand rounded to bf16 carries a relative error of at most 2^-9 = 2.0e-3, a product of two at most 3.9e-3; sums of
# such products over K = 64..1536 terms of mixed sign, passed through up to 100 recurrent steps, are held to
#   loss            1e-2 relative
#   mu, logvar, z   2e-2 relative (L2)
#   every gradient  1e-1 of the tensor's largest entry, cosine similarity >= 0.99
ORACLE_LOSS_RTOL = 1e-2
ORACLE_LATENT_RTOL = 2e-2
ORACLE_GRAD_TOL = 1e-1
ORACLE_GRAD_COS = 0.99


def test_bf16_step_vs_fp64_oracle_configs0():
    """The bf16-operand step against the fp64 ORACLE (not against the fp32 HIP path) at BASELINE configs[0] size:
    FULL model, batch 64 = 4 bundles x 16 SMILES of 20..100 tokens, decoder loss + graph loss, all gradients."""
    from oracle import graph_loss_oracle as O
    from oracle import stackgru_ref as R
    from paccmann_polymer_amd import _C
    from paccmann_polymer_amd.data_prep import hop_distance_matrix, packed_sequential_data_preparation
    from paccmann_polymer_amd.graph_constrained_loss import graph_loss_segmented
    from test_gpu_loop_parity import FULL, build_pair, make_bundle, rel
    from torch.nn.utils.rnn import pack_sequence
    lib = _C.lib
    rng = np.random.default_rng(23)
    ref, model = build_pair(FULL, 23)
    bundles = [make_bundle(rng, 16, FULL['vocab_size'], 20, 100) for _ in range(4)]
    seqs = [x for b in bundles for x in b.x]
    B = len(seqs)
    eps = torch.tensor(rng.standard_normal((B, FULL['latent_dim'])))
    gs = [hop_distance_matrix(b.edge_index, 16, order='networkx') for b in bundles]
    sizes = [16] * 4

    enc = pack_sequence(seqs, enforce_sorted=False)
    tgt = pack_sequence([torch.cat([s[1:], torch.zeros(1, dtype=torch.int64)]) for s in seqs], enforce_sorted=False)
    dl_r, mu_r, lv_r, z_r = R.train_step_ref(ref, enc, enc, tgt, eps)
    mu_r.retain_grad()
    z2 = z_r.squeeze(0)
    gl_r, dz = O.graph_loss_segmented_c(z2.detach().numpy(), gs, sizes, acc_fp32=False)
    (dl_r + (z2 * torch.tensor(dz)).sum()).backward()

    e, d, t = packed_sequential_data_preparation(seqs, device='cuda')
    assert lib.ppk_seq_persist_mode(1) == 1 and lib.ppk_mfma_mode(1) == 1
    try:
        dl, mu, lv, z = model.train_step(e, d, t, eps.float().cuda())
        gl = graph_loss_segmented(z.squeeze(0), gs, sizes)
        (dl + gl).backward()
        torch.cuda.synchronize()
    finally:
        lib.ppk_mfma_mode(0)
        lib.ppk_seq_persist_mode(-1)
    assert abs(float(dl) - float(dl_r)) <= ORACLE_LOSS_RTOL * abs(float(dl_r)), (float(dl), float(dl_r))
    assert abs(float(gl) - gl_r) <= ORACLE_LOSS_RTOL * abs(gl_r), (float(gl), gl_r)
    assert float(dl) != pytest.approx(float(dl_r), rel=1e-7), 'bf16 mode did not take effect'
    assert rel(mu, mu_r) <= ORACLE_LATENT_RTOL and rel(lv, lv_r) <= ORACLE_LATENT_RTOL and rel(z, z_r) <= ORACLE_LATENT_RTOL
    refg = dict(ref.named_parameters())
    worst, worst_cos = 0.0, 1.0
    for name, p in model.gru_vae.named_parameters():
        g_r = refg[name].grad
        if g_r is None or float(g_r.norm()) < 1e-12:
            continue
        a, b = g_r.flatten().double(), p.grad.detach().double().cpu().flatten()
        if name == 'encoder.hidden_to_mu.bias':
            # column sums of dL/dmu: the graph loss contributes exactly zero, the rest cancels to ~1e-3 of the summands;
            # held on the un-cancelled scale like the fp32 parity test (tests/test_gpu_loop_parity.py)
            scale = float(mu_r.grad.abs().sum(0).norm())
            assert float((a - b).norm()) <= ORACLE_LATENT_RTOL * scale, name
            continue
        err = float((a - b).abs().max()) / float(a.abs().max())
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        worst, worst_cos = max(worst, err), min(worst_cos, cos)
        assert err <= ORACLE_GRAD_TOL, (name, err)
        assert cos >= ORACLE_GRAD_COS, (name, cos)
    print(f'bf16 vs fp64 oracle: decoder loss rel {abs(float(dl) - float(dl_r)) / abs(float(dl_r)):.2e}, graph loss rel {abs(float(gl) - gl_r) / abs(gl_r):.2e}, '
          f'mu rel {rel(mu, mu_r):.2e}, worst gradient error {worst:.2e} of max, worst cosine {worst_cos:.5f}')
