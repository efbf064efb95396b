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
    assert lib.ppk_seq_persist_mode(1) == 1                   # the variant lives in the persistent kernels (auto: batch >= 1024 only)
    rng = np.random.default_rng(7 + bundles)
    seqs, gs, sizes = bench.synth_batch(rng, bundles, 16)
    enc, dec, tgt, g_flat = bench.to_device_batch(seqs, gs, 'cuda')
    B = len(seqs)
    eps = torch.randn(B, bench.PARAMS['latent_dim'], device='cuda', generator=torch.Generator('cuda').manual_seed(3))
    batch = (enc, dec, tgt, g_flat, sizes, eps, B)
    l32, g32 = one_step(lib, 0, batch)
    l16, g16 = one_step(lib, 1, batch)
    l32b, g32b = one_step(lib, 0, batch)
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
