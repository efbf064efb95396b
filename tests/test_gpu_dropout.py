"""Training-mode dropout of nn.GRU(dropout = p) inside the K1 sequence kernels (the reference trains with
`model.gru_vae.train()`, reactions_vae.py:44, on params whose `dropout` comes from the pretrained JSON,
paccmann_encoder.py:41-45).  Parity with INJECTED factors against oracle/stackgru_ref.py (SURVEY landmine 10: CPU and
device cannot share noise) under both launch schedules; keep-rate and key properties of the Philox factors."""
import numpy as np
import pytest
import torch

from test_gpu_stackgru import FULL, ODD, SMALL, build_pair, make_batch, prep, rel

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('seq_schedule')]


def factors(rng, n, NT, H, p):
    keep = rng.random((n, NT, H)) >= p
    return torch.tensor(keep / (1.0 - p), dtype=torch.float64)


@pytest.mark.parametrize('params,B,lo,hi,p,seed', [
    (SMALL, 6, 3, 9, 0.3, 0), (ODD, 9, 3, 12, 0.5, 1), (SMALL, 70, 2, 7, 0.1, 2), (FULL, 16, 5, 30, 0.2, 3),
    (dict(ODD, rnn_cell_size=64, n_layers=4), 40, 3, 10, 0.25, 4),
])
def test_train_step_with_injected_dropout_factors_matches_oracle(params, B, lo, hi, p, seed):
    from oracle import stackgru_ref as R
    params = dict(params, dropout=p)
    rng = np.random.default_rng(100 + seed)
    ref, gpu = build_pair(params, seed, torch.float64)
    ref.train()
    gpu.train()
    seqs = make_batch(rng, B, params['vocab_size'], lo, hi)
    enc, dec, tgt = prep(seqs)
    NT, H, L = int(enc.data.numel()), params['rnn_cell_size'], params['n_layers']
    eps = torch.tensor(rng.standard_normal((B, params['latent_dim'])), dtype=torch.float64)
    f_enc, f_dec = factors(rng, L - 1, NT, H, p), factors(rng, L - 1, NT, H, p)
    ref.encoder.drop_factors, ref.decoder.drop_factors = f_enc, f_dec
    gpu.encoder.drop_factors_fn = lambda n, nt, h: f_enc.float()
    gpu.decoder.drop_factors_fn = lambda n, nt, h: f_dec.float()

    loss_r, mu_r, lv_r, z_r = R.train_step_ref(ref, enc, dec, tgt, eps)
    (loss_r + 0.1 * z_r.pow(2).sum()).backward()
    mu, lv = gpu.encode(enc)
    z = gpu.reparameterize(mu, lv, eps.float().cuda()).unsqueeze(0)
    loss = gpu.decode(z, dec, tgt)
    (loss + 0.1 * z.pow(2).sum()).backward()

    assert abs(float(loss.detach()) - float(loss_r.detach())) <= 1e-5 * abs(float(loss_r.detach()))
    assert rel(mu, mu_r) <= 1e-5 and rel(lv, lv_r) <= 1e-5
    refg = dict(ref.named_parameters())
    for name, prm in gpu.named_parameters():
        g_r = refg[name].grad
        if float(g_r.norm()) < 1e-12:
            assert float(prm.grad.norm()) < 1e-9, name
            continue
        assert rel(prm.grad, g_r) <= 1e-4, (name, rel(prm.grad, g_r))
    # and the factors matter: the same step without them gives another loss
    ref.decoder.drop_factors = None
    ref.eval()
    assert abs(float(R.train_step_ref(ref, enc, dec, tgt, eps)[0]) - float(loss_r)) > 1e-6 * abs(float(loss_r))


def test_philox_factors_keep_rate_scale_keys_and_padding():
    from paccmann_polymer_amd import _C
    lib = _C.lib
    layers, NT, H, Hv, p = 2, 20000, 96, 80, 0.3

    def draw(seed, nt=NT):
        out = torch.full((layers, nt, H), float('nan'), device='cuda')
        _C.check(lib.ppk_dropout_factors(seed, p, layers, nt, H, Hv, _C.p(out), _C.stream()))
        return out
    a = draw(1234)
    assert torch.equal(a, draw(1234)) and not torch.equal(a, draw(1235))
    vals = torch.unique(a)
    assert len(vals) == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1.0 / (1.0 - p)) < 1e-6
    assert float(a[:, :, Hv:].abs().max()) == 0.0                       # zero padding of the hidden size stays dropped
    keep = (a[:, :, :Hv] > 0).double()
    n = keep.numel()
    sigma = (p * (1 - p) / n) ** 0.5
    assert abs(float(keep.mean()) - (1 - p)) <= 5 * sigma                # 3.2 M draws: 5 sigma = 1.3e-3
    # per layer / per unit rates (no stuck lanes), and rows are keyed by their index: a prefix of the rows is the same draw
    assert float((keep.mean(dim=(1, 2)) - (1 - p)).abs().max()) <= 6 * (p * (1 - p) / (NT * Hv)) ** 0.5
    assert float((keep.mean(dim=(0, 1)) - (1 - p)).abs().max()) <= 6 * (p * (1 - p) / (layers * NT)) ** 0.5
    assert torch.equal(draw(1234, 5000), a[:, :5000])
    # neighbouring units are uncorrelated
    k = keep - keep.mean()
    corr = float((k[:, :, 1:] * k[:, :, :-1]).mean() / k.pow(2).mean())
    assert abs(corr) <= 5e-3


def test_training_mode_dropout_runs_is_seeded_and_is_off_in_eval():
    from paccmann_polymer_amd import stack_gru_vae as M
    params = dict(SMALL, dropout=0.4, rnn_cell_size=40)
    torch.manual_seed(0)
    vae = M.TeacherVAE(M.StackGRUEncoder(params), M.StackGRUDecoder(params)).cuda()
    rng = np.random.default_rng(0)
    enc, dec, tgt = prep(make_batch(rng, 12, params['vocab_size'], 3, 9))
    eps = torch.zeros(12, params['latent_dim'], device='cuda')

    def loss():
        mu, lv = vae.encode(enc)
        return float(vae.decode(vae.reparameterize(mu, lv, eps).unsqueeze(0), dec, tgt))
    vae.eval()
    e1, e2 = loss(), loss()
    vae.train()
    torch.manual_seed(5)
    t1 = loss()
    torch.manual_seed(5)
    t2 = loss()
    t3 = loss()
    assert e1 == e2 and t1 == t2 and t1 != t3 and t1 != e1
    vae.train()
    mu, lv = vae.encode(enc)
    vae.decode(vae.reparameterize(mu, lv, eps).unsqueeze(0), dec, tgt).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in vae.parameters())
