"""The oracle's explicit-factor GRU layering (oracle/stackgru_ref.py::gru_layers_with_factors, used to inject dropout
factors into both the oracle and the HIP path) pinned to torch's own nn.GRU: with every factor equal to 1 it must
reproduce `nn.GRU(seq_len = 1)` — outputs, hidden state and gradients — and a whole training step."""
import numpy as np
import torch
from torch.nn.utils.rnn import pack_sequence

PARAMS = dict(vocab_size=17, embedding_size=7, rnn_cell_size=12, n_layers=3, stack_width=5, stack_depth=4,
              latent_dim=6, dropout=0.0, padding_index=0, batch_size=1, batch_mode='packed')


def test_factor_layering_equals_nn_gru_when_all_factors_are_one():
    from oracle import stackgru_ref as R
    torch.manual_seed(0)
    m = R.StackGRURef(PARAMS).double()
    B = 5
    x = torch.randn(1, B, 7 + 5, dtype=torch.float64, requires_grad=True)
    h = torch.randn(3, B, 12, dtype=torch.float64, requires_grad=True)
    out_a, h_a = m.gru(x, h)
    out_b, h_b = m.gru_layers_with_factors(x, h, torch.ones(2, B, 12, dtype=torch.float64))
    assert torch.allclose(out_a, out_b, atol=1e-14) and torch.allclose(h_a, h_b, atol=1e-14)
    ga = torch.autograd.grad((out_a.sum() + h_a.pow(2).sum()), [x, h, m.gru.weight_ih_l1])
    gb = torch.autograd.grad((out_b.sum() + h_b.pow(2).sum()), [x, h, m.gru.weight_ih_l1])
    for a, b in zip(ga, gb):
        assert torch.allclose(a, b, atol=1e-13)


def test_train_step_with_unit_factors_equals_plain_step_and_zero_factors_cut_the_layers():
    from oracle import stackgru_ref as R
    torch.manual_seed(1)
    vae = R.TeacherVAERef(R.StackGRUEncoderRef(PARAMS), R.StackGRUDecoderRef(PARAMS)).double()
    rng = np.random.default_rng(0)
    seqs = [torch.tensor(np.concatenate([[2], rng.integers(4, 17, size=int(n)), [3]])) for n in (4, 7, 3)]
    enc = pack_sequence(seqs, enforce_sorted=False)
    tgt = pack_sequence([torch.cat([s[1:], torch.tensor([0])]) for s in seqs], enforce_sorted=False)
    eps = torch.tensor(rng.standard_normal((3, 6)))
    NT = int(enc.data.numel())
    plain = R.train_step_ref(vae, enc, enc, tgt, eps)[0]
    vae.encoder.drop_factors = vae.decoder.drop_factors = torch.ones(2, NT, 12, dtype=torch.float64)
    ones = R.train_step_ref(vae, enc, enc, tgt, eps)[0]
    assert abs(float(plain) - float(ones)) <= 1e-13
    vae.decoder.drop_factors = torch.zeros(2, NT, 12, dtype=torch.float64)
    cut = R.train_step_ref(vae, enc, enc, tgt, eps)[0]
    assert abs(float(plain) - float(cut)) > 1e-6          # layers >= 1 of the decoder saw zeros
