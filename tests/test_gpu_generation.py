"""Single-step StackGRU call and token generation (paccmann_encoder.py:115-139 -> TeacherVAE.generate) against
the oracle's step function (oracle/stackgru_ref.py; parity unpinned vs upstream)."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('seq_schedule')]

PARAMS = dict(vocab_size=18, embedding_size=9, rnn_cell_size=40, n_layers=2, stack_width=7, stack_depth=4,
              latent_dim=6, dropout=0.0, padding_index=0, batch_size=1, batch_mode='packed')


def build():
    from oracle import stackgru_ref as R
    from paccmann_polymer_amd import stack_gru_vae as M
    torch.manual_seed(5)
    ref = R.TeacherVAERef(R.StackGRUEncoderRef(PARAMS), R.StackGRUDecoderRef(PARAMS)).double()
    gpu = M.TeacherVAE(M.StackGRUEncoder(PARAMS), M.StackGRUDecoder(PARAMS))
    gpu.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    return ref, gpu.cuda().eval()


def test_single_steps_carry_hidden_and_stack():
    ref, gpu = build()
    rng = np.random.default_rng(0)
    B = 5
    hr = torch.tensor(rng.standard_normal((2, B, 40)) * 0.3)
    sr = torch.tensor(rng.standard_normal((B, 4, 7)) * 0.3)
    hg, sg = hr.float().cuda(), sr.float().cuda()
    for step in range(4):
        tok = torch.tensor(rng.integers(1, 18, size=B))
        out_r, hr, sr = ref.decoder(tok.view(1, B), hr, sr)
        out_g, hg, sg = gpu.decoder(tok.view(1, B), hg, sg)
        assert out_g.shape == (1, B, 40) and hg.shape == (2, B, 40) and sg.shape == (B, 4, 7)
        for a, b in ((out_g, out_r), (hg, hr), (sg, sr)):
            assert float((a.double().cpu() - b).norm() / b.norm()) <= 1e-5, step


def test_greedy_generation_matches_oracle_loop():
    from paccmann_polymer_amd.stack_gru_vae import GreedySearch
    ref, gpu = build()
    rng = np.random.default_rng(1)
    B, n = 4, 12
    z = torch.tensor(rng.standard_normal((1, B, 6)))
    got = list(gpu.generate(z.float().cuda(), torch.tensor([2]), torch.tensor([3]), generate_len=n, search=GreedySearch()))
    # the same loop on the oracle
    with torch.no_grad():
        d = ref.decoder
        hidden = d.latent_to_hidden(z.repeat(d.n_layers, 1, 1))
        _, stack = d.init_state(B, torch.float64, 'cpu')
        tok = torch.full((B,), 2, dtype=torch.int64)
        rows = []
        for _ in range(n):
            out, hidden, stack = d(tok.view(1, B), hidden, stack)
            tok = d.output_layer(out.squeeze(0)).argmax(-1)
            rows.append(tok)
        want = torch.stack(rows, 1)
    assert len(got) == B
    for g, w in zip(got, want):
        hit = (w == 3).nonzero()
        w = w[:int(hit[0]) + 1] if hit.numel() else w
        assert g.tolist() == w.tolist()


def _decode_logits_statements(gru_vae, latent_z, start_index, stop_index, device, generate_len=50, temperature=1):
    """The statements of the reference's `decode_logits`
    (experiments/latent_space_exploration/latent_space_map_fns.py:273-314) — the ONLY code in the reference tree that
    shows how the third-party decoder is stepped: `_update_batch_size`, `latent_z.repeat(n_layers, 1, 1)`,
    `latent_to_hidden(latent_z)` applied to the [L, B, Z] tensor, `init_stack`, priming loop, `decoder(token, hidden,
    stack)`, `output_layer(output).squeeze()`, softmax over the tokens.  Kept statement for statement; the two
    deviations: `latent_z` keeps its dtype (the reference casts to float32: the oracle runs in float64) and
    `.detach().numpy()` reads through `.cpu()` (the reference runs this on the CPU)."""
    latent_z = torch.as_tensor(latent_z).view(1, latent_z.shape[0], latent_z.shape[1]).to(device)
    prime_input = torch.tensor([start_index]).to(device)
    batch_size = latent_z.shape[1]
    gru_vae.decoder._update_batch_size(batch_size)
    latent_z = latent_z.repeat(gru_vae.decoder.n_layers, 1, 1)
    hidden = gru_vae.decoder.latent_to_hidden(latent_z)
    stack = gru_vae.decoder.init_stack
    generated_seq = prime_input.repeat(batch_size, 1)
    prime_input = generated_seq.transpose(1, 0).unsqueeze(1)
    for prime_entry in prime_input[:-1]:
        _, hidden, stack = gru_vae.decoder(prime_entry, hidden, stack)
    input_token = prime_input[-1]
    logits_list, probabilities_list = [], []
    for idx in range(generate_len):
        output, hidden, stack = gru_vae.decoder(input_token, hidden, stack)
        logits = gru_vae.decoder.output_layer(output).squeeze()
        probabilities = torch.softmax(logits.div(temperature), dim=0)
        logits_list.append(logits.detach().cpu().numpy())
        probabilities_list.append(probabilities.detach().cpu().numpy())
    return np.stack(logits_list), np.stack(probabilities_list).astype(np.double)


def test_reference_decode_logits_statements_on_hip_decoder_and_oracle():
    """Replays `decode_logits` on the HIP decoder and on oracle/stackgru_ref.py with the same weights: every attribute
    and call of the upstream decoder that the reference's own code uses exists with the shapes it expects, and the
    two agree on the logits of every step (the reference feeds the start token at EVERY step — `input_token` is never
    updated — so 12 steps carry hidden state and stack through 12 stack updates)."""
    ref, gpu = build()
    rng = np.random.default_rng(9)
    z = rng.standard_normal((1, 6))                              # decode_logits requires latent_z.shape[0] == 1 (:271-272)
    with torch.no_grad():
        want_l, want_p = _decode_logits_statements(ref, torch.tensor(z), 2, 3, 'cpu', generate_len=12, temperature=0.7)
        got_l, got_p = _decode_logits_statements(gpu, torch.tensor(z).float(), 2, 3, 'cuda', generate_len=12, temperature=0.7)
    assert got_l.shape == want_l.shape == (12, PARAMS['vocab_size'])
    assert np.linalg.norm(got_l - want_l) <= 1e-5 * np.linalg.norm(want_l)
    assert np.abs(got_p - want_p).max() <= 1e-5 and np.allclose(got_p.sum(1), 1.0, atol=1e-5)
    # the attributes themselves
    gpu.decoder._update_batch_size(7)
    assert gpu.decoder.init_stack.shape == (7, PARAMS['stack_depth'], PARAMS['stack_width']) and gpu.decoder.init_stack.is_cuda
    assert gpu.decoder.init_hidden.shape == (PARAMS['n_layers'], 7, PARAMS['rnn_cell_size']) and float(gpu.decoder.init_hidden.abs().sum()) == 0.0
    assert gpu.encoder.init_stack.shape[0] == gpu.encoder.batch_size


def test_encoder_facade_decode_and_reconstruct():
    from paccmann_polymer_amd.paccmann_encoder import Encoder
    from torch.nn.utils.rnn import pack_sequence
    enc = Encoder(PARAMS, None, None, batch_size=1, batch_mode='Packed', seed=3)
    z = torch.randn(3, 6, device='cuda')
    a = enc.decode(z, generate_len=8)
    b = enc.decode(z, generate_len=8)
    assert a == b and len(a) == 3 and all(t[0] == 2 and 1 <= len(t) - 1 <= 8 for t in a)      # seeded sampling is repeatable
    seqs = [torch.tensor([2, 5, 6, 7, 3]), torch.tensor([2, 9, 3])]
    rec, mu, logvar = enc.reconstruct(pack_sequence(seqs, enforce_sorted=False))
    assert len(rec) == 2 and mu.shape == (2, 6) and all(isinstance(r, str) for r in rec)


def test_encoder_from_files_round_trip(tmp_path):
    """The facade's file interface (paccmann_encoder.py:37-68: params JSON, language file, weights .pt written by
    `gru_vae.save`): a model saved by one Encoder and loaded by another encodes identically; a state dict with the
    oracle's (upstream-named) keys loads strictly.  The language file is JSON here (pytoda's pickle is never loaded)."""
    import json
    from oracle import stackgru_ref as R
    from paccmann_polymer_amd.paccmann_encoder import Encoder
    params = dict(vocab_size=16, embedding_size=8, rnn_cell_size=32, n_layers=2, stack_width=5, stack_depth=4,
                  latent_dim=6, dropout=0.0, padding_index=0)
    alphabet = 'CNOcn()=12'
    pfile, lfile, wfile = (str(tmp_path / n) for n in ('params.json', 'lang.json', 'weights.pt'))
    json.dump(params, open(pfile, 'w'))
    json.dump(dict(token_to_index={c: 4 + i for i, c in enumerate(alphabet)}), open(lfile, 'w'))
    torch.manual_seed(5)
    first = Encoder(pfile, lfile, None, batch_size=1, batch_mode='Packed')
    first.gru_vae.save(wfile)
    second = Encoder(pfile, lfile, wfile, batch_size=1, batch_mode='Packed')
    for smi in ('CCO', 'c1ccccc1N', 'C(=O)O'):
        assert torch.equal(first(smi), second(smi))
        assert first(smi).shape == (params['latent_dim'],)
    ref = R.TeacherVAERef(R.StackGRUEncoderRef(dict(params, batch_size=1, batch_mode='packed')),
                          R.StackGRUDecoderRef(dict(params, batch_size=1, batch_mode='packed')))
    torch.save(ref.state_dict(), wfile)
    second.update_vae_weights(wfile)                       # strict load: same key set and shapes
    mu = second('CCO')
    tokens = torch.tensor(second.smiles_language.smiles_to_token_indexes('CCO'))
    from torch.nn.utils.rnn import pack_sequence
    mu_ref, _ = ref.encode(pack_sequence([tokens]))
    assert torch.allclose(mu.cpu(), mu_ref.detach().reshape(-1), rtol=1e-5, atol=1e-6)
