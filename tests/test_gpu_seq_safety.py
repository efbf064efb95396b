"""Safety behaviour of the persistent sequence kernels through the C-ABI (round-2 advisor findings):

* a hand-off that never arrives must be VISIBLE in everything the caller consumes (not a quiet PPK_OK with
  half-written buffers): `libppk_spin.so` is the same library built with PPK_SPIN_LIMIT=1, where every wait that
  finds its counter short gives up at once; all outputs must then be NaN and the process must end normally;
* a C-ABI caller that passes wider paddings (ks / ldcs) than the workspace query assumes must not reach the
  persistent kernels (their workspace is sized for the minimal paddings): it takes the per-step launches and
  produces the same numbers.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_TIMEOUT_CHILD = r'''
import sys, numpy as np, torch
from torch.nn.utils.rnn import pack_sequence
from paccmann_polymer_amd import _C
from paccmann_polymer_amd import stack_gru_vae as M
assert _C.LIB_PATH.endswith('libppk_spin.so'), _C.LIB_PATH
MODE, NSEQ = int(sys.argv[1]), int(sys.argv[2])
assert _C.lib.ppk_seq_persist_mode(MODE) == MODE
params = dict(vocab_size=64, embedding_size=128, rnn_cell_size=512, n_layers=2, stack_width=50, stack_depth=50,
              latent_dim=256, dropout=0.0, padding_index=0, batch_size=1, batch_mode='packed')
torch.manual_seed(0)
vae = M.TeacherVAE(M.StackGRUEncoder(params), M.StackGRUDecoder(params)).cuda()
rng = np.random.default_rng(0)
seqs = [torch.tensor(np.concatenate([[2], rng.integers(4, 64, size=int(rng.integers(8, 20))), [3]])) for _ in range(NSEQ)]
enc = pack_sequence(seqs, enforce_sorted=False)
tgt = pack_sequence([torch.cat([s[1:], torch.tensor([0])]) for s in seqs], enforce_sorted=False)
mu, lv = vae.encode(enc)
z = vae.reparameterize(mu, lv, torch.zeros_like(mu)).unsqueeze(0)
loss = vae.decode(z, enc, tgt)
loss.backward()
torch.cuda.synchronize()
print('MU_NAN', bool(torch.isnan(mu).all()))
print('LOSS_NAN', bool(torch.isnan(loss)))
g = vae.encoder.gru.weight_hh_l0.grad
print('GRAD_NAN', bool(torch.isnan(g).any()))
'''


@pytest.mark.parametrize('mode,nseq', [(1, 96), (2, 12), (2, 27)])
def test_lost_handoff_poisons_every_output(mode, nseq):
    """mode 1: the row-block persistent kernels; mode 2 with <= 32 sequences: the weights-stationary kernels (ppk_seq_ws.hip)."""
    lib = os.path.join(ROOT, 'paccmann_polymer_amd', 'libppk_spin.so')
    assert os.path.exists(lib), 'libppk_spin.so missing: run make -C paccmann_polymer_amd/csrc'
    env = dict(os.environ, PPK_LIB_PATH=lib, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    r = subprocess.run([sys.executable, '-c', _TIMEOUT_CHILD, str(mode), str(nseq)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'MU_NAN True' in r.stdout and 'LOSS_NAN True' in r.stdout and 'GRAD_NAN True' in r.stdout, r.stdout + r.stderr


def _seq_fwd_raw(lib, _C, T, bs, B, NT, V, H, L, W, D, tokens, table, wx0s, ks, w_ih, w_hh, b_ih, b_hh, wcs, bcs, h0, ldcs):
    f32 = dict(dtype=torch.float32, device='cuda')
    # poisoned tails: a kernel that wrote past a buffer sized for the minimal paddings would be seen
    XS = torch.full((NT, ks), float('nan'), **f32)
    Hs = torch.empty(L, NT, H, **f32)
    HP = torch.empty(L, NT, H, **f32)
    gates = torch.empty(L, NT, 4 * H, **f32)
    S = torch.empty(NT, D, W, **f32)
    CS = torch.full((NT, ldcs), float('nan'), **f32)
    wsb = lib.ppk_stackgru_seq_fwd_workspace_bytes(B, H, L, W)
    guard = 1 << 20
    ws = torch.zeros(wsb + guard, dtype=torch.uint8, device='cuda')
    ws[wsb:] = 0xA5

    def arr(ts):
        return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    bs_host = torch.tensor(bs, dtype=torch.int64)
    _C.check(lib.ppk_stackgru_seq_fwd(
        T, ctypes.c_void_p(bs_host.data_ptr()), B, NT, V, H, L, W, D,
        _C.p(tokens), _C.p(table), _C.p(wx0s), ks, arr(w_ih), arr(w_hh), arr(b_ih), arr(b_hh), _C.p(wcs), _C.p(bcs), _C.p(h0), None,
        _C.p(XS), _C.p(Hs), _C.p(HP), _C.p(gates), _C.p(S), _C.p(CS), ldcs, None, None, _C.p(ws), wsb, _C.stream()))
    torch.cuda.synchronize()
    assert bool((ws[wsb:] == 0xA5).all()), 'the entry point wrote past the workspace it asked for'
    return Hs, S, CS, XS


@pytest.mark.parametrize('mode,B', [(1, 70), (2, 20)])
def test_wider_paddings_than_the_workspace_query_take_the_per_step_path(mode, B):
    """W = 20: minimal ks = ldcs = 32.  ks = ldcs = 64 passes check_dims; the persistent kernels would carve
    H * 64 and G * ntiles * 64 * 64 floats out of a workspace sized for 32 (advisor, round 2)."""
    from paccmann_polymer_amd import _C
    lib = _C.lib
    rng = np.random.default_rng(5)
    V, H, L, W, D = 16, 64, 2, 20, 6
    lens = np.sort(rng.integers(3, 12, size=B))[::-1]
    T = int(lens[0])
    bs = [int((lens > t).sum()) for t in range(T)]
    NT = int(sum(bs))
    g = torch.Generator().manual_seed(1)

    def rnd(*shape):
        return (torch.rand(*shape, generator=g) * 0.4 - 0.2).cuda()
    tokens = torch.tensor(rng.integers(0, V, size=NT), dtype=torch.int64).cuda()
    table = rnd(V, 3 * H)
    w_ih = [rnd(3 * H, H) for _ in range(L)]
    w_hh = [rnd(3 * H, H) for _ in range(L)]
    b_ih = [rnd(3 * H) for _ in range(L)]
    b_hh = [rnd(3 * H) for _ in range(L)]
    wcs, bcs, h0 = rnd(3 + W, H), rnd(3 + W), rnd(L, B, H)
    w0 = rnd(3 * H, W)
    out = {}
    try:
        assert lib.ppk_seq_persist_mode(mode) == mode
        for pad in (32, 64):
            wx0s = torch.zeros(3 * H, pad, device='cuda')
            wx0s[:, :W] = w0
            out[pad] = _seq_fwd_raw(lib, _C, T, bs, B, NT, V, H, L, W, D, tokens, table, wx0s, pad, w_ih, w_hh, b_ih, b_hh, wcs, bcs, h0, pad)
    finally:
        lib.ppk_seq_persist_mode(-1)
    Hs_a, S_a, CS_a, XS_a = out[32]
    Hs_b, S_b, CS_b, XS_b = out[64]
    assert not torch.isnan(Hs_a).any() and not torch.isnan(Hs_b).any()
    # different launch schedules (persistent at the minimal paddings, per-step launches otherwise): fp32 summation order differs
    assert float((Hs_a - Hs_b).abs().max()) <= 2e-5
    assert float((S_a - S_b).abs().max()) <= 2e-5
    assert float((CS_a[:, :3 + W] - CS_b[:, :3 + W]).abs().max()) <= 2e-5
    assert float((XS_a[:, :W] - XS_b[:, :W]).abs().max()) <= 2e-5


def _seq_kernels_launched(H, B, T=6, mode=-1, W=10):
    """Which sequence-kernel families one encoder forward + backward launches (AUTO mode by default; profiling table of the library)."""
    from torch.nn.utils.rnn import pack_sequence
    from paccmann_polymer_amd import _C
    from paccmann_polymer_amd import stack_gru_vae as M
    lib = _C.lib
    params = dict(vocab_size=32, embedding_size=32, rnn_cell_size=H, n_layers=2, stack_width=W, stack_depth=8,
                  latent_dim=16, dropout=0.0, padding_index=0, batch_size=1, batch_mode='packed')
    torch.manual_seed(0)
    enc_net = M.StackGRUEncoder(params).cuda()
    rng = np.random.default_rng(B + H)
    seqs = [torch.tensor(np.concatenate([[2], rng.integers(4, 32, size=int(rng.integers(2, T))), [3]])) for _ in range(B)]
    enc = pack_sequence(seqs, enforce_sorted=False)
    assert lib.ppk_seq_persist_mode(mode) == mode
    lib.ppk_prof_enable(1)
    mu, lv = enc_net.encoder_train_step(enc)
    (mu.sum() + lv.sum()).backward()
    torch.cuda.synchronize()
    lib.ppk_prof_enable(0)
    nk = lib.ppk_prof_kind_count()
    ms, work, launches = (ctypes.c_double * nk)(), (ctypes.c_double * nk)(), (ctypes.c_int64 * nk)()
    lib.ppk_prof_collect(ms, work, launches)
    lib.ppk_seq_persist_mode(-1)
    return {lib.ppk_prof_kind_name(k).decode() for k in range(nk) if launches[k]}


@pytest.mark.parametrize('H,B,fwd_single,bwd_single', [
    (512, 1024, True, True), (512, 2048, True, True), (512, 512, False, False), (256, 2048, False, False), (256, 4096, True, True),
    (384, 1024, True, True), (768, 512, True, True), (768, 2048, False, False), (1024, 512, True, True),
    (128, 2048, True, False),      # H 128 never fills the machine with row-block groups; its forward takes the looped weights-stationary kernel
])
def test_auto_mode_picks_the_measured_faster_schedule(H, B, fwd_single, bwd_single):
    """ppk_seq_persist_mode(-1): the rule of seq_auto_persistent over batch AND hidden size (a group is H / 32 workgroups;
    profiles/r03_persist_shapes.txt has the timings of both schedules at these shapes).  The helper's 10 stack columns have no
    weights-stationary instantiation at H >= 256, so those shapes show the row-block rule alone."""
    kinds = _seq_kernels_launched(H, B)
    assert ('k_seq_fwd' in kinds) == fwd_single, (H, B, sorted(kinds))
    assert ('k_seq_bwd' in kinds) == bwd_single, (H, B, sorted(kinds))


@pytest.mark.parametrize('H,B,mode,fwd_single,bwd_single', [
    (512, 16, -1, True, True), (512, 32, -1, True, True), (512, 33, -1, True, True), (512, 256, -1, True, True), (512, 400, -1, True, False),
    (512, 600, -1, False, False), (512, 16, 0, False, False), (512, 7, 2, True, True), (512, 400, 2, False, False), (512, 700, 3, True, False),
    (256, 21, -1, True, True), (256, 500, -1, True, True), (128, 9, 2, True, True), (96, 9, 2, False, False),
])
def test_small_batches_take_the_weights_stationary_kernels(H, B, mode, fwd_single, bwd_single):
    """Where one launch of replicas (<= 32 sequences each, H / 16 workgroups each, one workgroup per CU) fits the device the
    whole time loop is ONE launch per pass with the weights resident in registers (ppk_seq_ws.hip; reported under the
    k_seq_fwd / k_seq_bwd profiling kinds) in auto mode and mode 2: H 512 up to 256 sequences, H 256 up to 512.  Beyond that the
    forward alone has a looped variant (replicas of up to 256 rows): auto takes it up to 64 rows per replica (H 512: batch 512),
    mode 3 wherever it fits.  In mode 0 and at hidden sizes without an instantiation the per-step launches run."""
    kinds = _seq_kernels_launched(H, B, mode=mode, W=10 if H <= 128 else 40)      # H >= 256 is instantiated for 64 stack columns only
    assert ('k_seq_fwd' in kinds) == fwd_single, (H, B, sorted(kinds))
    assert ('k_seq_bwd' in kinds) == bwd_single, (H, B, sorted(kinds))
    assert ('k_stack_fwd' in kinds) != fwd_single, (H, B, sorted(kinds))
