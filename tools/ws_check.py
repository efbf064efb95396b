"""Small-batch weights-stationary sequence kernels (ppk_seq_ws.hip) against the per-step launches: which kernel families run
(profiling table of the library), largest difference of the outputs / gradients, time per encoder pass.
Usage: python tools/ws_check.py [B=16] [T=100]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch
from torch.nn.utils.rnn import pack_sequence

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from paccmann_polymer_amd import _C  # noqa: E402
from paccmann_polymer_amd import stack_gru_vae as M  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lib = _C.lib
torch.manual_seed(0)
net = M.StackGRUEncoder(bench.PARAMS).cuda()
rng = np.random.default_rng(B)
seqs = [torch.tensor(np.concatenate([[2], rng.integers(4, 64, size=int(rng.integers(18, T - 1))), [3]])) for _ in range(B)]
enc = pack_sequence(seqs, enforce_sorted=False)


def run(mode, reps=10):
    assert lib.ppk_seq_persist_mode(mode) == mode
    outs = None
    for r in range(2 + reps):
        if r == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        net.zero_grad(set_to_none=True)
        mu, lv = net.encoder_train_step(enc)
        (mu.sum() + (lv * lv).sum()).backward()
        if r == 0:
            outs = [mu.detach().clone(), lv.detach().clone()] + [p.grad.detach().clone() for p in net.parameters()]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    lib.ppk_prof_enable(1)
    mu, lv = net.encoder_train_step(enc)
    torch.cuda.synchronize()
    lib.ppk_prof_enable(0)
    nk = lib.ppk_prof_kind_count()
    ms, work, launches = (ctypes.c_double * nk)(), (ctypes.c_double * nk)(), (ctypes.c_int64 * nk)()
    lib.ppk_prof_collect(ms, work, launches)
    kinds = {lib.ppk_prof_kind_name(k).decode(): (launches[k], round(ms[k], 3)) for k in range(nk) if launches[k]}
    return outs, dt, kinds


base, dt0, k0 = run(0)
for mode in (1, 2, 3):
    outs, dt, kinds = run(mode)
    worst = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(outs, base))
    print(f'mode {mode}: fwd+bwd {dt:.3f} ms (per-step launches {dt0:.3f} ms), worst rel-to-max difference vs mode 0 {worst:.2e}')
    print('   forward kernels:', kinds)
print('mode 0 forward kernels:', k0)
lib.ppk_seq_persist_mode(-1)
