"""Where a workgroup of the small-batch weights-stationary sequence kernels (ppk_seq_ws.hip) spends its cycles.  Needs a
PPK_WS_TIMING build in its own library:  bash tools/build_alt.sh -DPPK_WS_TIMING && PPK_LIB_PATH=paccmann_polymer_amd/libppk_alt.so python tools/ws_timing.py [B=16]
One encoder forward + backward at the bench architecture; mean / max over the workgroups of the cycles per phase and per timestep."""
import ctypes
import os
import sys

import numpy as np
import torch
from torch.nn.utils.rnn import pack_sequence

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from paccmann_polymer_amd import _C  # noqa: E402
from paccmann_polymer_amd import stack_gru_vae as M  # noqa: E402

fnames = ['W_hh0 h0(t-1)', 'wait heads', 'stack update', 'signal x', 'wait x', 'W_ih0 x', 'epilogue 0', 'signal h0', 'W_hh1 h1(t-1)', 'wait h0', 'W_ih1 h0',
          'epilogue 1', 'heads partial', 'signal h1', '-', '-']
bnames = ['wait stack bwd (t+1)', 'TOP product', 'reduce (sum of both layers)', 'gate backward (both)', 'signal dgi_1', 'prefetch + wait dgi_1', 'P_1 + dXS partial*', 'signal dgi_0',
          'stack backward after its wait', 'signal DCS', 'carry_0 product', 'stack bwd: DMA + wait dgi_0', '-', '-', '-', '-']
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lib = _C.lib
lib.ppk_dbg_ws_timing_buffer.argtypes = [ctypes.c_void_p]
torch.manual_seed(0)
net = M.StackGRUEncoder(bench.PARAMS).cuda()
rng = np.random.default_rng(B)
seqs = [torch.tensor(np.concatenate([[2], rng.integers(4, 64, size=int(rng.integers(18, 99))), [3]])) for _ in range(B)]
enc = pack_sequence(seqs, enforce_sorted=False)
T = enc.batch_sizes.numel()
buf = torch.zeros(2 * 64 * 16, dtype=torch.int64, device='cuda')
assert lib.ppk_seq_persist_mode(2) == 2
for it in range(3):
    buf.zero_()
    lib.ppk_dbg_ws_timing_buffer(ctypes.c_void_p(buf.data_ptr()))
    net.zero_grad(set_to_none=True)
    mu, lv = net.encoder_train_step(enc)
    (mu.sum() + lv.sum()).backward()
    torch.cuda.synchronize()
nwg = bench.PARAMS['rnn_cell_size'] // 16
for name, names, t in (('forward', fnames, buf[:64 * 16]), ('backward', bnames, buf[64 * 16:])):
    t = t.view(64, 16)[:nwg].double().cpu().numpy()
    tot = t.sum(1)
    print(f'{name}: B {B}, T {T}, {nwg} workgroups; cycles per workgroup mean {tot.mean():.3e} max {tot.max():.3e} = {tot.mean() / T:.0f} per timestep')
    for k, n in enumerate(names):
        if n == '-':
            continue
        print(f'  {n:32s} mean {t[:, k].mean() / T:8.0f} per step ({100 * t[:, k].mean() / tot.mean():5.1f} %)   max {t[:, k].max() / T:8.0f}  min {t[:, k].min() / T:8.0f}')
