"""Where a workgroup of the persistent sequence kernel spends its cycles (needs a PPK_SEQ_TIMING build of
ppk_seq_persist.hip into its own library: see the command in the header of the output).  One forward pass of the
encoder at bench shape; prints the mean / max over workgroups of the cycles per phase."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from paccmann_polymer_amd import _C  # noqa: E402
from paccmann_polymer_amd.paccmann_encoder import Encoder  # noqa: E402

names = ['kloop A', 'wait input', 'kloop B', 'heads partial', 'signal', 'wait top', 'stack: XS store', 'signal S',
         'stack: loads', 'stack: sync', 'stack: act', 'stack: update', 'stack: sync2', 'epilogue', '-', '-']
lib = _C.lib
lib.ppk_dbg_seq_timing_buffer.argtypes = [ctypes.c_void_p]
lib.ppk_dbg_seq_bwd_timing_buffer.argtypes = [ctypes.c_void_p]
bnames = ['TOP (wait SB + K = ldcs)', 'gate backward', 'signal + wait E_l', 'P_l K loops', 'dXS partial', 'signal + wait E_0', 'stack backward',
          'signal SB + P_0 + carry'] + ['-'] * 8
rng = np.random.default_rng(1234)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
seqs, gs, sizes = bench.synth_batch(rng, B // 16, 16)
enc, dec, tgt, g_flat = bench.to_device_batch(seqs, gs, 'cuda')
torch.manual_seed(0)
model = Encoder(bench.PARAMS, None, None, batch_size=1, batch_mode='Packed')
G = ((B + 31) // 32 + 1) // 2
nwg = G * 16
buf = torch.zeros(nwg * 16, dtype=torch.int64, device='cuda')
bbuf = torch.zeros(nwg * 16, dtype=torch.int64, device='cuda')
model.gru_vae.train()
for it in range(3):
    buf.zero_()
    bbuf.zero_()
    lib.ppk_dbg_seq_timing_buffer(ctypes.c_void_p(buf.data_ptr()))
    lib.ppk_dbg_seq_bwd_timing_buffer(ctypes.c_void_p(bbuf.data_ptr()))
    torch.cuda.synchronize()
    ev0, ev1, ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev0.record()
    mu, lv = model.gru_vae.encode(enc)
    ev1.record()
    (mu.sum() + lv.sum()).backward()
    ev2.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1)
    msb = ev1.elapsed_time(ev2)
t = buf.view(nwg, 16).double().cpu().numpy()
tot = t.sum(1)
print(f'encode forward {ms:.3f} ms, T = {enc.batch_sizes.numel()}, workgroups {nwg}; cycles per workgroup: mean {tot.mean():.3e} max {tot.max():.3e}')
for k, n in enumerate(names):
    print(f'  {n:12s} mean {t[:, k].mean():12.0f} ({100 * t[:, k].mean() / tot.mean():5.1f} %)   max {t[:, k].max():12.0f}  min {t[:, k].min():12.0f}')
print('  per group (mean over its workgroups), total / K loops / waits, Mcycles:')
for g in range(0, G, max(1, G // 8)):
    r = t[g * 16:(g + 1) * 16]
    print(f'    group {g:3d}: {r.sum(1).mean() / 1e6:7.2f} / {(r[:, 0] + r[:, 2]).mean() / 1e6:6.2f} / {(r[:, 1] + r[:, 5]).mean() / 1e6:6.2f}')
t = bbuf.view(nwg, 16).double().cpu().numpy()
tot = t.sum(1)
print(f'encode backward (whole autograd pass incl. weight gradients) {msb:.3f} ms; k_seq_bwd cycles per workgroup: mean {tot.mean():.3e} max {tot.max():.3e}')
for k, n in enumerate(bnames):
    if n != '-':
        print(f'  {n:26s} mean {t[:, k].mean():12.0f} ({100 * t[:, k].mean() / tot.mean():5.1f} %)   max {t[:, k].max():12.0f}  min {t[:, k].min():12.0f}')
print('  per group (mean over its workgroups), total / K loops / waits, Mcycles:')
for g in range(0, G, max(1, G // 8)):
    r = t[g * 16:(g + 1) * 16]
    print(f'    group {g:3d}: {r.sum(1).mean() / 1e6:7.2f} / {(r[:, 3] + r[:, 7]).mean() / 1e6:6.2f} / {(r[:, 2] + r[:, 5]).mean() / 1e6:6.2f}')
