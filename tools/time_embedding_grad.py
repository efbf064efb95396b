"""Embedding-table gradient of K1 at a real vocabulary (V = 575, the pretrained PaccMann SMILES language) and bench batch
(2048 sequences, NT ~ 124 k packed rows, 3H = 1536): the round-2 dense one-hot product [V x NT] . [NT x 3H] against the
round-3 sort-free segmented reduction (ppk_rowsum_by_key).  Prints ms per call
and checks the two agree; the segmented form must be bitwise reproducible."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paccmann_polymer_amd import _C  # noqa: E402
from paccmann_polymer_amd import stack_gru_vae as M  # noqa: E402

lib = _C.lib
V = int(sys.argv[1]) if len(sys.argv) > 1 else 575
NT, H3 = 124072, 1536
g = torch.Generator('cuda').manual_seed(0)
# Zipf-like token frequencies (a few very frequent tokens: 'C', 'c', '(', ')', ...)
p = 1.0 / torch.arange(1, V + 1, dtype=torch.float64)
tokens = torch.multinomial(p / p.sum(), NT, replacement=True).cuda()
dgi = torch.randn(NT, H3, device='cuda', generator=g)


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


def onehot_gemm():
    onehot = torch.zeros(NT, V, device='cuda')
    onehot.scatter_(1, tokens.view(-1, 1), 1.0)
    out = torch.empty(V, H3, device='cuda')
    wsb = lib.ppk_gemm_f32_workspace_bytes(V, H3, NT)
    ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device='cuda')
    _C.check(lib.ppk_gemm_f32(1, 0, V, H3, NT, 1.0, _C.p(onehot), V, _C.p(dgi), H3, 0.0, _C.p(out), H3, None, _C.p(ws), wsb, _C.stream()))
    return out


def segmented():
    return M._rowsum_by_token(tokens, V, dgi)


t_gemm, a = timed(onehot_gemm)
t_seg, b = timed(segmented)
b2 = segmented()
ref = torch.zeros(V, H3, dtype=torch.float64, device='cuda').index_add_(0, tokens, dgi.double())
print(f'V = {V}, NT = {NT}, 3H = {H3}: largest token row count {int(torch.bincount(tokens).max())}')
print(f'  one-hot GEMM ({2.0 * NT * V * H3 / 1e9:.0f} GFLOP + {NT * V * 4 / 1e6:.0f} MB of one-hot): {t_gemm:.3f} ms')
print(f'  per-token row sums, ppk_rowsum_by_key (counting sort + {NT * H3 * 4 / 1e6:.0f} MB read): {t_seg:.3f} ms = {NT * H3 * 4 / t_seg / 1e9:.2f} TB/s')
print(f'  rel. error vs fp64 index_add: one-hot {float((a.double() - ref).norm() / ref.norm()):.2e}, segmented {float((b.double() - ref).norm() / ref.norm()):.2e}; '
      f'segmented bitwise reproducible: {bool(torch.equal(b, b2))}')
