"""The allocation-free entry point `ppk_colsum_noalloc` (include/ppk.h): called through the C-ABI and checked
bit for bit against `ppk_colsum` (the existing entry point whose workspace it replaces by the library's own
file-scope device array) on two calls in a row, the second with a smaller input; sizes beyond the built-in
scratch are refused with nothing launched.  Collected last on purpose (the round-end run uses `pytest -x`)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def colsum_ws(lib, _C, A, rows, cols, beta, out):
    wsb = lib.ppk_colsum_workspace_bytes(rows, cols)
    ws = torch.empty(max(int(wsb), 256), dtype=torch.uint8, device='cuda')
    _C.check(lib.ppk_colsum(_C.p(A), A.stride(0), rows, cols, beta, _C.p(out), _C.p(ws), wsb, _C.stream()))


def test_noalloc_colsum_matches_workspace_colsum_twice():
    from paccmann_polymer_amd import _C
    lib = _C.lib
    torch.manual_seed(0)
    # (rows, cols, lda, beta): first the larger problem, then a smaller one that reuses the front of the scratch
    for rows, cols, lda, beta in [(70000, 1536, 1536, 0.0), (1300, 53, 64, 0.5), (1, 1, 4, 0.0)]:
        A = torch.randn(rows, lda, device='cuda')
        base = torch.randn(cols, device='cuda')
        want, got = base.clone(), base.clone()
        colsum_ws(lib, _C, A, rows, cols, beta, want)
        before = torch.cuda.memory_allocated()
        _C.check(lib.ppk_colsum_noalloc(_C.p(A), lda, rows, cols, beta, _C.p(got), _C.stream()))
        assert torch.cuda.memory_allocated() == before
        torch.cuda.synchronize()
        assert torch.equal(got, want), (rows, cols)
        ref = A[:, :cols].double().sum(0) + beta * base.double()
        assert float((got.double() - ref).norm() / ref.norm()) <= 1e-5


def test_noalloc_colsum_rejects_sizes_beyond_the_scratch():
    from paccmann_polymer_amd import _C
    lib = _C.lib
    rows, cols = 512 * 200, 1536          # 200 partial rows x 1536 columns = 307200 floats > 262144
    A = torch.zeros(8, cols, device='cuda')            # never read: the call must refuse before launching
    out = torch.full((cols,), 7.0, device='cuda')
    rc = lib.ppk_colsum_noalloc(_C.p(A), cols, rows, cols, 0.0, _C.p(out), _C.stream())
    assert rc == _C.PPK_ERR_UNSUP
    assert b'scratch' in lib.ppk_last_error()
    torch.cuda.synchronize()
    assert bool((out == 7.0).all())
