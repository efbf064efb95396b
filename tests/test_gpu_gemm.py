"""ppk_gemm_f32 (fp32 MFMA) against torch fp64 matmul; tolerance 2e-6 rel
(fp32 fmaf-chain accumulation)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [
    # M, N, K
    (128, 128, 32), (64, 64, 8), (1, 1, 1), (37, 53, 178), (2048, 1536, 512), (300, 1536, 178),
    (2048, 53, 512), (77, 690, 1536), (1536, 512, 5000), (53, 512, 3001), (130, 200, 4),
    # long K, N > 192: the TN layout takes the LDS-DMA kernel k_tn128 (split-K slabs, odd K tail, partial tiles)
    (384, 256, 9001), (200, 196, 8200), (1536, 512, 20011),
]


@pytest.mark.parametrize('M,N,K', SHAPES)
@pytest.mark.parametrize('tA,tB', [(0, 1), (0, 0), (1, 0), (1, 1)])
def test_gemm_layouts(M, N, K, tA, tB):
    from paccmann_polymer_amd import ops
    torch.manual_seed(M * 7 + N * 3 + K)
    lda = ((M if tA else K) + 3) // 4 * 4 + 4
    ldb = ((K if tB else N) + 3) // 4 * 4
    Afull = torch.randn(K if tA else M, lda, device='cuda')
    Bfull = torch.randn(N if tB else K, ldb, device='cuda')
    A = Afull[:, :(M if tA else K)]
    B = Bfull[:, :(K if tB else N)]
    bias = torch.randn(N, device='cuda')
    C0 = torch.randn(M, N, device='cuda')
    ws = torch.empty(ops._C.lib.ppk_gemm_f32_workspace_bytes(M, N, K), dtype=torch.uint8, device='cuda')
    C = C0.clone()
    ops.gemm(A, B, bool(tA), bool(tB), alpha=0.5, beta=2.0, C=C, bias=bias, ws=ws)
    opA = (A.T if tA else A).double()
    opB = (B.T if tB else B).double()
    ref = 0.5 * (opA @ opB) + 2.0 * C0.double() + bias.double()
    scale = (opA.abs() @ opB.abs()).max().item() + 1.0
    assert (C.double() - ref).abs().max().item() <= 2e-6 * scale
    # no bias / beta = 0 must not read C (NaN-poisoned)
    C2 = torch.full((M, N), float('nan'), device='cuda')
    ops.gemm(A, B, bool(tA), bool(tB), C=C2, beta=0.0, ws=ws)
    assert (C2.double() - opA @ opB).abs().max().item() <= 2e-6 * scale


def test_gemm_asymmetric_identity():
    # A = I with an asymmetric B catches a transposed C write
    from paccmann_polymer_amd import ops
    n = 96
    A = torch.eye(n, device='cuda')
    B = torch.arange(n * n, device='cuda', dtype=torch.float32).reshape(n, n)
    assert torch.equal(ops.gemm(A, B), B)
    assert torch.equal(ops.gemm(A, B, transB=True), B.T)


def test_gemm_reproducible_splitk():
    from paccmann_polymer_amd import ops
    torch.manual_seed(0)
    A = torch.randn(20000, 256, device='cuda')
    B = torch.randn(20000, 128, device='cuda')
    ws = torch.empty(ops._C.lib.ppk_gemm_f32_workspace_bytes(256, 128, 20000), dtype=torch.uint8, device='cuda')
    c1 = ops.gemm(A, B, transA=True, ws=ws)
    c2 = ops.gemm(A, B, transA=True, ws=ws)
    assert torch.equal(c1, c2)
    c3 = ops.gemm(A, B, transA=True, ws=False)                 # no workspace: no split-K, same product
    assert torch.allclose(c3, c1, rtol=1e-4, atol=1e-2)
    assert torch.equal(ops.gemm(A, B, transA=True), c1)        # ws=None allocates the workspace itself
    ref = A.double().T @ B.double()
    assert (c1.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


def test_gemm_unaligned_leading_dims():
    # lda = 178 / ldb = 53 (not multiples of 4) take the scalar-load path
    from paccmann_polymer_amd import ops
    torch.manual_seed(3)
    A = torch.randn(100, 178, device='cuda')
    W = torch.randn(1536, 178, device='cuda')
    ref = A.double() @ W.double().T
    assert (ops.gemm(A, W, transB=True).double() - ref).abs().max().item() <= 2e-6 * 178 * 16
    B = torch.randn(178, 53, device='cuda')
    ref = A.double() @ B.double()
    assert (ops.gemm(A, B).double() - ref).abs().max().item() <= 2e-6 * 178 * 16


def test_gemm_errors():
    from paccmann_polymer_amd import ops
    A = torch.zeros(8, 6, device='cuda')
    B = torch.zeros(7, 8, device='cuda')
    with pytest.raises(ValueError):
        ops.gemm(A, B)


@pytest.mark.parametrize('M,N,K', [(384, 256, 9001), (1536, 512, 20011), (200, 196, 8200)])
def test_gemm_tn_colsum_fused(M, N, K):
    """C = A^T B and the column sums of A in one pass (weight gradient + bias gradient)."""
    from paccmann_polymer_amd import _C
    torch.manual_seed(M + N + K)
    A = torch.randn(K, M, device='cuda')
    B = torch.randn(K, N, device='cuda')
    C = torch.full((M, N), float('nan'), device='cuda')
    cs = torch.full((M,), float('nan'), device='cuda')
    wsb = _C.lib.ppk_gemm_f32_workspace_bytes(M, N, K)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device='cuda')
    for _ in range(2):
        _C.check(_C.lib.ppk_gemm_tn_colsum_f32(M, N, K, _C.p(A), M, _C.p(B), N, _C.p(C), N, _C.p(cs), _C.p(ws), wsb, _C.stream()))
    ref = A.double().T @ B.double()
    assert (C.double() - ref).abs().max().item() <= 2e-6 * ((A.abs().double().T @ B.abs().double()).max().item() + 1.0)
    ref_cs = A.double().sum(0)
    assert (cs.double() - ref_cs).abs().max().item() <= 2e-6 * A.abs().double().sum(0).max().item()
    C2, cs2 = C.clone(), cs.clone()
    _C.check(_C.lib.ppk_gemm_tn_colsum_f32(M, N, K, _C.p(A), M, _C.p(B), N, _C.p(C), N, _C.p(cs), _C.p(ws), wsb, _C.stream()))
    assert torch.equal(C, C2) and torch.equal(cs, cs2)          # fixed-order sums


def test_gemm_tn_colsum_unsupported_shape_launches_nothing():
    from paccmann_polymer_amd import _C
    A = torch.randn(100, 64, device='cuda')
    B = torch.randn(100, 32, device='cuda')
    C = torch.full((64, 32), 7.0, device='cuda')
    cs = torch.full((64,), 7.0, device='cuda')
    rc = _C.lib.ppk_gemm_tn_colsum_f32(64, 32, 100, _C.p(A), 64, _C.p(B), 32, _C.p(C), 32, _C.p(cs), None, 0, _C.stream())
    assert rc == _C.PPK_ERR_UNSUP
    torch.cuda.synchronize()
    assert bool((C == 7.0).all()) and bool((cs == 7.0).all())


@pytest.mark.parametrize('n_rows,n_keys,F', [(5000, 575, 96), (124072, 64, 1536), (3000, 3000, 64), (777, 8192, 8), (1, 1, 4)])
def test_rowsum_by_key_vs_fp64_index_add(n_rows, n_keys, F):
    """ppk_rowsum_by_key (the embedding-table gradient of K1): rows summed per key in a fixed order — against fp64
    index_add_, bitwise reproducible, skewed key frequencies (a hub key holding a fifth of the rows), empty keys,
    an out-of-range key reported through `status` and skipped."""
    from paccmann_polymer_amd import _C
    lib = _C.lib
    g = torch.Generator().manual_seed(n_rows + n_keys)
    p = 1.0 / torch.arange(1, n_keys + 1, dtype=torch.float64)
    keys = torch.multinomial(p / p.sum(), n_rows, replacement=True, generator=g).cuda()
    x = torch.randn(n_rows, F, generator=g).cuda()
    wsb = lib.ppk_rowsum_by_key_workspace_bytes(n_rows, n_keys, F)
    ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device='cuda')

    def run(k):
        out = torch.full((n_keys, F), float('nan'), device='cuda')
        status = torch.zeros(1, dtype=torch.int32, device='cuda')
        _C.check(lib.ppk_rowsum_by_key(_C.p(k), n_rows, n_keys, _C.p(x), F, F, _C.p(out), F, _C.p(status), _C.p(ws), wsb, _C.stream()))
        return out, int(status)
    a, st = run(keys)
    b, _ = run(keys)
    assert st == 0 and torch.equal(a, b)
    ref = torch.zeros(n_keys, F, dtype=torch.float64, device='cuda').index_add_(0, keys, x.double())
    scale = torch.zeros(n_keys, F, dtype=torch.float64, device='cuda').index_add_(0, keys, x.double().abs())
    assert float(((a.double() - ref).abs() / (scale + 1.0)).max()) <= 2e-6
    if n_rows > 10:
        bad = keys.clone()
        bad[3] = n_keys
        bad[7] = -1
        c, st = run(bad)
        ok = torch.ones(n_rows, dtype=torch.bool, device='cuda')
        ok[3] = ok[7] = False
        ref2 = torch.zeros(n_keys, F, dtype=torch.float64, device='cuda').index_add_(0, keys[ok], x[ok].double())
        assert st != 0 and float(((c.double() - ref2).abs() / (scale + 1.0)).max()) <= 2e-6
