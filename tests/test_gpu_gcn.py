"""GPU parity of K6 (CSR build: bit-exact integers) and K7 (SpMM) / GCNConv /
PReLU / DeepGraphInfomax against oracle/gcn_ref.py.  Parity versus
torch_geometric itself is UNPINNED (package absent offline)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def random_graph(rng, n, e, hubs=0, self_loops=True):
    src = rng.integers(0, n, size=e)
    dst = rng.integers(0, n, size=e)
    if hubs:
        dst[: e // 3] = rng.integers(0, hubs, size=e // 3)      # a few very long rows
    if not self_loops:
        dst = np.where(src == dst, (dst + 1) % n, dst)
    return torch.tensor(np.stack([src, dst]), dtype=torch.int64)


@pytest.mark.parametrize('n,e,hubs', [(9, 8, 0), (50, 400, 0), (1000, 30000, 2), (5000, 200000, 1), (7, 0, 0), (3000, 2999, 0),
                                      (3000, 40000, 1), (900, 100000, 2), (50, 120000, 1)])     # rows beyond one LDS sort image: chunk sort + merge passes
def test_csr_build_bit_exact(n, e, hubs):
    from oracle import gcn_ref as R
    from paccmann_polymer_amd.gcn import CSRGraph
    rng = np.random.default_rng(n + e)
    ei = random_graph(rng, n, e, hubs)
    g = CSRGraph(ei.cuda(), n, check=True)
    rowptr, col, eid, val, _ = g.fwd
    rp, c, ed, nv = R.csr_ref(ei, n)
    nnz = int(rp[-1])
    assert np.array_equal(rowptr.cpu().numpy(), rp)
    assert np.array_equal(col.cpu().numpy()[:nnz], c)
    assert np.array_equal(eid.cpu().numpy()[:nnz], ed)
    assert np.allclose(val.cpu().numpy()[:nnz], nv, rtol=1e-6, atol=0)
    # reversed CSR: same multiset of (src, dst, weight), grouped by source, edge order kept
    rrow, rcol, reid, rval, _ = g.rev
    assert int(rrow[-1]) == e + n
    rr = rrow.cpu().numpy()
    src_aug = np.concatenate([ei[0].numpy(), np.arange(n)])
    for i in rng.integers(0, n, size=min(n, 20)):
        ids = reid.cpu().numpy()[rr[i]:rr[i + 1]]
        assert np.all(src_aug[ids] == i) and np.all(np.diff(ids) > 0)


def test_reference_example_edge_index():
    # SURVEY §0.4 example; one direction per undirected edge
    from oracle import gcn_ref as R
    from paccmann_polymer_amd.gcn import GCNConv
    ei = torch.tensor([[0, 0, 1, 1, 1, 1, 2, 2], [1, 2, 3, 5, 7, 8, 4, 6]])
    torch.manual_seed(0)
    conv = GCNConv(16, 32).cuda()
    x = torch.randn(9, 16)
    out = conv(x.cuda(), ei.cuda())
    ref = R.gcn_conv_ref(x.double(), ei, conv.weight.detach().cpu().double(), conv.bias.detach().cpu().double())
    assert rel(out, ref) <= 1e-6


@pytest.mark.parametrize('n,e,fin,fout', [(200, 1500, 256, 512), (64, 300, 30, 7), (4000, 50000, 64, 64), (500, 9000, 30, 7),
                                            (300, 20000, 320, 16)])      # hub rows: k_spmm_long, scalar and vector
def test_gcnconv_fwd_bwd(n, e, fin, fout):
    from oracle import gcn_ref as R
    from paccmann_polymer_amd.gcn import GCNConv
    rng = np.random.default_rng(e)
    ei = random_graph(rng, n, e, hubs=1)
    torch.manual_seed(1)
    conv = GCNConv(fin, fout).cuda()
    with torch.no_grad():
        conv.bias.uniform_(-0.1, 0.1)
    x = torch.randn(n, fin)
    xg = x.cuda().requires_grad_(True)
    out = conv(xg, ei.cuda())
    wgt = torch.randn(n, fout)
    (out * wgt.cuda()).sum().backward()
    xr = x.double().requires_grad_(True)
    wr = conv.weight.detach().cpu().double().requires_grad_(True)
    br = conv.bias.detach().cpu().double().requires_grad_(True)
    ref = R.gcn_conv_ref(xr, ei, wr, br)
    (ref * wgt.double()).sum().backward()
    assert rel(out, ref) <= 1e-5
    assert rel(xg.grad, xr.grad) <= 1e-4
    assert rel(conv.weight.grad, wr.grad) <= 1e-4
    assert rel(conv.bias.grad, br.grad) <= 1e-4
    # deterministic: bitwise equal on a second run
    out2 = conv(xg, ei.cuda())
    assert torch.equal(out, out2)


def test_dgi_pipeline():
    from oracle import gcn_ref as R
    from paccmann_polymer_amd.gcn import DeepGraphInfomax, GraphEncoder
    rng = np.random.default_rng(5)
    n, fin, hid = 21, 24, 40
    ei = random_graph(rng, n, 20, self_loops=False)
    torch.manual_seed(2)
    perm = torch.randperm(n)
    enc = GraphEncoder(fin, hid).cuda()
    model = DeepGraphInfomax(hid, enc, corruption=lambda x, e: (x[perm.to(x.device)], e)).cuda()
    x = torch.randn(n, fin)
    pos, neg, s = model(x.cuda(), ei.cuda())
    loss = model.loss(pos, neg, s)
    loss.backward()
    # reference in fp64
    W = enc.conv.weight.detach().cpu().double().requires_grad_(True)
    b = enc.conv.bias.detach().cpu().double().requires_grad_(True)
    a = enc.prelu.weight.detach().cpu().double().requires_grad_(True)
    Wd = model.weight.detach().cpu().double().requires_grad_(True)

    def f(xx):
        return torch.nn.functional.prelu(R.gcn_conv_ref(xx, ei, W, b), a)
    pz, nz = f(x.double()), f(x.double()[perm])
    sr = torch.sigmoid(pz.mean(0))
    lr = R.dgi_loss_ref(pz, nz, sr, Wd)
    lr.backward()
    assert abs(float(loss) - float(lr)) <= 1e-5 * abs(float(lr))
    assert rel(enc.conv.weight.grad, W.grad) <= 1e-4
    assert rel(enc.conv.bias.grad, b.grad) <= 1e-4
    assert rel(enc.prelu.weight.grad, a.grad) <= 1e-4
    assert rel(model.weight.grad, Wd.grad) <= 1e-4


def test_bad_edge_index_reported():
    from paccmann_polymer_amd.gcn import CSRGraph
    ei = torch.tensor([[0, 1], [1, 7]]).cuda()
    with pytest.raises(IndexError):
        CSRGraph(ei, 3, check=True)


def test_full_size_csr_and_spmm_vs_torch_on_device():
    """K6 + K7 at BASELINE configs[2] size (2.4 M nodes, 5 M directed edges, F = 256).  The CPU oracle would
    take minutes here, so the same statements (stable sort of the augmented edge list by target; fp64
    index_add in edge order) are evaluated with torch on the device: integers bit-exact, sums to 1e-6."""
    from paccmann_polymer_amd import gcn
    rng = np.random.default_rng(9)
    N, E, F = 2_400_000, 5_000_000, 256
    ei = random_graph(rng, N, E, hubs=3).cuda()
    g = gcn.CSRGraph(ei, N, check=True)
    rowptr, col, eid, val, _ = g.fwd
    keep = ei[0] != ei[1]
    loops = torch.arange(N, device='cuda')
    src = torch.cat([ei[0][keep], loops])
    dst = torch.cat([ei[1][keep], loops])
    ids = torch.cat([torch.arange(E, device='cuda')[keep], E + loops])
    nnz = int(src.numel())
    assert int(rowptr[-1]) == nnz
    deg = torch.bincount(dst, minlength=N)
    assert torch.equal(rowptr[1:].long(), torch.cumsum(deg, 0))
    _, order = torch.sort(dst, stable=True)
    assert torch.equal(col[:nnz].long(), src[order])
    assert torch.equal(eid[:nnz].long(), ids[order])
    dinv = deg.double().pow(-0.5)
    norm = dinv[src] * dinv[dst]
    assert torch.allclose(val[:nnz].double(), norm[order], rtol=1e-6, atol=0)
    x = torch.randn(N, F, device='cuda')
    y = gcn._spmm(g.fwd, x, g.nnz_cap)
    ref = torch.zeros(N, F, dtype=torch.float64, device='cuda').index_add_(0, dst, norm[:, None] * x.double()[src])
    assert rel(y, ref) <= 1e-6
    del ref
    yt = gcn._spmm(g.rev, x, g.nnz_cap)
    ref = torch.zeros(N, F, dtype=torch.float64, device='cuda').index_add_(0, src, norm[:, None] * x.double()[dst])
    assert rel(yt, ref) <= 1e-6
    assert torch.equal(y, gcn._spmm(g.fwd, x, g.nnz_cap))


@pytest.mark.parametrize('F', [256, 30])
def test_hub_rows_chunked_entry_point_vs_single_workgroup_entry_point(F):
    """ppk_spmm_csr_ws (long rows cut into 1024-entry chunks over the whole grid, partials added in chunk order)
    against ppk_spmm_csr (one workgroup per long row) through the C-ABI: short rows bit-identical, long rows to
    fp32 summation accuracy against an fp64 index_add; twice in a row bit-identical (the listing order of the
    long rows varies from run to run, the sums must not)."""
    from paccmann_polymer_amd import _C, gcn
    lib = _C.lib
    rng = np.random.default_rng(21)
    N, E = 20000, 400000
    ei = random_graph(rng, N, E, hubs=3).cuda()
    g = gcn.CSRGraph(ei, N)
    rowptr, col, eid, val, _ = g.fwd
    x = torch.randn(N, F, device='cuda')
    old = torch.empty(N, F, device='cuda')
    _C.check(lib.ppk_spmm_csr(_C.p(rowptr), _C.p(col), _C.p(val), N, g.nnz_cap, _C.p(x), F, F, _C.p(old), F, _C.stream()))
    wsb = lib.ppk_spmm_csr_workspace_bytes(N, g.nnz_cap, F)
    ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
    outs = []
    for _ in range(2):
        new = torch.full((N, F), float('nan'), device='cuda')
        _C.check(lib.ppk_spmm_csr_ws(_C.p(rowptr), _C.p(col), _C.p(val), N, g.nnz_cap, _C.p(x), F, F, _C.p(new), F, _C.p(ws), wsb, _C.stream()))
        outs.append(new)
    assert torch.equal(outs[0], outs[1])
    deg = (rowptr[1:] - rowptr[:-1]).long()
    short = deg <= 1024
    assert int((~short).sum()) == 3
    assert torch.equal(outs[0][short], old[short])
    nnz = int(rowptr[-1])
    rows = torch.repeat_interleave(torch.arange(N, device='cuda'), deg)
    ref = torch.zeros(N, F, dtype=torch.float64, device='cuda').index_add_(0, rows, val[:nnz].double()[:, None] * x.double()[col[:nnz].long()])
    assert rel(outs[0], ref) <= 1e-6 and rel(old, ref) <= 1e-6
    assert lib.ppk_spmm_csr_ws(_C.p(rowptr), _C.p(col), _C.p(val), N, g.nnz_cap, _C.p(x), F, F, _C.p(new), F, _C.p(ws), 16, _C.stream()) == -3
