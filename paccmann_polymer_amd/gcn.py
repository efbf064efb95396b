"""GNN message passing on HIP kernels K6 (CSR build) and K7 (SpMM): `GCNConv`,
`PReLU`-wrapped `GraphEncoder` and `DeepGraphInfomax`.

Host-side mirror of what the reference takes from torch_geometric==1.6.1
(absent offline; semantics SURVEY.md Appendix C, CPU restatement oracle/gcn_ref.py):
  .../chem_reactions/reactions_infomax.py:60      GCNConv(in_channels, hidden_channels, cached=False)
  .../chem_reactions/reactions_infomax.py:64-68   Encoder.forward: encode -> conv -> prelu
  .../chem_reactions/reactions_infomax.py:251-264 DeepGraphInfomax(hidden_channels, encoder, summary, corruption)
  .../chem_reactions/utils.py:22-31               corruption
`edge_index` is [2, E] int64 with flow source -> target (row 0 = source j, row 1 =
target i); only the listed direction propagates, plus self loops.
The aggregation runs BEFORE the dense GEMM when in_channels <= out_channels
((A x) W == A (x W) mathematically; half the gather bytes at 256 -> 512).
"""
import math

import torch
import torch.nn as nn
from torch.nn.utils.rnn import PackedSequence

from . import _C
from .stack_gru_vae import _Linear, _ws, linear

_lib = _C.lib
EPS = 1e-15


class CSRGraph:
    """Forward (by target) and reverse (by source) CSR of the self-loop-augmented,
    symmetrically normalised graph.  Built once per edge_index (cached=False in the
    reference means per call; building is K6, a few launches)."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, normalize: bool = True, check: bool = False):
        _C.require_gpu(edge_index)
        dev = edge_index.device
        ei = edge_index.to(torch.int64).contiguous()
        E, N = int(ei.shape[1]), int(num_nodes)
        self.N, self.E, self.nnz_cap = N, E, E + N
        src, dst = ei[0].contiguous(), ei[1].contiguous()
        self.fwd = self._build(src, dst, E, N, 1, 1, 1 if normalize else 0, None, dev)
        # per-edge weights in augmented edge-id order, then the CSR of the reversed edges
        w_aug = torch.zeros(E + N, 1, dtype=torch.float32, device=dev)
        rowptr, col, eid, val, status = self.fwd
        _C.check(_lib.ppk_scatter_rows(_C.p(val), 1, _C.p(eid), E + N, 1, _C.p(w_aug), 1, E + N, _C.stream()))
        ar = torch.arange(N, dtype=torch.int64, device=dev)
        src_aug, dst_aug = torch.cat([src, ar]), torch.cat([dst, ar])
        self.rev = self._build(dst_aug, src_aug, E + N, N, 0, 0, 0, w_aug.reshape(-1), dev)
        self.status = status
        if check and int(status.item()) != 0:
            raise IndexError('edge_index holds a node id outside [0, num_nodes)')

    @staticmethod
    def _build(src, dst, E, N, drop, add, norm, weight, dev):
        cap = E + (N if add else 0)
        rowptr = torch.empty(N + 1, dtype=torch.int64, device=dev)
        col = torch.zeros(max(cap, 1), dtype=torch.int64, device=dev)
        eid = torch.full((max(cap, 1),), -1, dtype=torch.int64, device=dev)
        val = torch.zeros(max(cap, 1), dtype=torch.float32, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        wsb = _lib.ppk_csr_build_workspace_bytes(N, E)
        ws = _ws(wsb, dev)
        _C.check(_lib.ppk_csr_build(_C.p(src), _C.p(dst), E, N, drop, add, norm, _C.p(weight),
                                    _C.p(rowptr), _C.p(col), _C.p(eid), _C.p(val), _C.p(status), _C.p(ws), wsb, _C.stream()))
        return rowptr, col, eid, val, status


def _spmm(csr, x, cap):
    rowptr, col, eid, val, _ = csr
    x = x.contiguous()
    N, F = rowptr.numel() - 1, x.shape[1]
    out = torch.empty(N, F, dtype=torch.float32, device=x.device)
    wsb = _lib.ppk_spmm_csr_workspace_bytes(N, cap, F)            # hub rows: chunked over the whole grid
    ws = _ws(wsb, x.device)
    _C.check(_lib.ppk_spmm_csr_ws(_C.p(rowptr), _C.p(col), _C.p(val), N, cap, _C.p(x), x.stride(0), F, _C.p(out), F,
                                  _C.p(ws), wsb, _C.stream()))
    return out


class _Propagate(torch.autograd.Function):
    """out = A_hat x  (K7); backward = A_hat^T dout on the reversed CSR."""

    @staticmethod
    def forward(ctx, x, graph):
        _C.require_gpu(x)
        ctx.graph = graph
        return _spmm(graph.fwd, x, graph.nnz_cap)

    @staticmethod
    def backward(ctx, dout):
        g = ctx.graph
        return _spmm(g.rev, dout, g.nnz_cap), None


def propagate(x, graph: CSRGraph):
    return _Propagate.apply(x, graph)


class _MatmulBias(torch.autograd.Function):
    """y = x @ W + b with W stored [in, out] as GCNConv keeps it.  Over millions of rows the NN kernel with 128 x 128
    tiles is the fastest form of each product (tools/bench_gcn_gemm.py, 2.4 M x 256 x 512: forward 112 TFLOP/s against
    106 through the transposed copy and the 64 x 64 NT tiles; input gradient 122 against 113), so the backward takes a
    transposed copy of the 0.5 MB weight instead of re-reading 5 GB through the slower form; weight and bias gradient
    come from one pass over dy."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _C.require_gpu(x, weight, bias)
        x, weight = x.contiguous(), weight.contiguous()
        M, K = x.shape
        N = weight.shape[1]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        _C.check(_lib.ppk_gemm_f32(0, 0, M, N, K, 1.0, _C.p(x), K, _C.p(weight), N, 0.0, _C.p(y), N, _C.p(bias), None, 0, _C.stream()))
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        M, K = x.shape
        N = weight.shape[1]
        dev = x.device
        wt = weight.t().contiguous()                                   # [out, in]
        dx = torch.empty_like(x)
        _C.check(_lib.ppk_gemm_f32(0, 0, M, K, N, 1.0, _C.p(dy), N, _C.p(wt), K, 0.0, _C.p(dx), K, None, None, 0, _C.stream()))
        dwt = torch.empty(N, K, dtype=torch.float32, device=dev)       # dy^T . x = dW^T
        wsb = _lib.ppk_gemm_f32_workspace_bytes(N, K, M)
        ws = _ws(wsb, dev)
        db = torch.empty(N, dtype=torch.float32, device=dev) if ctx.has_bias else None
        rc = _C.PPK_ERR_UNSUP
        if ctx.has_bias:
            rc = _lib.ppk_gemm_tn_colsum_f32(N, K, M, _C.p(dy), N, _C.p(x), K, _C.p(dwt), K, _C.p(db), _C.p(ws), wsb, _C.stream())
            if rc != _C.PPK_ERR_UNSUP:
                _C.check(rc)
        if rc == _C.PPK_ERR_UNSUP:
            _C.check(_lib.ppk_gemm_f32(1, 0, N, K, M, 1.0, _C.p(dy), N, _C.p(x), K, 0.0, _C.p(dwt), K, None, _C.p(ws), wsb, _C.stream()))
            if ctx.has_bias:
                cb = _lib.ppk_colsum_workspace_bytes(M, N)
                cws = _ws(cb, dev)
                _C.check(_lib.ppk_colsum(_C.p(dy), N, M, N, 0.0, _C.p(db), _C.p(cws), cb, _C.stream()))
        return dx, dwt.t(), db


class GCNConv(nn.Module):
    """torch_geometric.nn.GCNConv(in_channels, out_channels, cached=False) subset:
    unit edge weights, add_self_loops=True, normalize=True, bias=True."""

    def __init__(self, in_channels, out_channels, cached=False, bias=True):
        super().__init__()
        self.in_channels, self.out_channels, self.cached = in_channels, out_channels, cached
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self._cache = None
        self.reset_parameters()

    def reset_parameters(self):
        a = math.sqrt(6.0 / (self.in_channels + self.out_channels))     # glorot
        nn.init.uniform_(self.weight, -a, a)
        if self.bias is not None:
            nn.init.zeros_(self.bias)
        self._cache = None

    def forward(self, x, edge_index):
        N = x.shape[0]
        if isinstance(edge_index, CSRGraph):
            graph = edge_index
        elif self.cached and self._cache is not None:
            graph = self._cache
        else:
            graph = CSRGraph(edge_index.to(x.device), N)
            if self.cached:
                self._cache = graph
        if self.in_channels <= self.out_channels:
            return _MatmulBias.apply(propagate(x, graph), self.weight, self.bias)
        out = propagate(_MatmulBias.apply(x, self.weight, None), graph)
        return out + self.bias if self.bias is not None else out


class _PReLU(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, alpha):
        _C.require_gpu(x, alpha)
        x = x.contiguous()
        rows, C = x.shape
        y = torch.empty_like(x)
        _C.check(_lib.ppk_prelu_fwd(_C.p(x), _C.p(alpha), rows, C, _C.p(y), _C.stream()))
        ctx.save_for_backward(x, alpha)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, alpha = ctx.saved_tensors
        dy = dy.contiguous()
        rows, C = x.shape
        dx, t = torch.empty_like(x), torch.empty_like(x)
        _C.check(_lib.ppk_prelu_bwd(_C.p(x), _C.p(alpha), _C.p(dy), rows, C, _C.p(dx), _C.p(t), _C.stream()))
        dalpha = torch.empty(C, dtype=torch.float32, device=x.device)
        cb = _lib.ppk_colsum_workspace_bytes(rows, C)
        cws = _ws(cb, x.device)
        _C.check(_lib.ppk_colsum(_C.p(t), C, rows, C, 0.0, _C.p(dalpha), _C.p(cws), cb, _C.stream()))
        return dx, dalpha


class PReLU(nn.Module):
    """nn.PReLU(num_parameters) on the HIP kernel (same `weight` parameter name / init 0.25)."""

    def __init__(self, num_parameters=1, init=0.25):
        super().__init__()
        self.num_parameters = num_parameters
        self.weight = nn.Parameter(torch.full((num_parameters,), float(init)))

    def forward(self, x):
        alpha = self.weight if self.num_parameters == x.shape[1] else self.weight.expand(x.shape[1]).contiguous()
        return _PReLU.apply(x, alpha)


class _ColMean(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        rows, C = x.shape
        out = torch.empty(C, dtype=torch.float32, device=x.device)
        cb = _lib.ppk_colsum_workspace_bytes(rows, C)
        cws = _ws(cb, x.device)
        _C.check(_lib.ppk_colsum(_C.p(x), C, rows, C, 0.0, _C.p(out), _C.p(cws), cb, _C.stream()))
        ctx.rows = rows
        return out / rows

    @staticmethod
    def backward(ctx, d):
        return (d / ctx.rows).unsqueeze(0).expand(ctx.rows, -1)


class _DGILoss(torch.autograd.Function):

    @staticmethod
    def forward(ctx, p, n):
        p, n = p.contiguous().reshape(-1), n.contiguous().reshape(-1)
        N = p.numel()
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        wsb = _lib.ppk_dgi_loss_workspace_bytes(N)
        ws = _ws(wsb, p.device)
        _C.check(_lib.ppk_dgi_loss_fwd(_C.p(p), _C.p(n), N, _C.p(loss), _C.p(ws), wsb, _C.stream()))
        ctx.save_for_backward(p, n)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        p, n = ctx.saved_tensors
        N = p.numel()
        dl = dloss.to(torch.float32).reshape(1).contiguous()
        dp, dn = torch.empty_like(p), torch.empty_like(n)
        _C.check(_lib.ppk_dgi_loss_bwd(_C.p(p), _C.p(n), N, _C.p(dl), _C.p(dp), _C.p(dn), _C.stream()))
        return dp.reshape(-1, 1), dn.reshape(-1, 1)


def default_summary(z, *args, **kwargs):
    """reactions_infomax.py:254: sigmoid(z.mean(dim=0))."""
    return torch.sigmoid(_ColMean.apply(z))


def corruption(x, edge_index):
    """utils.py:22-31.  Tensor features: rows shuffled with `torch.randperm(N)` drawn from the CPU generator, as
    the reference draws it.  PackedSequence (the branch reactions_infomax.py:131 hits): the reference pads, permutes
    the samples with the same `randperm` draw, and re-packs with `enforce_sorted=False`; the permuted batch holds the
    same multiset of lengths, so the result is the same token stream gathered by index — computed here without the
    [B, T] padded round trip, bit-identical in `data`, `batch_sizes`, `sorted_indices` and `unsorted_indices`
    (tests/test_corruption.py runs the reference statements beside it)."""
    if not isinstance(x, PackedSequence):
        return x[torch.randperm(x.size(0)).to(x.device)], edge_index
    bs = x.batch_sizes.to('cpu', torch.int64)
    T, B = int(bs.numel()), int(bs[0])
    dev = x.data.device
    off = torch.zeros(T + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(bs, 0)
    len_sorted = (bs.view(T, 1) > torch.arange(B, dtype=torch.int64).view(1, B)).sum(0)        # lengths in packed (sorted) order
    unsorted = x.unsorted_indices.cpu() if x.unsorted_indices is not None else torch.arange(B, dtype=torch.int64)
    lengths = len_sorted[unsorted]                                                             # ORIGINAL sample order (pad_packed_sequence)
    idxs = torch.randperm(B)                                                                   # same draw as utils.py:27
    new_len = lengths[idxs]
    sorted_len, sorted_idx = torch.sort(new_len, descending=True)                              # pack_padded_sequence(enforce_sorted=False)
    new_unsorted = torch.empty_like(sorted_idx)
    new_unsorted[sorted_idx] = torch.arange(B, dtype=torch.int64)
    # packed slot (t, i) of the corrupted batch holds step t of original sample idxs[sorted_idx[i]],
    # which sits at packed slot (t, unsorted[that sample]) of the input
    src_col = unsorted[idxs[sorted_idx]]
    valid = torch.arange(T, dtype=torch.int64).view(T, 1) < sorted_len.view(1, B)
    t_idx, i_idx = valid.nonzero(as_tuple=True)
    src = (off[t_idx] + src_col[i_idx]).to(dev)
    return PackedSequence(x.data[src], bs, sorted_idx.to(dev), new_unsorted.to(dev)), edge_index


class DeepGraphInfomax(nn.Module):
    """torch_geometric.nn.DeepGraphInfomax subset used by the reference."""

    def __init__(self, hidden_channels, encoder, summary=default_summary, corruption=corruption):
        super().__init__()
        self.hidden_channels = hidden_channels
        self.encoder = encoder
        self.summary = summary
        self.corruption = corruption
        self.weight = nn.Parameter(torch.empty(hidden_channels, hidden_channels))
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(self.hidden_channels)
        nn.init.uniform_(self.weight, -bound, bound)

    def forward(self, *args, **kwargs):
        pos_z = self.encoder(*args, **kwargs)
        cor = self.corruption(*args, **kwargs)
        cor = cor if isinstance(cor, tuple) else (cor,)
        neg_z = self.encoder(*cor)
        summary = self.summary(pos_z, *args, **kwargs)
        return pos_z, neg_z, summary

    def _logits(self, z, summary):
        u = linear(summary.reshape(1, -1), self.weight, None)        # (W s)^T  [1, h]
        return linear(z, u, None)                                     # z . (W s) [N, 1]

    def discriminate(self, z, summary, sigmoid=True):
        value = self._logits(z, summary).reshape(-1)
        return torch.sigmoid(value) if sigmoid else value

    def loss(self, pos_z, neg_z, summary):
        return _DGILoss.apply(self._logits(pos_z, summary), self._logits(neg_z, summary))

    def test(self, train_z, train_y, test_z, test_y, solver='lbfgs', multi_class='auto', *args, **kwargs):
        """torch_geometric's `DeepGraphInfomax.test` (the Planetoid baselines call it,
        experiments/cora/infomax.py): accuracy of a logistic regression fitted on the frozen latents.  Host-side
        scikit-learn exactly as upstream does it; `multi_class` is forwarded only where the installed
        scikit-learn still takes it (removed from LogisticRegression in 1.7; 'auto' was its default)."""
        import inspect
        from sklearn.linear_model import LogisticRegression
        if 'multi_class' in inspect.signature(LogisticRegression.__init__).parameters and multi_class != 'auto':
            kwargs['multi_class'] = multi_class
        clf = LogisticRegression(solver=solver, *args, **kwargs).fit(train_z.detach().cpu().numpy(), train_y.detach().cpu().numpy())
        return clf.score(test_z.detach().cpu().numpy(), test_y.detach().cpu().numpy())


class GraphEncoder(nn.Module):
    """reactions_infomax.py:38-83 with the frozen sequence encoder factored out:
    `features -> GCNConv -> PReLU` (the reference feeds `paccmann_vae.encode(x)`,
    computed under no_grad, as the features)."""

    def __init__(self, in_channels=256, hidden_channels=512, feature_fn=None):
        super().__init__()
        self.feature_fn = feature_fn
        self.conv = GCNConv(in_channels, hidden_channels, cached=False)
        self.prelu = PReLU(hidden_channels)

    def forward(self, x, edge_index):
        if self.feature_fn is not None:
            with torch.no_grad():
                x = self.feature_fn(x)
        return self.prelu(self.conv(x, edge_index))
