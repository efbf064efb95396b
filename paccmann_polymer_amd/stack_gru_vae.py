"""Stack-augmented GRU VAE (encoder / decoder / TeacherVAE) on HIP kernels.

Host-side mirror of the slice of `paccmann_chemistry@0.0.4` that the
reference's hot path uses (the package itself is absent offline; semantics per
SURVEY.md Appendix B, CPU restatement in oracle/stackgru_ref.py):

  reference call sites
    paccmann_polymer/models/paccmann_encoder.py:52-59   StackGRUEncoder(params), StackGRUDecoder(params), TeacherVAE(enc, dec)
    paccmann_polymer/models/paccmann_encoder.py:87-88   ._update_batch_size(int)
    paccmann_polymer/models/paccmann_encoder.py:166-168 .encode / .reparameterize / .decode
    .../chem_reactions/reactions_vae.py:177             .encoder.set_batch_mode('packed')
    .../chem_reactions/reactions_vae.py:209             .save(path)

The nn.Module objects are PARAMETER CONTAINERS with upstream-compatible
`state_dict` keys (`encoder.gru.weight_ih_l0`, `decoder.output_layer.weight`, …);
their torch `forward`s are never used.  All arithmetic runs through libppk.so:
K1 sequence kernels + fp32-MFMA GEMMs, K2 cross-entropy, K3 reparameterise+KL,
each with a hand-written backward wired in through `torch.autograd.Function`.

Restrictions of the HIP path (raise instead of silently falling back):
`use_stack=True`, `bidirectional=False`, packed batches (`PackedSequence`), fp32.
Training-mode dropout (`nn.GRU(dropout=p)` between the layers) runs inside the sequence kernels.
"""
import ctypes
import os
from typing import Optional

import torch
import torch.nn as nn
from torch.nn.utils.rnn import PackedSequence

from . import _C
from .data_prep import known_token_range, note_token_range

_lib = _C.lib


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


# --------------------------------------------------------------------------
# small autograd ops on the C-ABI
# --------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    """y = x @ W^T + b  (fp32 MFMA GEMM; backward: NN / TN GEMMs + column sum)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _C.require_gpu(x, weight, bias)
        x = x.contiguous()
        M, K = x.shape
        N = weight.shape[0]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        _C.check(_lib.ppk_gemm_f32(0, 1, M, N, K, 1.0, _C.p(x), x.stride(0), _C.p(weight), weight.stride(0), 0.0,
                                   _C.p(y), N, _C.p(bias), None, 0, _C.stream()))
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        M, K = x.shape
        N = weight.shape[0]
        dev = x.device
        dx = torch.empty_like(x)
        _C.check(_lib.ppk_gemm_f32(0, 0, M, K, N, 1.0, _C.p(dy), N, _C.p(weight), weight.stride(0), 0.0,
                                   _C.p(dx), K, None, None, 0, _C.stream()))
        dw = torch.empty(N, K, dtype=torch.float32, device=dev)
        wsb = _lib.ppk_gemm_f32_workspace_bytes(N, K, M)
        ws = _ws(wsb, dev)
        db = torch.empty(N, dtype=torch.float32, device=dev) if ctx.has_bias else None
        rc = _C.PPK_ERR_UNSUP
        if ctx.has_bias:                # dw = dy^T . x and db = column sums of dy from one pass over dy, when the shape allows
            rc = _lib.ppk_gemm_tn_colsum_f32(N, K, M, _C.p(dy), N, _C.p(x), K, _C.p(dw), K, _C.p(db), _C.p(ws), wsb, _C.stream())
            if rc != _C.PPK_ERR_UNSUP:
                _C.check(rc)
        if rc == _C.PPK_ERR_UNSUP:
            _C.check(_lib.ppk_gemm_f32(1, 0, N, K, M, 1.0, _C.p(dy), N, _C.p(x), K, 0.0,
                                       _C.p(dw), K, None, _C.p(ws), wsb, _C.stream()))
            if ctx.has_bias:
                cb = _lib.ppk_colsum_workspace_bytes(M, N)
                cws = _ws(cb, dev)
                _C.check(_lib.ppk_colsum(_C.p(dy), N, M, N, 0.0, _C.p(db), _C.p(cws), cb, _C.stream()))
        return dx, dw, db


class _GatherRows(torch.autograd.Function):
    """out[i] = src[idx[i]] with DISTINCT idx (backward = scatter into zeros)."""

    @staticmethod
    def forward(ctx, src, idx):
        _C.require_gpu(src, idx)
        src = src.contiguous()
        n, w = idx.numel(), src.shape[1]
        out = torch.empty(n, w, dtype=torch.float32, device=src.device)
        _C.check(_lib.ppk_gather_rows(_C.p(src), src.stride(0), src.shape[0], _C.p(idx), n, w, _C.p(out), w, _C.stream()))
        ctx.save_for_backward(idx)
        ctx.src_rows = src.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dout = dout.contiguous()
        n, w = dout.shape
        dsrc = torch.zeros(ctx.src_rows, w, dtype=torch.float32, device=dout.device)
        _C.check(_lib.ppk_scatter_rows(_C.p(dout), w, _C.p(idx), n, w, _C.p(dsrc), w, ctx.src_rows, _C.stream()))
        return dsrc, None


class _CrossEntropy(torch.autograd.Function):
    """K2: mean CE over rows with target != ignore_index."""

    @staticmethod
    def forward(ctx, logits, targets, ignore_index):
        _C.require_gpu(logits, targets)
        logits = logits.contiguous()
        rows, V = logits.shape
        wsb = _lib.ppk_ce_workspace_bytes(rows)
        ws = _ws(wsb, logits.device)
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        _C.check(_lib.ppk_ce_fwd(_C.p(logits), V, _C.p(targets), rows, V, ignore_index, _C.p(loss), _C.p(ws), wsb, _C.stream()))
        ctx.save_for_backward(logits, targets, ws)
        ctx.ignore_index = ignore_index
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        logits, targets, ws = ctx.saved_tensors
        rows, V = logits.shape
        dl = dloss.to(torch.float32).reshape(1).contiguous()
        dlogits = torch.empty_like(logits)
        _C.check(_lib.ppk_ce_bwd(_C.p(logits), V, _C.p(targets), rows, V, ctx.ignore_index, _C.p(dl), _C.p(dlogits), V,
                                 _C.p(ws), ws.numel(), _C.stream()))
        return dlogits, None, None


class _ReparamKL(torch.autograd.Function):
    """K3: z = eps * exp(0.5 logvar) + mu and KL = -0.5 sum(1 + logvar - mu^2 - exp(logvar))."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        _C.require_gpu(mu, logvar, eps)
        mu, logvar, eps = mu.contiguous(), logvar.contiguous(), eps.contiguous()
        n = mu.numel()
        z = torch.empty_like(mu)
        kl = torch.empty(1, dtype=torch.float32, device=mu.device)
        wsb = _lib.ppk_reparam_workspace_bytes(n)
        ws = _ws(wsb, mu.device)
        _C.check(_lib.ppk_reparam_fwd(_C.p(mu), _C.p(logvar), _C.p(eps), n, _C.p(z), _C.p(kl), _C.p(ws), wsb, _C.stream()))
        ctx.save_for_backward(mu, logvar, eps)
        return z, kl.reshape(())

    @staticmethod
    def backward(ctx, dz, dkl):
        mu, logvar, eps = ctx.saved_tensors
        n = mu.numel()
        dz = dz.contiguous() if dz is not None else None
        dk = dkl.to(torch.float32).reshape(1).contiguous() if dkl is not None else None
        dmu, dlv = torch.empty_like(mu), torch.empty_like(logvar)
        _C.check(_lib.ppk_reparam_bwd(_C.p(mu), _C.p(logvar), _C.p(eps), n, _C.p(dz), _C.p(dk), _C.p(dmu), _C.p(dlv), _C.stream()))
        return dmu, dlv, None


def reparameterize_kl(mu, logvar, eps):
    """(z, KL) — fused K3 op (cora/vae.py:31-34,50)."""
    return _ReparamKL.apply(mu, logvar, eps)


def linear(x, weight, bias=None):
    return _Linear.apply(x, weight, bias)


class HIPLinear(nn.Linear):
    """`nn.Linear` whose call runs the fp32-MFMA GEMM of libppk.so (same parameters, same `state_dict` keys).  The
    reference's in-tree callers apply these layers directly — `decoder.latent_to_hidden(latent_z)`,
    `decoder.output_layer(output)` (experiments/latent_space_exploration/latent_space_map_fns.py:288,306) — on
    tensors with leading batch dimensions."""

    def forward(self, x):
        lead = x.shape[:-1]
        y = linear(x.reshape(-1, x.shape[-1]).to(torch.float32), self.weight, self.bias)
        return y.reshape(*lead, self.out_features)


def _round_up(x, m):
    return (x + m - 1) // m * m


def _pad_units(t, H, Hp, dim, gates=1):
    """Zero-pad `gates` consecutive blocks of H hidden units along `dim` to Hp units each."""
    if Hp == H:
        t = t.contiguous()
        return t if t.data_ptr() % 16 == 0 else t.clone()      # kernels want 16-byte aligned operands
    shape = list(t.shape)
    shape[dim:dim + 1] = [gates, H]
    t = t.reshape(shape)
    pad_shape = list(shape)
    pad_shape[dim + 1] = Hp - H
    t = torch.cat([t, t.new_zeros(pad_shape)], dim=dim + 1)
    shape[dim:dim + 2] = [gates * Hp]
    return t.reshape(shape).contiguous()


def _unpad_units(t, H, Hp, dim, gates=1):
    if Hp == H:
        return t
    shape = list(t.shape)
    shape[dim:dim + 1] = [gates, Hp]
    t = t.reshape(shape).narrow(dim + 1, 0, H)
    shape[dim:dim + 2] = [gates * H]
    return t.reshape(shape).contiguous()


def _gemm(ta, tb, M, N, K, A, lda, Bm, ldb, C, ldc, bias=None, ws=None, wsb=0):
    _C.check(_lib.ppk_gemm_f32(ta, tb, M, N, K, 1.0, _C.p(A), lda, _C.p(Bm), ldb, 0.0, _C.p(C), ldc,
                               _C.p(bias), _C.p(ws), wsb, _C.stream()))


def _rowsum_by_token(tokens, V, rows):
    """[V, F] = the rows of `rows` [NT, F] summed per token id, in a fixed order (ppk_rowsum_by_key: sort-free stable
    counting sort + chunked partial sums; bitwise reproducible).  Replaces the dense one-hot product
    [V x NT] . [NT x 3H] of round 2 (2 NT V 3H FLOP: 219 GFLOP at the pretrained language's V = 575).  Vocabularies
    beyond the kernel's 8192 LDS bins go through K6 + K7 (ppk_csr_build_rect, ppk_spmm_csr_ws): same fixed-order sums."""
    dev = rows.device
    NT, F = rows.shape
    out = torch.empty(V, F, dtype=torch.float32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    if V <= 8192:
        wsb = _lib.ppk_rowsum_by_key_workspace_bytes(NT, V, F)
        ws = _ws(wsb, dev)
        _C.check(_lib.ppk_rowsum_by_key(_C.p(tokens), NT, V, _C.p(rows), rows.stride(0), F, _C.p(out), F, _C.p(status), _C.p(ws), wsb, _C.stream()))
        return out
    idx = torch.arange(NT, dtype=torch.int64, device=dev)
    rowptr = torch.empty(V + 1, dtype=torch.int64, device=dev)
    col = torch.zeros(NT, dtype=torch.int64, device=dev)
    eid = torch.empty(NT, dtype=torch.int64, device=dev)
    val = torch.zeros(NT, dtype=torch.float32, device=dev)
    wsb = _lib.ppk_csr_build_workspace_bytes(V, NT)
    ws = _ws(wsb, dev)
    _C.check(_lib.ppk_csr_build_rect(_C.p(idx), _C.p(tokens), NT, V, NT, 0, 0, 0, None, _C.p(rowptr), _C.p(col), _C.p(eid), _C.p(val),
                                     _C.p(status), _C.p(ws), wsb, _C.stream()))
    sws_b = _lib.ppk_spmm_csr_workspace_bytes(V, NT, F)
    sws = _ws(sws_b, dev)
    _C.check(_lib.ppk_spmm_csr_ws(_C.p(rowptr), _C.p(col), _C.p(val), V, NT, _C.p(rows), rows.stride(0), F, _C.p(out), F, _C.p(sws), sws_b, _C.stream()))
    return out


def _seq_forward(tokens, batch_sizes, dims, h0, S0, emb, wc, bc, wsi, bsi, gru, drop=None):
    """Runs ppk_stackgru_seq_fwd; returns every buffer the backward (or the single-step API) needs.
    drop: [L-1, NT, Hp] inter-layer dropout factors (0 or 1 / (1 - p)), None outside training-mode dropout."""
    V, E, H, L, W, D, pad = dims
    _C.require_gpu(tokens, h0, emb, wc, bc, wsi, bsi, *gru)
    dev = emb.device
    T = batch_sizes.numel()
    B = int(batch_sizes[0])
    NT = int(tokens.numel())
    Hp = _round_up(H, 32)
    ks = _round_up(W, 32)
    ldcs = _round_up(3 + W, 32)
    f32 = dict(dtype=torch.float32, device=dev)
    w_ih = [gru[4 * l + 0] for l in range(L)]
    w_hh = [gru[4 * l + 1] for l in range(L)]
    b_ih = [gru[4 * l + 2] for l in range(L)]
    b_hh = [gru[4 * l + 3] for l in range(L)]

    # [3+W, Hp] stack heads (controls rows 0..2, then the stack input layer)
    wcs = _pad_units(torch.cat([wc, wsi], dim=0), H, Hp, 1)
    bcs = torch.cat([bc, bsi], dim=0).contiguous()
    w_hh_p = [_pad_units(_pad_units(w, H, Hp, 0, 3), H, Hp, 1) for w in w_hh]
    b_hh_p = [_pad_units(b, H, Hp, 0, 3) for b in b_hh]
    w_ih_p = [None] + [_pad_units(_pad_units(w, H, Hp, 0, 3), H, Hp, 1) for w in w_ih[1:]]
    b_ih_p = [None] + [_pad_units(b, H, Hp, 0, 3) for b in b_ih[1:]]
    w0 = _pad_units(w_ih[0], H, Hp, 0, 3)                       # [3Hp, E + W]
    b0 = _pad_units(b_ih[0], H, Hp, 0, 3)
    w0e = w0[:, :E]                                             # view, row stride E + W
    wx0s = torch.zeros(3 * Hp, ks, **f32)
    wx0s[:, :W] = w0[:, E:E + W]
    embc = emb.contiguous()
    table = torch.empty(V, 3 * Hp, **f32)
    _gemm(0, 1, V, 3 * Hp, E, embc, E, w0e, E + W, table, 3 * Hp, bias=b0)
    h0p = _pad_units(h0, H, Hp, 2)
    if S0 is not None:
        S0 = S0.to(torch.float32).contiguous()
        if tuple(S0.shape) != (B, D, W):
            raise ValueError()

    XS = torch.empty(NT, ks, **f32)
    Hs = torch.empty(L, NT, Hp, **f32)
    HP = torch.empty(L, NT, Hp, **f32)
    gates = torch.empty(L, NT, 4 * Hp, **f32)
    S = torch.empty(NT, D, W, **f32)
    CS = torch.empty(NT, ldcs, **f32)
    HD = None
    if drop is not None and L > 1:
        if tuple(drop.shape) != (L - 1, NT, Hp) or drop.dtype != torch.float32 or not drop.is_contiguous():
            raise ValueError()
        HD = torch.empty(L - 1, NT, Hp, **f32)
    else:
        drop = None
    wsb = _lib.ppk_stackgru_seq_fwd_workspace_bytes(B, Hp, L, W)
    ws = _ws(wsb, dev)
    bs_host = batch_sizes.contiguous()
    dummy = w_hh_p[0]
    _C.check(_lib.ppk_stackgru_seq_fwd(
        T, ctypes.c_void_p(bs_host.data_ptr()), B, NT, V, Hp, L, W, D,
        _C.p(tokens), _C.p(table), _C.p(wx0s), ks,
        _ptr_array([dummy] + w_ih_p[1:]), _ptr_array(w_hh_p),
        _ptr_array([dummy] + b_ih_p[1:]), _ptr_array(b_hh_p), _C.p(wcs), _C.p(bcs), _C.p(h0p), _C.p(S0),
        _C.p(XS), _C.p(Hs), _C.p(HP), _C.p(gates), _C.p(S), _C.p(CS), ldcs, _C.p(drop), _C.p(HD),
        _C.p(ws), wsb, _C.stream()))
    return dict(XS=XS, Hs=Hs, HP=HP, gates=gates, S=S, CS=CS, HD=HD, drop=drop, wcs=wcs, wx0s=wx0s, w_ih_p=w_ih_p, w_hh_p=w_hh_p,
                embc=embc, w0=w0, bs_host=bs_host, dims=(V, E, H, Hp, L, W, D, T, B, NT, ks, ldcs), pad=pad)


class _StackGRUSeq(torch.autograd.Function):
    """K1: the whole packed StackGRU sequence.  Returns the top layer's output of
    every step, [NT, H] in packed order.

    Host-side preparation per call (all on the GPU, outside the time loop):
      * hidden units are zero-padded to a multiple of 32 (a padded unit has zero
        weights and zero initial state, so it stays exactly 0 and receives zero gradient);
      * the embedding half of the layer-0 input product becomes a [V, 3H] table,
        table[v] = emb[v] . W_ih0[:, :E]^T + b_ih0, looked up per token inside the kernel;
      * the backward takes transposed weight copies so its GEMMs read k-contiguous operands.
    """

    @staticmethod
    def forward(ctx, tokens, batch_sizes, dims, drop, h0, emb, wc, bc, wsi, bsi, *gru):
        r = _seq_forward(tokens, batch_sizes, dims, h0, None, emb, wc, bc, wsi, bsi, gru, drop)
        V, E, H, Hp, L, W, D, T, B, NT, ks, ldcs = r['dims']
        f32 = dict(dtype=torch.float32, device=emb.device)
        # transposed copies for the backward's data-gradient GEMMs (k-contiguous operands)
        wcs_t = torch.zeros(Hp, ldcs, **f32)
        wcs_t[:, :3 + W] = r['wcs'].t()
        w_ih_t = [r['wx0s'][:, :W].t().contiguous()] + [w.t().contiguous() for w in r['w_ih_p'][1:]]
        w_hh_t = [w.t().contiguous() for w in r['w_hh_p']]
        ctx.save_for_backward(tokens, r['XS'], r['Hs'], r['HP'], r['gates'], r['S'], r['CS'], wcs_t, r['embc'], r['w0'],
                              *w_ih_t, *w_hh_t)
        ctx.bs_host = r['bs_host']
        ctx.drop, ctx.HD = r['drop'], r['HD']                 # inter-layer dropout: factors and the dropped-out inputs of layers >= 1
        ctx.dims = r['dims']
        ctx.padding_idx = -1 if r['pad'] is None else int(r['pad'])
        return _unpad_units(r['Hs'][L - 1], H, Hp, 1)

    @staticmethod
    def backward(ctx, dHtop):
        V, E, H, Hp, L, W, D, T, B, NT, ks, ldcs = ctx.dims
        saved = ctx.saved_tensors
        tokens, XS, Hs, HP, gates, S, CS, wcs_t, embc, w0 = saved[:10]
        w_ih_t = list(saved[10:10 + L])
        w_hh_t = list(saved[10 + L:10 + 2 * L])
        dev = XS.device
        f32 = dict(dtype=torch.float32, device=dev)
        dHtop = _pad_units(dHtop, H, Hp, 1)
        dXS = torch.empty(NT, ks, **f32)
        DGI = torch.empty(L, NT, 3 * Hp, **f32)
        DGH = torch.empty(L, NT, 3 * Hp, **f32)
        DCS = torch.empty(NT, ldcs, **f32)
        dh0 = torch.empty(L, B, Hp, **f32)
        wsb = _lib.ppk_stackgru_seq_bwd_workspace_bytes(B, Hp, L, W, D)
        ws = _ws(wsb, dev)
        st = _C.stream()
        _C.check(_lib.ppk_stackgru_seq_bwd(
            T, ctypes.c_void_p(ctx.bs_host.data_ptr()), B, NT, Hp, L, W, D,
            _ptr_array(w_ih_t), _ptr_array(w_hh_t), _C.p(wcs_t),
            _C.p(HP), _C.p(gates), _C.p(S), _C.p(CS), ldcs,
            _C.p(dHtop), _C.p(dXS), ks, _C.p(DGI), _C.p(DGH), _C.p(DCS), _C.p(dh0), _C.p(ctx.drop),
            _C.p(ws), wsb, st))

        # ---- weight gradients: single large TN GEMMs over the saved activations
        gws_b = _lib.ppk_gemm_f32_workspace_bytes(3 * Hp, max(Hp, ks), NT)
        gws = _ws(gws_b, dev)
        cws_b = _lib.ppk_colsum_workspace_bytes(NT, 3 * Hp)
        cws = _ws(cws_b, dev)

        def tn(A, lda, M, Bm, ldb, N, K=NT):
            out = torch.empty(M, N, **f32)
            _gemm(1, 0, M, N, K, A, lda, Bm, ldb, out, N, ws=gws, wsb=gws_b)
            return out

        def colsum(A, lda, C):
            out = torch.empty(C, **f32)
            _C.check(_lib.ppk_colsum(_C.p(A), lda, NT, C, 0.0, _C.p(out), _C.p(cws), cws_b, st))
            return out

        def tn_colsum(A, lda, M, Bm, ldb, N):
            """(A^T . B, column sums of A): one pass over A when the fused kernel takes the shape."""
            if os.environ.get('PPK_FUSED_COLSUM', '1') == '0':          # tuning aid
                return tn(A, lda, M, Bm, ldb, N), colsum(A, lda, M)
            out = torch.empty(M, N, **f32)
            cs = torch.empty(M, **f32)
            rc = _lib.ppk_gemm_tn_colsum_f32(M, N, NT, _C.p(A), lda, _C.p(Bm), ldb, _C.p(out), N, _C.p(cs), _C.p(gws), gws_b, st)
            if rc == _C.PPK_ERR_UNSUP:
                return tn(A, lda, M, Bm, ldb, N), colsum(A, lda, M)
            _C.check(rc)
            return out, cs

        def w_grad(g):          # [3Hp, Hp] -> [3H, H]
            return _unpad_units(_unpad_units(g, H, Hp, 0, 3), H, Hp, 1)

        # layer 0: stack columns from XS; embedding columns through the per-token table:
        # dtable[v] = sum of the rows of DGI0 whose token is v, in ascending row order (K6 + K7, bitwise reproducible)
        dtable = _rowsum_by_token(tokens, V, DGI[0])                          # [V, 3Hp]
        dw0s = tn(DGI[0], 3 * Hp, 3 * Hp, XS, ks, W)                          # [3Hp, W]
        dw0e = tn(dtable, 3 * Hp, 3 * Hp, embc, E, E, K=V)                    # [3Hp, E] = dtable^T . emb
        demb = torch.empty(V, E, **f32)
        _gemm(0, 0, V, E, 3 * Hp, dtable, 3 * Hp, w0, E + W, demb, E)         # dtable . W_ih0[:, :E]
        if ctx.padding_idx >= 0:
            demb[ctx.padding_idx].zero_()
        dwhh0, dbhh0 = tn_colsum(DGH[0], 3 * Hp, 3 * Hp, HP[0], Hp, Hp)
        gru_grads = [
            _unpad_units(torch.cat([dw0e, dw0s], dim=1), H, Hp, 0, 3),
            w_grad(dwhh0),
            _unpad_units(dtable.sum(0), H, Hp, 0, 3),            # sum_rows DGI0 = sum over tokens of the per-token sums
            _unpad_units(dbhh0, H, Hp, 0, 3),
        ]
        for l in range(1, L):
            below = ctx.HD[l - 1] if ctx.HD is not None else Hs[l - 1]      # what layer l read: h_{l-1} (* its dropout factors)
            dwih, dbih = tn_colsum(DGI[l], 3 * Hp, 3 * Hp, below, Hp, Hp)
            dwhh, dbhh = tn_colsum(DGH[l], 3 * Hp, 3 * Hp, HP[l], Hp, Hp)
            gru_grads += [w_grad(dwih), w_grad(dwhh), _unpad_units(dbih, H, Hp, 0, 3), _unpad_units(dbhh, H, Hp, 0, 3)]
        dwcs = _unpad_units(tn(DCS, ldcs, 3 + W, HP[L - 1], Hp, Hp), H, Hp, 1)
        dbcs = colsum(DCS, ldcs, 3 + W)
        dh0 = _unpad_units(dh0, H, Hp, 2)
        return (None, None, None, None, dh0, demb, dwcs[:3], dbcs[:3], dwcs[3:], dbcs[3:], *gru_grads)


# --------------------------------------------------------------------------
# modules (parameter containers + HIP forward)
# --------------------------------------------------------------------------
class StackGRU(nn.Module):

    def __init__(self, params):
        super().__init__()
        self.params = dict(params)
        p = self.params
        self.vocab_size = p['vocab_size']
        self.embedding_size = p['embedding_size']
        self.rnn_cell_size = p['rnn_cell_size']
        self.n_layers = p['n_layers']
        self.stack_width = p['stack_width']
        self.stack_depth = p['stack_depth']
        self.dropout = p.get('dropout', 0.0)
        self.padding_index = p.get('padding_index', 0)
        self.batch_size = p.get('batch_size', 1)
        self.batch_mode = str(p.get('batch_mode', 'packed')).lower()
        self.use_stack = p.get('use_stack', True)
        self.bidirectional = p.get('bidirectional', False)
        if not self.use_stack or self.bidirectional:
            raise NotImplementedError('the HIP path implements use_stack=True, bidirectional=False')
        self.embedding = nn.Embedding(self.vocab_size, self.embedding_size, padding_idx=self.padding_index)
        self.gru = nn.GRU(self.embedding_size + self.stack_width, self.rnn_cell_size, self.n_layers,
                          dropout=self.dropout)
        self.stack_controls_layer = HIPLinear(self.rnn_cell_size, 3)
        self.stack_input_layer = HIPLinear(self.rnn_cell_size, self.stack_width)

    # upstream API kept for drop-in callers (reactions_vae.py:62-63,177)
    def _update_batch_size(self, batch_size: int):
        self.batch_size = int(batch_size)

    def set_batch_mode(self, mode: str):
        mode = str(mode).lower()
        if mode not in ('packed', 'padded'):
            raise NotImplementedError(f'Unknown batch mode: {mode}')
        self.batch_mode = mode

    def _dims(self):
        return (self.vocab_size, self.embedding_size, self.rnn_cell_size, self.n_layers,
                self.stack_width, self.stack_depth, self.padding_index)

    def _gru_params(self):
        out = []
        for l in range(self.n_layers):
            out += [getattr(self.gru, f'weight_ih_l{l}'), getattr(self.gru, f'weight_hh_l{l}'),
                    getattr(self.gru, f'bias_ih_l{l}'), getattr(self.gru, f'bias_hh_l{l}')]
        return out

    def _run_sequence(self, packed: PackedSequence, h0):
        """-> top-layer outputs [NT, H] in packed order."""
        if not isinstance(packed, PackedSequence):
            raise TypeError('the HIP path takes PackedSequence batches (batch_mode="packed")')
        dev = self.embedding.weight.device
        tokens = packed.data
        # ids index a [V, 3H] table inside the kernels: validate like nn.Embedding does (IndexError), on the host copy
        # of the range when data_prep recorded one, otherwise once per token tensor (one sync, then remembered)
        rng = known_token_range(tokens)
        if rng is None and tokens.numel():
            lo, hi = torch.aminmax(tokens)
            rng = (int(lo), int(hi))
            note_token_range(tokens, *rng)
        if rng is not None and (rng[0] < 0 or rng[1] >= self.vocab_size):
            raise IndexError('token index out of range')
        if tokens.dim() != 1:
            tokens = tokens.reshape(-1)
        tokens = tokens.to(device=dev, dtype=torch.int64).contiguous()
        bs = packed.batch_sizes.to('cpu', torch.int64)
        drop = self._dropout_factors(int(tokens.numel()), dev)
        return _StackGRUSeq.apply(tokens, bs, self._dims(), drop, h0, self.embedding.weight,
                                  self.stack_controls_layer.weight, self.stack_controls_layer.bias,
                                  self.stack_input_layer.weight, self.stack_input_layer.bias,
                                  *self._gru_params())


    # Training-mode dropout of nn.GRU(dropout = p): between the layers, fresh factors at every timestep (upstream calls
    # nn.GRU with seq_len 1 per step).  `drop_factors_fn(n_masked_layers, NT, H) -> [n, NT, H]` injects the factors in
    # packed-row order (parity tests: the CPU oracle cannot share the device's noise, SURVEY landmine 10); otherwise
    # they come from Philox keyed by (seed, layer, packed row, unit) with the seed drawn from torch's CPU generator,
    # so `torch.manual_seed` makes a run repeatable and the mask does not depend on the launch schedule.
    drop_factors_fn = None

    def _dropout_factors(self, NT, dev):
        if not (self.training and self.dropout > 0 and self.n_layers > 1):
            return None
        H, L = self.rnn_cell_size, self.n_layers
        Hp = _round_up(H, 32)
        if self.drop_factors_fn is not None:
            f = self.drop_factors_fn(L - 1, NT, H).to(device=dev, dtype=torch.float32)
            if tuple(f.shape) != (L - 1, NT, H):
                raise ValueError()
            return _pad_units(f, H, Hp, 2)
        if not 0.0 < self.dropout < 1.0:
            raise ValueError('dropout must lie in [0, 1)')
        out = torch.empty(L - 1, NT, Hp, dtype=torch.float32, device=dev)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        _C.check(_lib.ppk_dropout_factors(seed, float(self.dropout), L - 1, NT, Hp, H, _C.p(out), _C.stream()))
        return out

    def init_state(self, batch_size: Optional[int] = None):
        """Zero hidden state [L, B, H] and empty stack [B, D, W] (upstream `init_hidden` / `init_stack`)."""
        B = int(batch_size if batch_size is not None else self.batch_size)
        dev = self.embedding.weight.device
        return (torch.zeros(self.n_layers, B, self.rnn_cell_size, dtype=torch.float32, device=dev),
                torch.zeros(B, self.stack_depth, self.stack_width, dtype=torch.float32, device=dev))

    # upstream attributes (`_update_batch_size` re-allocates them, which is why the reference calls it
    # before every step, reactions_vae.py:62-63; read by latent_space_map_fns.py:289).  Here they follow `batch_size`
    # and the module's current device by construction.
    @property
    def init_hidden(self):
        return self.init_state()[0]

    @property
    def init_stack(self):
        return self.init_state()[1]

    def forward(self, token, hidden, stack):
        """One StackGRU step (the call of paccmann_chemistry's generation loop that the
        reference reaches through paccmann_encoder.py:115-131): token [1, B] or [B] int64, hidden [L, B, H],
        stack [B, D, W] -> (out [1, B, H], hidden, stack).  Inference only (no autograd): training goes
        through the packed-sequence path."""
        tok = token.reshape(-1).to(self.embedding.weight.device, torch.int64).contiguous()
        B = int(tok.numel())
        if hidden.shape != (self.n_layers, B, self.rnn_cell_size) or stack.shape != (B, self.stack_depth, self.stack_width):
            raise ValueError()
        with torch.no_grad():
            r = _seq_forward(tok, torch.tensor([B], dtype=torch.int64), self._dims(), hidden.to(torch.float32), stack,
                             self.embedding.weight, self.stack_controls_layer.weight, self.stack_controls_layer.bias,
                             self.stack_input_layer.weight, self.stack_input_layer.bias, self._gru_params())
            H = self.rnn_cell_size
            new_hidden = r['Hs'][:, :, :H].contiguous()
            return new_hidden[-1].unsqueeze(0), new_hidden, r['S']


def _require_packed(x):
    if not isinstance(x, PackedSequence):
        raise TypeError('the HIP path takes PackedSequence batches (batch_mode="packed")')


def _final_row_index(packed: PackedSequence, device):
    """Packed row of each ORIGINAL sample's last step."""
    bs = packed.batch_sizes.to(torch.int64)
    T = bs.numel()
    B = int(bs[0])
    off = torch.zeros(T + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(bs, 0)
    # length of sorted sequence s = number of steps with batch_size > s
    lengths = (bs.unsqueeze(1) > torch.arange(B).unsqueeze(0)).sum(0)
    fin_sorted = off[lengths - 1] + torch.arange(B)
    if packed.unsorted_indices is not None:
        fin = fin_sorted[packed.unsorted_indices.cpu()]
    else:
        fin = fin_sorted
    return fin.to(device)


class StackGRUEncoder(StackGRU):

    def __init__(self, params):
        super().__init__(params)
        self.latent_dim = self.params['latent_dim']
        self.hidden_to_mu = HIPLinear(self.rnn_cell_size, self.latent_dim)
        self.hidden_to_logvar = HIPLinear(self.rnn_cell_size, self.latent_dim)

    def encoder_train_step(self, packed: PackedSequence):
        """-> (mu, logvar) [B, Z] in ORIGINAL sample order."""
        _require_packed(packed)
        dev = self.embedding.weight.device
        B = int(packed.batch_sizes[0])
        h0 = torch.zeros(self.n_layers, B, self.rnn_cell_size, dtype=torch.float32, device=dev)
        htop = self._run_sequence(packed, h0)
        hfin = _GatherRows.apply(htop, _final_row_index(packed, dev))
        mu = linear(hfin, self.hidden_to_mu.weight, self.hidden_to_mu.bias)
        logvar = linear(hfin, self.hidden_to_logvar.weight, self.hidden_to_logvar.bias)
        return mu, logvar



class StackGRUDecoder(StackGRU):

    def __init__(self, params):
        super().__init__(params)
        self.latent_dim = self.params['latent_dim']
        self.latent_to_hidden = HIPLinear(self.latent_dim, self.rnn_cell_size)
        self.output_layer = HIPLinear(self.rnn_cell_size, self.vocab_size)

    def decoder_train_step(self, latent_z, packed_in: PackedSequence, packed_tgt: PackedSequence):
        """latent_z [1 or L, B, Z] in ORIGINAL sample order -> mean CE over non-pad targets."""
        _require_packed(packed_in)
        _require_packed(packed_tgt)
        dev = self.embedding.weight.device
        L = self.n_layers
        if latent_z.dim() == 2:
            latent_z = latent_z.unsqueeze(0)
        if latent_z.shape[0] not in (1, L):
            raise ValueError('latent_z must be [1, B, Z] or [n_layers, B, Z]')
        B = latent_z.shape[1]
        if B != int(packed_in.batch_sizes[0]):
            raise ValueError('latent batch does not match the packed batch')
        sidx = packed_in.sorted_indices
        hs = []
        for l in range(latent_z.shape[0]):
            zl = latent_z[l]
            if sidx is not None:
                zl = _GatherRows.apply(zl, sidx.to(dev))
            hs.append(linear(zl, self.latent_to_hidden.weight, self.latent_to_hidden.bias))
        h0 = torch.stack(hs * (L // len(hs)), dim=0)
        htop = self._run_sequence(packed_in, h0)
        logits = linear(htop, self.output_layer.weight, self.output_layer.bias)
        tgt = packed_tgt.data.reshape(-1).to(device=dev, dtype=torch.int64).contiguous()
        return _CrossEntropy.apply(logits, tgt, 0)



class GreedySearch:
    """argmax of the logits (stand-in for paccmann_chemistry.utils.search.GreedySearch)."""

    def step(self, logits):
        return logits.argmax(dim=-1)


class SamplingSearch:
    """multinomial sample of softmax(logits / temperature) with the torch global RNG (stand-in for
    paccmann_chemistry.utils.search.SamplingSearch, the default of paccmann_encoder.py:115)."""

    def __init__(self, temperature: float = 1.0):
        self.temperature = float(temperature)

    def step(self, logits):
        probs = torch.softmax(logits / self.temperature, dim=-1)
        return torch.multinomial(probs, 1).squeeze(-1)


class TeacherVAE(nn.Module):

    def __init__(self, encoder: StackGRUEncoder, decoder: StackGRUDecoder):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self._last_kl = None

    def encode(self, input_seq):
        return self.encoder.encoder_train_step(input_seq)

    def reparameterize(self, mu, logvar, eps: Optional[torch.Tensor] = None):
        """z = eps * exp(0.5 logvar) + mu.  `eps` may be injected (parity tests);
        otherwise it is drawn with torch.randn_like as upstream does.  The fused
        kernel also yields the KL term, kept in `self.last_kl`."""
        if eps is None:
            eps = torch.randn_like(mu)
        z, kl = _ReparamKL.apply(mu, logvar, eps.to(mu.device))
        self._last_kl = kl
        return z

    @property
    def last_kl(self):
        return self._last_kl

    def decode(self, latent_z, input_seq, target_seq):
        return self.decoder.decoder_train_step(latent_z, input_seq, target_seq)

    def forward(self, input_seq, decoder_seq, target_seq, eps=None):
        mu, logvar = self.encode(input_seq)
        z = self.reparameterize(mu, logvar, eps).unsqueeze(0)
        return self.decode(z, decoder_seq, target_seq), mu, logvar

    def generate(self, latent_z, prime_input, end_token, generate_len: int = 100, search=None):
        """Token generation from latent codes (the upstream `TeacherVAE.generate` call of
        paccmann_encoder.py:119-127; its body is not in the reference tree, so this is BUILD-DEFINED and
        parity-unpinned): hidden = latent_to_hidden(z) on every layer, empty stack, feed `prime_input`, then
        `generate_len` single steps, the next token chosen by `search` from the logits.  Yields one int64
        tensor per sample: the generated tokens up to and including the first `end_token`."""
        dec = self.decoder
        search = search if search is not None else SamplingSearch()
        with torch.no_grad():
            z = latent_z.reshape(-1, latent_z.shape[-1]).to(dec.embedding.weight.device, torch.float32)
            B = z.shape[0]
            h = dec.latent_to_hidden(z)
            hidden = h.unsqueeze(0).repeat(dec.n_layers, 1, 1).contiguous()
            _, stack = dec.init_state(B)
            token = prime_input.reshape(-1)[:1].to(z.device, torch.int64).repeat(B)
            end = int(end_token.reshape(-1)[0])
            out_tokens = []
            for _ in range(int(generate_len)):
                out, hidden, stack = dec(token, hidden, stack)
                logits = dec.output_layer(out.squeeze(0))
                token = search.step(logits).reshape(-1).to(torch.int64)
                out_tokens.append(token)
            tokens = torch.stack(out_tokens, dim=1).cpu() if out_tokens else torch.zeros(B, 0, dtype=torch.int64)
        for row in tokens:
            hit = (row == end).nonzero()
            yield row[:int(hit[0]) + 1] if hit.numel() else row

    def save(self, path, *args, **kwargs):
        torch.save(self.state_dict(), path, *args, **kwargs)

    def load(self, path, *args, **kwargs):
        self.load_state_dict(torch.load(path, *args, weights_only=True, **kwargs))
