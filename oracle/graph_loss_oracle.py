"""ORACLE (test infrastructure, NOT product code).

CPU restatements of the reference graph-regularisation loss
(`/root/reference/paccmann_polymer/topologically_regularized_models/
graph_constrained_loss.py:4-67`):

* `graph_loss_loop`      – literal pure-Python triple loop (small N only),
* `graph_loss_c`         – the same loop in C (`graph_loss_ref.c`), fp32
                           sequential accumulation like the reference, or
                           fp64 accumulation as the high-precision arbiter,
                           with the analytic gradient of SURVEY Appendix A.3,
* `graph_loss_closed_form` – vectorised numpy float64 closed form (A.1).

Parity pin: `tests/golden/graphreg_*.npz` were produced by the real
reference `graph_loss` (+ autograd) in the build container by
`tests/golden/make_golden.py`; `tests/test_oracle_graph_loss.py` checks all
three restatements against them.

Only tests/, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, 'liboracle.so')
    src = os.path.join(_HERE, 'graph_loss_ref.c')
    if force or not os.path.exists(so) or (
        os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)
    ):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'liboracle.so'])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build())
        f32p = ctypes.POINTER(ctypes.c_float)
        f64p = ctypes.POINTER(ctypes.c_double)
        lib.oracle_cdist.argtypes = [f32p, ctypes.c_int64, ctypes.c_int64, f32p]
        lib.oracle_cdist.restype = None
        lib.oracle_graph_loss_from_dists.argtypes = [
            f32p, f64p, ctypes.c_int64, ctypes.c_int, f64p]
        lib.oracle_graph_loss_from_dists.restype = ctypes.c_double
        lib.oracle_graph_loss.argtypes = [
            f32p, f64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, f64p]
        lib.oracle_graph_loss.restype = ctypes.c_double
        lib.oracle_graph_loss_min_margin.argtypes = [
            f32p, f64p, ctypes.c_int64, ctypes.c_int64]
        lib.oracle_graph_loss_min_margin.restype = ctypes.c_double
        lib.oracle_graph_loss_sparse.argtypes = [
            f32p, f64p, ctypes.POINTER(ctypes.c_ubyte), ctypes.c_int64, ctypes.c_int64, f64p]
        lib.oracle_graph_loss_sparse.restype = ctypes.c_double
        _LIB = lib
    return _LIB


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def cdist_c(z):
    z, zp = _f32(z)
    n, zd = z.shape
    D = np.empty((n, n), dtype=np.float32)
    _lib().oracle_cdist(zp, n, zd, D.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return D


def graph_loss_c(z, g_dists, acc_fp32=True, want_grad=True):
    """-> (loss: float, dz: float64 [N, Z] or None)."""
    z, zp = _f32(z)
    g, gp = _f64(g_dists)
    n, zd = z.shape
    if g.shape != (n, n):
        raise ValueError()   # graph_constrained_loss.py:42-43
    dz = np.zeros((n, zd), dtype=np.float64) if want_grad else None
    dzp = dz.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) if want_grad else None
    loss = _lib().oracle_graph_loss(zp, gp, n, zd, 1 if acc_fp32 else 0, dzp)
    return float(loss), dz


def graph_loss_sparse_c(z, edge_index, edge_attr):
    """Sparse generalisation on an undirected edge list [2, E] with
    attribute [E] -> (loss, dz float64 [N, Z])."""
    z, zp = _f32(z)
    n, zd = z.shape
    ei = np.asarray(edge_index, dtype=np.int64)
    G = np.zeros((n, n), dtype=np.float64)
    adj = np.zeros((n, n), dtype=np.uint8)
    for (u, v), a in zip(ei.T, np.asarray(edge_attr, dtype=np.float64)):
        G[u, v] = G[v, u] = a
        adj[u, v] = adj[v, u] = 1
    dz = np.zeros((n, zd), dtype=np.float64)
    loss = _lib().oracle_graph_loss_sparse(
        zp, G.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), adj.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)),
        n, zd, dz.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return float(loss), dz


def min_margin(z, g_dists):
    z, zp = _f32(z)
    g, gp = _f64(g_dists)
    return float(_lib().oracle_graph_loss_min_margin(zp, gp, z.shape[0], z.shape[1]))


def graph_loss_segmented_c(z, g_list, seg_sizes, acc_fp32=True):
    """loss = sum_b graph_loss(z_b, G_b); dz concatenated."""
    z = np.ascontiguousarray(z, dtype=np.float32)
    off = 0
    total = 0.0
    dzs = []
    for n, g in zip(seg_sizes, g_list):
        l, dz = graph_loss_c(z[off:off + n], g, acc_fp32=acc_fp32)
        total += l
        dzs.append(dz)
        off += n
    return total, np.concatenate(dzs, axis=0)


# --------------------------------------------------------------------------
# literal pure-Python loop (graph_constrained_loss.py:25-67), tiny N only
# --------------------------------------------------------------------------
def _f(delta, di, dj):
    # graph_constrained_loss.py:4-10
    if delta == 0:
        return (di - dj) ** 2
    elif delta > 0:
        return max(dj - di, np.float32(0))
    return max(di - dj, np.float32(0))


def graph_loss_loop(z, g_dists):
    D = cdist_c(z)
    G = np.asarray(g_dists, dtype=np.float64)
    if D.shape != G.shape:
        raise ValueError()
    mx = D.max()
    n = len(D)
    loss = np.float32(0)
    for i in range(n):
        for j in range(i + 1, n):
            d_ij, g_ij = D[i, j], G[i, j]
            k = 0
            while i + k < n:
                if i + k != j:
                    loss = np.float32(loss + np.float32(_f(g_ij - G[i + k, j], d_ij, D[i + k, j])) / mx)
                k += 1
            k = 0
            while k + j < n:
                if i != j + k:
                    d_k = D[i, j + k]
                    loss = np.float32(loss + np.float32(_f(g_ij - np.float64(d_k), d_ij, d_k)) / mx)
                k += 1
    return float(loss)


# --------------------------------------------------------------------------
# vectorised closed form, float64 arithmetic on the
# fp32 distance matrix
# --------------------------------------------------------------------------
def graph_loss_closed_form(z, g_dists):
    D32 = cdist_c(z)
    D = D32.astype(np.float64)
    G = np.asarray(g_dists, dtype=np.float64)
    n = len(D)
    if D.shape != G.shape:
        raise ValueError()
    idx = np.arange(n)

    def f(delta, a, b):
        return np.where(delta == 0, (a - b) ** 2,
                        np.where(delta > 0, np.maximum(b - a, 0), np.maximum(a - b, 0)))

    # column raster: (i, j, r), i<j, r>=i, r!=j ; compares (i,j) with (r,j)
    i, j, r = np.meshgrid(idx, idx, idx, indexing='ij')
    m = (i < j) & (r >= i) & (r != j)
    col = f(G[i, j] - G[r, j], D[i, j], D[r, j])[m].sum()
    # row raster: (i, j, c), i<j<=c ; compares (i,j) with (i,c), delta uses D (Quirk A)
    m = (i < j) & (r >= j)
    row = f(G[i, j] - D[i, r], D[i, j], D[i, r])[m].sum()
    return float((col + row) / D.max())


# --------------------------------------------------------------------------
# torch restatements WITH autograd: what the reference's step actually pays for on the CPU (the loss is built term
# by term from 0-dim tensors, graph_constrained_loss.py:44-66, and differentiated by autograd).  Timing legs of
# bench.py's cpu_baseline only.
# --------------------------------------------------------------------------
def graph_loss_loop_torch(z, g_dists):
    """Term-for-term restatement (same terms, same accumulation order, same torch ops per term: subtraction,
    pow / relu, division by the max, in-place add into a 0-dim tensor) of graph_constrained_loss.py:25-67.
    z: [N, Z] torch tensor (may require grad), g_dists: [N, N] torch tensor."""
    import torch
    D = torch.cdist(z, z, p=2.0).squeeze()                      # :13-22
    if D.shape != g_dists.shape:
        raise ValueError()
    top = D.max()                                               # :40
    n = len(D)
    total = torch.tensor(0.0)

    def term(delta, a, b):                                      # :4-10
        if delta == 0:
            return torch.pow(a - b, 2)
        if delta > 0:
            return torch.relu(b - a)
        return torch.relu(a - b)

    for i in range(n):
        for j in range(i + 1, n):
            a, ga = D[i, j], g_dists[i, j]
            for r in range(i, n):                               # down column j from row i (:49-57)
                if r != j:
                    total += term(ga - g_dists[r, j], a, D[r, j]) / top
            for c in range(j, n):                               # along row i from column j; the "graph" value read
                b = D[i, c]                                     # here is the LATENT distance (:61-62, SURVEY Quirk A)
                total += term(ga - b, a, b) / top
    return total


def graph_loss_vectorised_torch(z, g_dists):
    """Closed form of the same sum in a handful of torch tensor ops, with autograd."""
    import torch
    D = torch.cdist(z, z, p=2.0).squeeze()
    if D.shape != g_dists.shape:
        raise ValueError()
    n = len(D)
    G = g_dists.to(D.dtype)
    idx = torch.arange(n)
    i, j, r = torch.meshgrid(idx, idx, idx, indexing='ij')

    def f(delta, a, b):
        return torch.where(delta == 0, (a - b) ** 2, torch.where(delta > 0, torch.relu(b - a), torch.relu(a - b)))

    col = (f(G[i, j] - G[r, j], D[i, j], D[r, j]) * ((i < j) & (r >= i) & (r != j))).sum()
    row = (f(G[i, j] - D[i, r], D[i, j], D[i, r]) * ((i < j) & (r >= j))).sum()
    return (col + row) / D.max()
