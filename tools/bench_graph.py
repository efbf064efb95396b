"""Roofline measurements of the gather / segmented-reduce kernels at BASELINE.json configs[2] and [3]
(SURVEY.md §8d): K6 CSR build + K7 SpMM on a 5 M-edge synthetic molecular graph batch, and K5 sparse
graph-regularisation loss on a 1 M-node / 10 M-edge synthetic relation graph.

    python tools/bench_graph.py [--small]          (one MI355X; prints one JSON line per kernel)

Timed with torch.cuda events on the current stream (the stream every ppk op is launched on), inputs
resident in HBM.  `achieved` = algorithmic bytes (SURVEY §8d formulas) / mean launch time.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PEAK_HBM_GBS = 8000.0


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def line(kernel, workload, seconds, nbytes, extra=None):
    gbs = nbytes / seconds / 1e9
    d = dict(kernel=kernel, workload=workload, ms=round(seconds * 1e3, 4),
             roofline=dict(bound='hbm', achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(gbs / PEAK_HBM_GBS, 4),
                           algorithmic_bytes=int(nbytes)))
    if extra:
        d.update(extra)
    print(json.dumps(d), flush=True)


def molecule_batch(rng, n_nodes):
    """Blocks of 20..60 atoms: a random tree per block plus ~8 % ring-closing bonds; each bond listed once."""
    sizes = []
    left = n_nodes
    while left > 0:
        s = int(min(left, rng.integers(20, 61)))
        sizes.append(s)
        left -= s
    sizes = np.asarray(sizes)
    starts = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    block = np.repeat(np.arange(len(sizes)), sizes)
    first = starts[block]
    idx = np.arange(n_nodes)
    child = idx[idx > first]
    parent = first[child] + (rng.random(child.size) * (child - first[child])).astype(np.int64)
    n_ring = int(0.08 * n_nodes)
    a = rng.integers(0, n_nodes, size=n_ring)
    b = first[a] + (rng.random(n_ring) * sizes[block[a]]).astype(np.int64)
    keep = a != b
    src = np.concatenate([parent, a[keep]])
    dst = np.concatenate([child, b[keep]])
    return np.stack([src, dst]).astype(np.int64)


def relation_graph(rng, n_nodes, m):
    """Preferential-attachment-like simple graph: node t links to m earlier nodes drawn towards low indices."""
    src = np.repeat(np.arange(1, n_nodes, dtype=np.int64), m)
    dst = (src * rng.random(src.size) ** 1.5).astype(np.int64)
    lo, hi = np.minimum(src, dst), np.maximum(src, dst)
    key = np.unique(lo * n_nodes + hi)
    key = key[(key // n_nodes) != (key % n_nodes)]
    return np.stack([key // n_nodes, key % n_nodes]).astype(np.int64)


def config3(small, kind='molecules'):
    """kind: 'molecules' (blocks of 20..60 atoms: neighbours sit within a few KB of each other — cache friendly),
    'random' (the same number of bonds between uniformly random endpoints: every gathered row is a cache miss),
    'hubs' (random endpoints with a third of the directed edges pointing at 3 nodes: K7's long-row path)."""
    from paccmann_polymer_amd import gcn
    rng = np.random.default_rng(0)
    N = 240_000 if small else 2_400_000
    F = 256
    bonds = molecule_batch(rng, N)
    if kind != 'molecules':
        nb = bonds.shape[1]
        a, b = rng.integers(0, N, size=nb), rng.integers(0, N, size=nb)
        keep = a != b
        bonds = np.stack([a[keep], b[keep]]).astype(np.int64)
    if kind == 'hubs':
        ei_np = np.concatenate([bonds, bonds[::-1]], axis=1)
        ei_np[1, : ei_np.shape[1] // 3] = rng.integers(0, 3, size=ei_np.shape[1] // 3)
        ei = torch.tensor(ei_np).cuda()
    else:
        ei = torch.tensor(np.concatenate([bonds, bonds[::-1]], axis=1)).cuda()      # symmetrised: directed edges
    E = int(ei.shape[1])
    x = torch.randn(N, F, device='cuda')
    wl = f'configs[2]: GCN message passing, {kind} graph, N={N} nodes, E={E} directed edges (+N self loops), F={F}'
    t_build = timeit(lambda: gcn.CSRGraph(ei, N), iters=3, warm=1)
    line('K6 csr_build (forward + reverse CSR, host wrapper)', wl, t_build, 2 * 40.0 * (E + N))
    g = gcn.CSRGraph(ei, N)
    nnz = E + N
    by = nnz * (4 * F + 12) + N * (4 * F + 8)
    line('K7 k_spmm forward (A_hat x)', wl, timeit(lambda: gcn._spmm(g.fwd, x, g.nnz_cap)), by)
    line('K7 k_spmm backward (A_hat^T dout, reverse CSR)', wl, timeit(lambda: gcn._spmm(g.rev, x, g.nnz_cap)), by)
    if kind != 'molecules':
        return
    conv = gcn.GCNConv(F, 512).cuda()
    with torch.no_grad():
        t_conv = timeit(lambda: conv(x, g), iters=5)
    print(json.dumps(dict(kernel='GCNConv(256, 512) forward = K7 + fp32 MFMA GEMM', workload=wl, ms=round(t_conv * 1e3, 3),
                          gemm_tflops=round(2.0 * N * F * 512 / t_conv / 1e12, 1))), flush=True)
    xg = x.clone().requires_grad_(True)
    up = torch.randn(N, 512, device='cuda')

    def fb():
        xg.grad = None
        conv.zero_grad(set_to_none=True)
        conv(xg, g).backward(up)
    t_fb = timeit(fb, iters=5)
    print(json.dumps(dict(kernel='GCNConv(256, 512) forward + backward (2 x K7, 3 GEMMs, bias column sum)', workload=wl,
                          ms=round(t_fb * 1e3, 3), gemm_tflops_if_all_gemm=round(6.0 * N * F * 512 / t_fb / 1e12, 1))), flush=True)


def config4(small):
    from paccmann_polymer_amd.graph_loss_sparse import SparseRelationGraph, graph_loss_sparse, _SparseGraphLoss
    rng = np.random.default_rng(1)
    N = 100_000 if small else 1_000_000
    Z = 256
    und = relation_graph(rng, N, 10)
    E = int(und.shape[1])
    attr = torch.tensor(rng.integers(1, 4, size=E).astype(np.float32)).cuda()
    graph = SparseRelationGraph(torch.tensor(und).cuda(), attr, N)
    z = torch.randn(N, Z, device='cuda', requires_grad=True)
    bal = graph.anchor_ranges(8)
    cases = [('all anchors (1 GPU holds the whole graph)', (0, N), 1.0)]
    rp = graph.rowptr.cpu()
    for name, starts in (('equal node counts', [N * r // 8 for r in range(9)]), ('work-balanced anchor ranges', bal)):
        for r in range(8):
            a0, a1 = starts[r], starts[r + 1]
            cases.append((f'rank {r} of 8, {name}', (a0, a1), float(rp[a1] - rp[a0]) / float(2 * E)))
    plan = graph.halo_plan(8)
    halo = [sum(int(plan[r][q].numel()) for q in range(8) if q != r) for r in range(8)]
    print(json.dumps(dict(kernel='K5 exchange, 8 ranks: latent rows a rank receives before its kernels (and sends gradient rows for afterwards)',
                          workload=f'configs[3]: N={N} nodes, E={E} undirected power-law edges, Z={Z}',
                          halo_rows_per_rank=halo, allgather_rows_per_rank=N - N // 8,
                          halo_over_allgather=round(float(np.mean(halo)) / (N - N // 8), 4),
                          note='row counts from the exchange plan; the exchange itself is NOT timed: no multi-GPU node was available')), flush=True)
    worst = {}
    for share, (a0, a1), frac in cases:
        wl = f'configs[3]: sparse graph-reg loss, N={N} nodes, E={E} undirected edges, Z={Z}, {share}'
        # per UNDIRECTED edge two latent rows, the int64 pair and the attribute, plus d / g / coefficient scratch
        by_f = E * frac * (2 * 4 * Z + 24) + 3 * E * frac * 4
        loss = _SparseGraphLoss.apply(z, graph, a0, a1, False)
        t_f = timeit(lambda: _SparseGraphLoss.apply(z.detach(), graph, a0, a1, False), iters=5)

        def fb():
            z.grad = None
            _SparseGraphLoss.apply(z, graph, a0, a1, False).backward()
        t_fb = timeit(fb, iters=5)
        if share.startswith('rank'):
            key = share.split(', ')[1]
            if key not in worst or t_fb > worst[key][1]:
                worst[key] = (t_f, t_fb, by_f, share)
            continue
        line('K5 sparse graph loss forward (dist + terms + finish)', wl, t_f, by_f, dict(loss=float(loss)))
        line('K5 sparse graph loss forward + backward', wl, t_fb, 2 * by_f + N * Z * 4)
    for key, (t_f, t_fb, by_f, share) in worst.items():
        wl = (f'configs[3]: sparse graph-reg loss, N={N} nodes, E={E} undirected edges, Z={Z}, SLOWEST of 8 ranks with {key} '
              f'({share.split(",")[0]}; each rank timed alone on one GPU)')
        line('K5 sparse graph loss forward, one rank of 8', wl, t_f, by_f)
        line('K5 sparse graph loss forward + backward, one rank of 8', wl, t_fb, 2 * by_f + N * Z * 4 / 8)


def dense(small):
    """K4 on ONE dense graph (SURVEY config 2: N = 2048, 4.29e9 reference terms): VALU/LDS-bound, priced in
    f-evaluations per second (each reference term is evaluated in both of its roles: N^2 (N-1) evaluations)."""
    from paccmann_polymer_amd import graph_loss
    import networkx as nx
    for N in ((64, 512) if small else (64, 2048)):
        g = nx.gnm_random_graph(N, 3 * N, seed=0)
        G = np.asarray(nx.floyd_warshall_numpy(g))
        G[np.isinf(G)] = 0
        Gd = torch.tensor(G, dtype=torch.float64).cuda()
        z = torch.randn(N, 256, device='cuda', requires_grad=True)

        def fb():
            z.grad = None
            graph_loss(z, Gd).backward()
        t = timeit(fb, iters=5)
        evals = float(N) * N * (N - 1)
        print(json.dumps(dict(kernel='K4 dense graph loss forward + backward (k_dist, k_terms, k_dz)',
                              workload=f'one dense graph N={N}, Z=256 ({int(evals // 2)} reference terms)', ms=round(t * 1e3, 4),
                              f_evals_per_s=round(evals / t / 1e9, 2), unit='G f-evaluations/s')), flush=True)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--small', action='store_true')
    ap.add_argument('--only', default='all', help='dense | spmm | spmm_random | spmm_hubs | sparse | all')
    args = ap.parse_args()
    if args.only in ('all', 'dense'):
        dense(args.small)
    if args.only in ('all', 'spmm'):
        config3(args.small)
    if args.only in ('all', 'spmm_random'):
        config3(args.small, 'random')
    if args.only in ('all', 'spmm_hubs'):
        config3(args.small, 'hubs')
    if args.only in ('all', 'sparse'):
        config4(args.small)
