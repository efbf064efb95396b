"""Generates the golden vectors under tests/golden/ from the REAL reference.

Run only in the build container (needs /root/reference):

    python tests/golden/make_golden.py

It imports the in-tree reference modules that are importable offline
(SURVEY.md §8c):
  * paccmann_polymer/topologically_regularized_models/graph_constrained_loss.py
  * paccmann_polymer/topologically_regularized_models/experiments/cora/vae.py
and records inputs + outputs (loss, autograd gradients) as small .npz files.
The reference's source text is never copied; only numbers are stored.
"""
import os
import sys

import networkx as nx
import numpy as np
import torch

REF = '/root/reference'
sys.path.insert(0, REF)
from paccmann_polymer.topologically_regularized_models.graph_constrained_loss import (  # noqa: E402
    graph_loss, compute_graph_dists)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import graph_loss_oracle as O  # noqa: E402  (margin screening only)


def hop_matrix(n, m, seed):
    """Same statements as reactions_vae.py:79-82 on a random graph."""
    g0 = nx.gnm_random_graph(n, m, seed=seed)
    edges = list(g0.edges())
    graph = nx.from_edgelist(edges)
    graph.add_nodes_from(list(range(n)))
    d = nx.floyd_warshall_numpy(graph)
    d[np.isinf(d)] = 0
    return np.asarray(d, dtype=np.float64)


def well_conditioned_case(n, zd, seed, scale=1.0):
    """Draw z until no row-raster delta is within 1e-3 of a sign flip."""
    s = seed
    while True:
        rng = np.random.default_rng(s)
        z = (rng.standard_normal((n, zd)) * scale).astype(np.float32)
        G = hop_matrix(n, max(n - 1, int(1.5 * n)), s)
        if O.min_margin(z, G) > 1e-3:
            return z, G, s
        s += 1000


def run_ref(z, G, g_kind='tensor64'):
    zt = torch.tensor(z, requires_grad=True)
    if g_kind == 'tensor64':
        g = torch.tensor(G)
    elif g_kind == 'tensor32':
        g = torch.tensor(G, dtype=torch.float32)
    elif g_kind == 'numpy':
        g = G
    loss = graph_loss(zt, g)
    loss.backward()
    return loss.detach().numpy().astype(np.float32), zt.grad.numpy().astype(np.float32)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    # dense single graphs
    for n in (5, 8, 16, 21):
        for zd in (3, 16, 256):
            # scale latents so that distances straddle the hop counts (both delta signs occur)
            scale = 1.5 / np.sqrt(zd)
            z, G, s = well_conditioned_case(n, zd, 100 * n + zd, scale)
            loss, dz = run_ref(z, G)
            name = f'graphreg_dense_N{n}_Z{zd}'
            np.savez(os.path.join(HERE, name + '.npz'), z=z, G=G, loss=loss, dz=dz, seed=s)
            print(name, float(loss), float(np.abs(dz).max()))
    # input-type conventions (numpy G as in cora/utils_vae.py:34-38; fp32 tensor G as in topo_vae.py:137)
    z, G, s = well_conditioned_case(12, 8, 777, 0.6)
    for kind in ('numpy', 'tensor32'):
        loss, dz = run_ref(z, G, kind)
        name = f'graphreg_{kind}G_N12_Z8'
        np.savez(os.path.join(HERE, name + '.npz'), z=z, G=G, loss=loss, dz=dz, seed=s)
        print(name, float(loss))
    # non-integer graph distances (toy scripts use float edge_attr)
    rng = np.random.default_rng(5)
    Gf = np.round(rng.uniform(0.5, 3.0, (10, 10)), 2)
    Gf = np.triu(Gf, 1)
    Gf = Gf + Gf.T
    z = (rng.standard_normal((10, 4)) * 0.8).astype(np.float32)
    loss, dz = run_ref(z, Gf)
    np.savez(os.path.join(HERE, 'graphreg_floatG_N10_Z4.npz'), z=z, G=Gf, loss=loss, dz=dz, seed=5)
    print('graphreg_floatG_N10_Z4', float(loss))
    # segmented: 4 bundles x 16 nodes, per-bundle reference summed
    zs, Gs, total, dzs = [], [], 0.0, []
    for b in range(4):
        z, G, s = well_conditioned_case(16, 32, 4000 + b, 1.5 / np.sqrt(32))
        loss, dz = run_ref(z, G)
        zs.append(z); Gs.append(G); dzs.append(dz); total += float(loss)
    np.savez(os.path.join(HERE, 'graphreg_segmented_K4x16_Z32.npz'),
             z=np.concatenate(zs), G=np.stack(Gs), loss=np.float32(total),
             dz=np.concatenate(dzs), sizes=np.array([16] * 4))
    print('graphreg_segmented_K4x16_Z32', total)
    # ragged segmented: bundle sizes 3, 21, 9, 2
    zs, Gs, total, dzs, sizes = [], [], 0.0, [], [3, 21, 9, 2]
    for b, n in enumerate(sizes):
        z, G, s = well_conditioned_case(n, 16, 5000 + b, 1.5 / 4)
        loss, dz = run_ref(z, G)
        zs.append(z); Gs.append(G.ravel()); dzs.append(dz); total += float(loss)
    np.savez(os.path.join(HERE, 'graphreg_segmented_ragged_Z16.npz'),
             z=np.concatenate(zs), G=np.concatenate(Gs), loss=np.float32(total),
             dz=np.concatenate(dzs), sizes=np.array(sizes))
    print('graphreg_segmented_ragged_Z16', total)
    # config 1a: N=64, Z=256 (cdist matmul path, N>25)
    z, G, s = well_conditioned_case(64, 256, 64256, 1.5 / 16)
    loss, dz = run_ref(z, G)
    np.savez(os.path.join(HERE, 'graphreg_dense_N64_Z256.npz'), z=z, G=G, loss=loss, dz=dz, seed=s)
    print('graphreg_dense_N64_Z256', float(loss))


if __name__ == '__main__':
    main()
