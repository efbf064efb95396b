"""Device bundle sampler (ppk_sample_bundles) against the host restatement of the reference's statements
(/root/reference/.../chem_reactions/utils.py:155-196 -> data_prep.sample_bundle / relabel_bundle) driven by the same
Philox draws (oracle/bundle_sampler_ref.py): integers, bit-exact.  The reference's own numbers (numpy global RNG) are
not reproducible on a device; what is checked against them is the distribution (uniformity of the picks)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.bundle_sampler_ref import PhiloxRng  # noqa: E402
from paccmann_polymer_amd import data_prep  # noqa: E402

pytestmark = pytest.mark.gpu


def random_graph(rng, n, m, hubs=0):
    """edge list with duplicates, both orientations, a few self loops and (optionally) hub nodes; every node appears"""
    edges = [[i, (i + 1) % n] for i in range(n)]                      # a ring: nobody is isolated, ids 0..n-1 all present
    for _ in range(m):
        u, v = int(rng.integers(n)), int(rng.integers(n))
        edges.append([u, v])
    for h in range(hubs):
        for v in rng.choice(n, size=n // 2, replace=False):
            edges.append([h, int(v)])
    order = rng.permutation(len(edges))
    return [edges[i] for i in order]


def host_bundles(adj, roots, max_close, max_total, seed):
    out = []
    for root in roots:
        rng = PhiloxRng(seed, root, max_close)
        shortlist = data_prep.sample_bundle(lambda x: adj[x], root, max_close, max_total, rng=rng)
        labels, ei = data_prep.relabel_bundle(shortlist)
        out.append((shortlist, labels, ei))
    return out


@pytest.mark.parametrize('n,m,hubs,max_close,max_total,seed', [
    (40, 60, 0, 5, 20, 0), (200, 900, 2, 5, 20, 7), (64, 10, 0, 3, 8, 2 ** 40 + 5), (300, 2000, 3, 8, 32, 123), (5, 0, 0, 5, 20, 1),
])
def test_device_sampler_equals_host_statements_on_the_same_draws(n, m, hubs, max_close, max_total, seed):
    rng = np.random.default_rng(n + m)
    adj = data_prep.adjacency_lists(random_graph(rng, n, m, hubs))
    # relabel nodes by first appearance like load_data assumes ids 0..n-1: here ids are already 0..n-1
    assert sorted(adj) == list(range(n))
    rowptr, col = data_prep.adjacency_csr(adj)
    dev = {k: v.cpu().numpy() for k, v in data_prep.sample_bundles_device(rowptr, col, None, max_close, max_total, seed).items()}
    ref = host_bundles(adj, range(n), max_close, max_total, seed)
    for root, (shortlist, labels, ei) in enumerate(ref):
        ns, nn, ne = int(dev['n_short'][root]), int(dev['n_nodes'][root]), int(dev['n_edges'][root])
        assert dev['shortlist'][root, :ns].tolist() == [[int(a), int(b)] for a, b in shortlist], root
        assert dev['labels'][root, :nn].tolist() == [int(l) for l in labels], root
        assert np.array_equal(dev['edges'][root, :ne].T, ei.reshape(2, -1)), root
        # the reference's structural guarantees
        assert ns <= max_total and nn <= max_total + 1
        for u, v in dev['shortlist'][root, :ns].tolist():
            assert v in adj[u]


def test_roots_subset_and_run_to_run_reproducible():
    rng = np.random.default_rng(5)
    adj = data_prep.adjacency_lists(random_graph(rng, 500, 3000, 2))
    rowptr, col = data_prep.adjacency_csr(adj)
    roots = [499, 0, 17, 17, 250]
    a = data_prep.sample_bundles_device(rowptr, col, roots, 5, 20, 99)
    b = data_prep.sample_bundles_device(rowptr, col, roots, 5, 20, 99)
    full = data_prep.sample_bundles_device(rowptr, col, None, 5, 20, 99)
    for k in a:
        assert torch.equal(a[k], b[k])
        assert torch.equal(a[k], full[k][torch.tensor(roots, device=full[k].device)])      # a root's bundle does not depend on the others
    c = data_prep.sample_bundles_device(rowptr, col, roots, 5, 20, 100)
    assert not torch.equal(a['shortlist'], c['shortlist'])
    with pytest.raises(IndexError):
        data_prep.sample_bundles_device(rowptr, col, [500], 5, 20, 0)


def test_picks_are_uniform():
    """The reference's distribution: the close neighbours are a uniform sample without replacement of the root's
    neighbours, a far neighbour is uniform over its list (np.random.permutation / np.random.choice)."""
    deg = 12
    edges = [[0, v] for v in range(1, deg + 1)] + [[v, 100 + 10 * v + w] for v in range(1, deg + 1) for w in range(4)]
    n = 100 + 10 * deg + 4
    edges += [[i, i] for i in range(n)]                               # every id present (self loops keep ids dense)
    adj = data_prep.adjacency_lists(edges)
    rowptr, col = data_prep.adjacency_csr(adj)
    close_counts = np.zeros(n)
    far_counts = {}
    outs = [data_prep.sample_bundles_device(rowptr, col, [0], 5, 20, s) for s in range(400)]
    sl = torch.stack([o['shortlist'][0] for o in outs]).cpu().numpy()
    for s in range(400):
        close = sl[s, :5, 1]
        assert len(set(close.tolist())) == 5 and set(close.tolist()) <= set(range(0, deg + 1))
        close_counts[close] += 1
        for u, v in sl[s, 5:20]:
            far_counts.setdefault(int(u), []).append(int(v))
    # root 0 has neighbours 1..12 and itself (self loop): 13 candidates, 5 picked per trial
    cand = adj[0]
    exp = 400 * 5 / len(cand)
    chi2 = sum((close_counts[c] - exp) ** 2 / exp for c in cand)
    assert chi2 < 45.0, chi2                                          # 12 degrees of freedom: P(chi2 > 45) ~ 1e-5
    # far picks of one close node: uniform over its list
    u = max(far_counts, key=lambda k: len(far_counts[k]))
    vals, cnt = np.unique(far_counts[u], return_counts=True)
    assert set(vals.tolist()) <= set(adj[u])
    e = len(far_counts[u]) / len(adj[u])
    assert sum((c - e) ** 2 / e for c in cnt) + e * (len(adj[u]) - len(vals)) < 40.0


def test_load_data_device_sampler(tmp_path):
    rng = np.random.default_rng(3)
    n = 60
    edgelist = random_graph(rng, n, 150, 1)
    smiles = {('C' * (i % 7 + 1)) + 'N' * (i // 7 + 1): i for i in range(n)}
    fn = tmp_path / 'g.json'
    fn.write_text(json.dumps({'smiles2nodes': smiles, 'edgelist': edgelist}))

    class Lang:
        def smiles_to_token_indexes(self, s):
            return [2] + [3 if ch == 'C' else 4 for ch in s] + [1]
    data = data_prep.load_data(str(fn), Lang(), sampler='device', seed=11)
    adj = data_prep.adjacency_lists(edgelist)
    ref = host_bundles(adj, range(n), 5, 20, 11)
    inv = {v: k for k, v in smiles.items()}
    assert len(data) == n
    for d, (shortlist, labels, ei) in zip(data, ref):
        assert d.num_nodes == len(labels)
        assert torch.equal(d.edge_index, torch.tensor(ei).reshape(2, -1))
        for xt, l in zip(d.x, labels):
            assert xt.tolist() == Lang().smiles_to_token_indexes(inv[l])


def test_bad_neighbour_id_is_reported_not_dereferenced():
    rowptr = np.array([0, 2, 3, 4], dtype=np.int64)
    col = np.array([1, 7, 0, 0], dtype=np.int64)          # node 0 lists a neighbour 7 of a 3-node graph
    with pytest.raises(IndexError):
        for seed in range(8):                              # whichever draw picks it
            data_prep.sample_bundles_device(rowptr, col, None, 5, 20, seed)
