"""Host-side batch / index construction (bit-exact integers) against torch's
pack_sequence and networkx run on the reference's own statements
(chem_reactions/utils.py:35-98,155-196; reactions_vae.py:79-83)."""
import networkx as nx
import numpy as np
import pytest
import torch
from torch.nn.utils.rnn import pack_sequence

from paccmann_polymer_amd import data_prep as DP


def test_packed_preparation_matches_pack_sequence():
    rng = np.random.default_rng(0)
    seqs = [torch.tensor(np.concatenate([[2], rng.integers(4, 30, size=n), [3]])) for n in (4, 7, 3, 7, 1)]
    enc, dec, tgt = DP.packed_sequential_data_preparation(seqs, input_keep=1., device='cpu')
    ref = pack_sequence(seqs, enforce_sorted=False)
    ref_t = pack_sequence([torch.cat([s[1:], torch.tensor([0])]) for s in seqs], enforce_sorted=False)
    for got, want in ((enc, ref), (dec, ref), (tgt, ref_t)):
        assert torch.equal(got.data, want.data)
        assert torch.equal(got.batch_sizes, want.batch_sizes)
        assert torch.equal(got.sorted_indices, want.sorted_indices)
        assert torch.equal(got.unsorted_indices, want.unsorted_indices)
    # SURVEY Appendix D.4 example: lengths (4, 7, 3)
    e2, _, _ = DP.packed_sequential_data_preparation([torch.arange(4), torch.arange(7), torch.arange(3)], device='cpu')
    assert e2.batch_sizes.tolist() == [3, 3, 3, 2, 1, 1, 1] and e2.sorted_indices.tolist() == [1, 0, 2]


def test_packed_preparation_rejects_2d_and_drops_tokens():
    with pytest.raises(ValueError):
        DP.packed_sequential_data_preparation([torch.zeros(3, 2)], device='cpu')
    torch.manual_seed(0)
    seqs = [torch.arange(2, 40)]
    enc, dec, _ = DP.packed_sequential_data_preparation(seqs, input_keep=0.5, device='cpu')
    changed = dec.data != enc.data
    assert changed.any() and bool((dec.data[changed] == 1).all())


def reference_hops(edge_index, n):
    # the statements of reactions_vae.py:79-82
    graph = nx.from_edgelist(edge_index.T.tolist())
    graph.add_nodes_from(list(range(n)))
    dists = nx.floyd_warshall_numpy(graph)
    dists[np.isinf(dists)] = 0
    return np.asarray(dists), list(graph)


def test_hop_distance_networkx_order_quirk_c():
    ei = torch.tensor([[0, 0, 1, 1, 1, 1, 2, 2], [1, 2, 3, 5, 7, 8, 4, 6]])      # SURVEY §0.4 example
    want, order = reference_hops(ei, 9)
    assert order == [0, 1, 2, 3, 5, 7, 8, 4, 6]
    assert DP.networkx_node_order(ei, 9) == order
    got = DP.hop_distance_matrix(ei, 9, order='networkx')
    assert got.dtype == np.float64 and np.array_equal(got, want)
    by_index = DP.hop_distance_matrix(ei, 9, order='index')
    assert not np.array_equal(by_index, want)
    perm = np.argsort(order)
    assert np.array_equal(by_index, want[np.ix_(perm, perm)])


@pytest.mark.parametrize('seed', range(6))
def test_hop_distance_random_graphs(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(2, 22))
    e = int(rng.integers(0, 25))
    ei = torch.tensor(rng.integers(0, n, size=(2, e)))
    want, _ = reference_hops(ei, n) if e else (np.zeros((n, n)), None)
    got = DP.hop_distance_matrix(ei, n)
    assert np.array_equal(got, want)                 # includes isolated nodes and disconnected parts (inf -> 0)


@pytest.mark.parametrize('seed', range(5))
def test_bundle_sampler_and_relabel_match_networkx(seed):
    G = nx.barabasi_albert_graph(200, 3, seed=seed)
    root = seed * 7 % 200
    # the statements of utils.py:161-179 on the global numpy RNG
    np.random.seed(seed)
    close = np.random.permutation(list(G.neighbors(root)))[:5]
    total_far = sum([len(list(G.neighbors(x))) for x in close])
    shortlist = [[root, x] for x in close]
    counter = 0
    while len(shortlist) < 20 and counter < total_far:
        cur = close[counter % len(close)]
        shortlist.append([cur, np.random.choice(list(G.neighbors(cur)))])
        counter += 1
    np.random.seed(seed)
    mine = DP.sample_bundle(G.neighbors, root)
    assert [[int(a), int(b)] for a, b in mine] == [[int(a), int(b)] for a, b in shortlist]
    # utils.py:183-194
    sub = nx.convert_node_labels_to_integers(nx.from_edgelist(shortlist), label_attribute='original')
    want_ei = np.asarray(nx.to_pandas_edgelist(sub).values, dtype=np.int64).T
    want_labels = [int(sub.nodes[i]['original']) for i in range(len(sub))]
    labels, ei = DP.relabel_bundle(shortlist)
    assert labels == want_labels
    assert np.array_equal(ei, want_ei)


@pytest.mark.parametrize('B,lo,hi,seed', [(1, 1, 1, 0), (300, 1, 6, 1), (2048, 20, 100, 2), (17, 5, 5, 3)])
def test_packed_preparation_vectorised_path_is_bit_exact(B, lo, hi, seed):
    """input_keep == 1 takes the loop-free path: same data, batch_sizes and tie order as pack_sequence."""
    rng = np.random.default_rng(seed)
    seqs = [torch.tensor(rng.integers(2, 60, size=int(rng.integers(lo, hi + 1)))) for _ in range(B)]
    enc, dec, tgt = DP.packed_sequential_data_preparation(seqs, input_keep=1., device='cpu')
    ref = pack_sequence(seqs, enforce_sorted=False)
    ref_t = pack_sequence([torch.cat([s[1:], torch.tensor([0])]) for s in seqs], enforce_sorted=False)
    for got, want in ((enc, ref), (dec, ref), (tgt, ref_t)):
        assert torch.equal(got.data, want.data)
        assert torch.equal(got.batch_sizes, want.batch_sizes) and got.batch_sizes.device.type == 'cpu'
        assert torch.equal(got.sorted_indices, want.sorted_indices)
        assert torch.equal(got.unsorted_indices, want.unsorted_indices)
    dec.data[0] = -7                                   # the decoder stream is its own storage (reference: input.clone())
    assert enc.data[0] != -7


@pytest.mark.parametrize('B,lo,hi,seed', [(1, 1, 1, 0), (300, 1, 6, 1), (2048, 20, 100, 2), (17, 5, 5, 3), (21, 1, 40, 4)])
def test_packed_preparation_device_arithmetic_is_bit_exact(B, lo, hi, seed):
    """The packer that runs on the device (position -> (step, row) -> source index by searchsorted + gathers): the
    same function evaluated on the CPU device against pack_sequence; the GPU run is in tests/test_gpu_stackgru.py."""
    rng = np.random.default_rng(seed)
    seqs = [torch.tensor(rng.integers(2, 60, size=int(rng.integers(lo, hi + 1)))) for _ in range(B)]
    enc, dec, tgt = DP._packed_preparation_device(seqs, 'cpu')
    ref = pack_sequence(seqs, enforce_sorted=False)
    ref_t = pack_sequence([torch.cat([s[1:], torch.tensor([0])]) for s in seqs], enforce_sorted=False)
    for got, want in ((enc, ref), (dec, ref), (tgt, ref_t)):
        assert torch.equal(got.data, want.data)
        assert torch.equal(got.batch_sizes, want.batch_sizes) and got.batch_sizes.device.type == 'cpu'
        assert torch.equal(got.sorted_indices, want.sorted_indices)
        assert torch.equal(got.unsorted_indices, want.unsorted_indices)
    assert DP.known_token_range(enc.data) == (int(ref.data.min()), int(ref.data.max()))
