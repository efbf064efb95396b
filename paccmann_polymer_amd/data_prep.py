"""Batch / index construction either side of the hot path (host side, integers).

Mirrors
  experiments/chem_reactions/utils.py:35-98    packed_sequential_data_preparation
  experiments/chem_reactions/reactions_vae.py:79-83   hop-distance matrix (Floyd–Warshall, networkx order)
  experiments/chem_reactions/utils.py:155-196  bundle sampler of load_data
(`hop_distance_matrices_device`: the same matrices for a whole step's bundles in one HIP launch)
with bit-exact integer outputs (tests/test_data_prep.py pins them against
torch.nn.utils.rnn.pack_sequence and networkx run on the same statements).
"""
import weakref
from typing import List, Sequence

import numpy as np
import torch
from torch.nn.utils.rnn import PackedSequence, pack_sequence


# Token-id ranges of device token streams built here, keyed by tensor identity.  The sequence kernels index a
# [V, 3H] table with the token id; torch's Embedding raises (CPU) or asserts (GPU) on an id outside [0, V).  The ids
# are host data while a batch is packed, so their range is recorded here for free and `StackGRU._run_sequence`
# checks it against vocab_size without a device-to-host sync; streams of unknown provenance are checked once
# (one sync) and remembered.
_TOKEN_RANGES = {}


def note_token_range(tensor, lo: int, hi: int):
    key = id(tensor)
    _TOKEN_RANGES[key] = (int(lo), int(hi))
    weakref.finalize(tensor, _TOKEN_RANGES.pop, key, None)


def known_token_range(tensor):
    """(min, max) recorded for this very tensor object, or None."""
    return _TOKEN_RANGES.get(id(tensor))


def packed_sequential_data_preparation(
    input_batch,
    input_keep=1,
    start_index=2,
    end_index=3,
    dropout_index=1,
    device=None,
    enforce_sorted=False,
):
    """Reference utils.py:35-98 (same signature; `start_index` / `end_index` are
    unused there too).  Returns [encoder_seq, decoder_seq, target_seq] as
    PackedSequences whose `.data` lives on `device`."""
    if device is None:
        device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')

    def _process_sample(sample):
        if len(sample.shape) != 1:
            raise ValueError
        inp = sample.long()
        decoder = inp.clone()
        if input_keep != 1:
            mask = torch.distributions.bernoulli.Bernoulli(input_keep).sample(inp.shape)
            decoder[mask.cpu() == 0] = dropout_index
        target = torch.cat([inp[1:].detach().clone(), torch.zeros(1, dtype=torch.int64, device=inp.device)])
        return inp, decoder, target

    if input_keep == 1 and len(input_batch) > 0 and all(torch.is_tensor(x) and x.dim() == 1 and x.numel() > 0 for x in input_batch):
        # (_packed_preparation_device does the index arithmetic on the GPU instead: measured 3.2-4.6 ms against 4.0 ms at
        # batch 2048 and 0.21 against 0.15 ms at 21 sequences — the time is the Python list -> one flat tensor part that
        # both share, so the host path stays the default)
        return _packed_preparation_vectorised(input_batch, device)
    batch = [_process_sample(s) for s in input_batch]
    out = []
    for entry in zip(*batch):
        # pack on the host (index arithmetic), ship the token stream once
        packed = pack_sequence([e.cpu() for e in entry], enforce_sorted=enforce_sorted)
        data_d = packed.data.to(device)
        if packed.data.numel():
            note_token_range(data_d, packed.data.min(), packed.data.max())
        out.append(PackedSequence(data_d, packed.batch_sizes,
                                  None if packed.sorted_indices is None else packed.sorted_indices.to(device),
                                  None if packed.unsorted_indices is None else packed.unsorted_indices.to(device)))
    return out


def _packed_preparation_vectorised(input_batch, device):
    """The three PackedSequences of `packed_sequential_data_preparation` for input_keep == 1 without a Python
    loop over samples: one concatenation, one sort (the same `torch.sort(lengths, descending=True)` that
    `pack_sequence(enforce_sorted=False)` performs, so ties order identically) and index arithmetic.  At batch
    2048 the per-sample path costs 75 ms on the host — longer than the whole GPU step — this one ~4 ms."""
    return packed_preparation_to_device(packed_preparation_host(input_batch), device)


def packed_preparation_host(input_batch, pin=False):
    """Host half of `_packed_preparation_vectorised` (input_keep == 1): everything that is index arithmetic, as CPU
    tensors (pinned when `pin`: an asynchronous copy engine transfer needs page-locked memory).  A worker thread can
    run it for batch t + 1 while the GPU works on batch t (trainer.train_bundled); `packed_preparation_to_device`
    is the other half."""
    lengths = torch.tensor([int(x.numel()) for x in input_batch], dtype=torch.int64)
    flat = torch.cat([x.reshape(-1) for x in input_batch]).long().cpu()
    starts = torch.cumsum(lengths, 0) - lengths
    sorted_len, sorted_idx = torch.sort(lengths, descending=True)
    unsorted_idx = torch.empty_like(sorted_idx)
    unsorted_idx[sorted_idx] = torch.arange(sorted_idx.numel(), dtype=torch.int64)
    T = int(sorted_len[0])
    valid = torch.arange(T, dtype=torch.int64).view(T, 1) < sorted_len.view(1, -1)
    batch_sizes = valid.sum(1)
    t_idx, i_idx = valid.nonzero(as_tuple=True)
    src = starts[sorted_idx[i_idx]] + t_idx
    data = flat[src]
    last = t_idx + 1 >= sorted_len[i_idx]
    tgt = flat[torch.where(last, src, src + 1)]
    tgt = torch.where(last, torch.zeros_like(tgt), tgt)
    out = dict(data=data, tgt=tgt, sorted_idx=sorted_idx, unsorted_idx=unsorted_idx)
    if pin and torch.cuda.is_available():
        out = {k: v.pin_memory() for k, v in out.items()}
    out.update(batch_sizes=batch_sizes, lo=int(flat.min()), hi=int(flat.max()))
    return out


def packed_preparation_to_device(host, device, non_blocking=False):
    """[encoder_seq, decoder_seq, target_seq] on `device` from `packed_preparation_host`'s tensors."""
    data_d = host['data'].to(device, non_blocking=non_blocking)
    tgt_d = host['tgt'].to(device, non_blocking=non_blocking)
    dec_d = data_d.clone()
    lo, hi = host['lo'], host['hi']
    note_token_range(data_d, lo, hi)
    note_token_range(dec_d, lo, hi)
    note_token_range(tgt_d, min(lo, 0), hi)
    si = host['sorted_idx'].to(device, non_blocking=non_blocking)
    ui = host['unsorted_idx'].to(device, non_blocking=non_blocking)
    bs = host['batch_sizes']
    return [PackedSequence(data_d, bs, si, ui), PackedSequence(dec_d, bs, si, ui), PackedSequence(tgt_d, bs, si, ui)]


def _packed_preparation_device(input_batch, device):
    """The same three PackedSequences with the packing done ON THE DEVICE: the host keeps only
    what PackedSequence needs there — the lengths, their sort (the `torch.sort(lengths, descending=True)` of
    `pack_sequence(enforce_sorted=False)`, so ties order identically) and `batch_sizes` (a histogram of the lengths) —
    and ships the concatenated token stream once; packed position p -> (timestep t, sorted row i) -> source index is
    index arithmetic on the device (a searchsorted over the step offsets and two gathers), no [T, B] mask anywhere.
    Integers bit-equal to the host path (tests/test_data_prep.py)."""
    lengths = torch.tensor([int(x.numel()) for x in input_batch], dtype=torch.int64)
    flat = torch.cat([x.reshape(-1) for x in input_batch]).long()
    lo, hi = int(flat.min()), int(flat.max())                      # host-side range for the token check (no device sync later)
    starts = torch.cumsum(lengths, 0) - lengths
    sorted_len, sorted_idx = torch.sort(lengths, descending=True)
    unsorted_idx = torch.empty_like(sorted_idx)
    unsorted_idx[sorted_idx] = torch.arange(sorted_idx.numel(), dtype=torch.int64)
    T = int(sorted_len[0])
    # batch_sizes[t] = number of sequences longer than t
    hist = torch.bincount(lengths, minlength=T + 1)
    batch_sizes = (lengths.numel() - torch.cumsum(hist, 0))[:T]
    off = torch.cumsum(batch_sizes, 0) - batch_sizes               # first packed position of step t
    n_tok = int(lengths.sum())
    flat_d = flat.to(device, non_blocking=True)
    off_d = off.to(device, non_blocking=True)
    si = sorted_idx.to(device, non_blocking=True)
    start_sorted_d = starts[sorted_idx].to(device, non_blocking=True)
    len_sorted_d = sorted_len.to(device, non_blocking=True)
    p = torch.arange(n_tok, dtype=torch.int64, device=device)
    t_idx = torch.searchsorted(off_d, p, right=True) - 1
    i_idx = p - off_d[t_idx]
    src = start_sorted_d[i_idx] + t_idx
    data_d = flat_d[src]
    last = t_idx + 1 >= len_sorted_d[i_idx]
    tgt_d = torch.where(last, torch.zeros_like(data_d), flat_d[torch.where(last, src, src + 1)])
    dec_d = data_d.clone()
    note_token_range(data_d, lo, hi)
    note_token_range(dec_d, lo, hi)
    note_token_range(tgt_d, min(lo, 0), hi)
    ui = unsorted_idx.to(device, non_blocking=True)
    return [PackedSequence(data_d, batch_sizes, si, ui), PackedSequence(dec_d, batch_sizes, si, ui),
            PackedSequence(tgt_d, batch_sizes, si, ui)]


def networkx_node_order(edge_index, num_nodes: int) -> List[int]:
    """Row order of `nx.floyd_warshall_numpy(graph)` for the graph built at
    reactions_vae.py:79-80: nodes in order of first appearance in the edge list
    (u then v per edge), then the remaining `range(num_nodes)` (Quirk C)."""
    ei = np.asarray(edge_index.cpu() if torch.is_tensor(edge_index) else edge_index, dtype=np.int64)
    order, seen = [], set()
    if ei.size:
        for u, v in ei.T.tolist():
            for x in (u, v):
                if x not in seen:
                    seen.add(x)
                    order.append(x)
    for x in range(num_nodes):
        if x not in seen:
            seen.add(x)
            order.append(x)
    return order


def hop_distance_matrix(edge_index, num_nodes: int, order: str = 'networkx') -> np.ndarray:
    """reactions_vae.py:79-83: undirected all-pairs hop counts, `inf -> 0`,
    float64.  `order='networkx'` reproduces the reference's row/column order
    (insertion order, Quirk C); `order='index'` orders rows by node index."""
    ei = np.asarray(edge_index.cpu() if torch.is_tensor(edge_index) else edge_index, dtype=np.int64)
    if order == 'networkx':
        nodes = networkx_node_order(ei, num_nodes)
    elif order == 'index':
        labels = set(range(num_nodes)) | set(ei.reshape(-1).tolist())
        nodes = sorted(labels)
    else:
        raise ValueError(order)
    n = len(nodes)
    pos = {x: i for i, x in enumerate(nodes)}
    adj = [[] for _ in range(n)]
    if ei.size:
        for u, v in ei.T.tolist():
            if u != v:
                adj[pos[u]].append(pos[v])
                adj[pos[v]].append(pos[u])
    dist = np.zeros((n, n), dtype=np.float64)           # unreachable and diagonal stay 0
    for s in range(n):                                  # BFS per source == Floyd–Warshall on unit weights
        seen = np.full(n, -1, dtype=np.int64)
        seen[s] = 0
        frontier = [s]
        while frontier:
            nxt = []
            for a in frontier:
                for b in adj[a]:
                    if seen[b] < 0:
                        seen[b] = seen[a] + 1
                        nxt.append(b)
            frontier = nxt
        reach = seen > 0
        dist[s, reach] = seen[reach]
    return dist


def hop_distance_matrices_device(edge_indices, num_nodes, order: str = 'networkx', device=None):
    """Batched device version of `hop_distance_matrix` (kernel `ppk_hop_distances`): one launch for all
    bundles of a step.  `edge_indices`: sequence of [2, E_k] integer arrays / tensors with LOCAL labels;
    `num_nodes`: sequence of n_k (each <= 64).  Returns (g_flat float64 device tensor holding the n_k x n_k
    matrices back to back — the layout `graph_loss_segmented` takes —, sizes, row_order int64 device tensor:
    original label of every row).  Raises IndexError on labels outside [0, n_k) (checked on the host: no sync)."""
    from . import _C
    if order not in ('networkx', 'index'):
        raise ValueError(order)
    if device is None:
        device = torch.device('cuda')
    sizes = [int(n) for n in num_nodes]
    if len(sizes) != len(edge_indices) or not sizes:
        raise ValueError()
    host = hop_inputs_host(edge_indices, sizes)
    return hop_distances_launch(host, order, device)


def hop_inputs_host(edge_indices, num_nodes, pin=False):
    """Host half of `hop_distance_matrices_device`: label validation (IndexError, where the edge lists still are host
    data: reading the kernel's status word back would cost a device-to-host sync per step) and ONE flat int64 array
    [esrc | edst | edge_ptr | seg_ptr | mat_ptr] for a single host-to-device copy."""
    sizes = [int(n) for n in num_nodes]
    if len(sizes) != len(edge_indices) or not sizes:
        raise ValueError()
    eis = [np.asarray(e.cpu() if torch.is_tensor(e) else e, dtype=np.int64).reshape(2, -1) for e in edge_indices]
    for e, n in zip(eis, sizes):
        if e.size and (int(e.min()) < 0 or int(e.max()) >= n):
            raise IndexError('edge_index holds a node label outside [0, num_nodes)')
    edge_ptr = np.zeros(len(sizes) + 1, dtype=np.int64)
    edge_ptr[1:] = np.cumsum([e.shape[1] for e in eis])
    seg_ptr = np.zeros(len(sizes) + 1, dtype=np.int64)
    seg_ptr[1:] = np.cumsum(sizes)
    mat_ptr = np.zeros(len(sizes) + 1, dtype=np.int64)
    mat_ptr[1:] = np.cumsum([n * n for n in sizes])
    packed = torch.from_numpy(np.concatenate([np.concatenate(eis, axis=1).reshape(-1), edge_ptr, seg_ptr, mat_ptr]))
    if pin and torch.cuda.is_available():
        packed = packed.pin_memory()
    return dict(packed=packed, E=int(edge_ptr[-1]), K=len(sizes), sizes=sizes, n_mat=int(mat_ptr[-1]), n_rows=int(seg_ptr[-1]))


def hop_distances_launch(host, order='networkx', device=None, non_blocking=False):
    """Device half: one copy, one launch of `ppk_hop_distances`.  -> (g_flat, sizes, row_order)."""
    from . import _C
    if device is None:
        device = torch.device('cuda')
    dev = host['packed'].to(device, non_blocking=non_blocking)
    E, K, sizes = host['E'], host['K'], host['sizes']
    esrc, edst = dev[:E], dev[E:2 * E]
    o = 2 * E
    d_edge_ptr, d_seg_ptr, d_mat_ptr = dev[o:o + K + 1], dev[o + K + 1:o + 2 * K + 2], dev[o + 2 * K + 2:o + 3 * K + 3]
    g = torch.empty(host['n_mat'], dtype=torch.float64, device=device)
    row_order = torch.empty(host['n_rows'], dtype=torch.int64, device=device)
    status = torch.zeros(1, dtype=torch.int32, device=device)
    _C.check(_C.lib.ppk_hop_distances(_C.p(esrc) if E else None, _C.p(edst) if E else None, _C.p(d_edge_ptr), _C.p(d_seg_ptr),
                                      _C.p(d_mat_ptr), K, max(sizes), 1 if order == 'networkx' else 0,
                                      _C.p(g), _C.p(row_order), _C.p(status), _C.stream()))
    return g, sizes, row_order


def sample_bundle(neighbors, root, max_close_neighbors=5, max_total_bundle=20, rng=np.random):
    """Shortlist of edges for one root (utils.py:161-179).  `neighbors(x)` lists
    x's neighbours in adjacency insertion order; `rng` is the numpy global RNG
    module by default, like the reference."""
    close = rng.permutation(list(neighbors(root)))[:max_close_neighbors]
    total_far = sum(len(list(neighbors(x))) for x in close)
    shortlist = [[root, x] for x in close]
    counter = 0
    while len(shortlist) < max_total_bundle and counter < total_far:
        cur = close[counter % len(close)]
        far = rng.choice(list(neighbors(cur)))
        shortlist.append([cur, far])
        counter += 1
    return shortlist


def relabel_bundle(shortlist):
    """utils.py:183-194: undirected graph from the shortlist (duplicates merged),
    nodes relabelled by first appearance; returns (original_labels, edge_index [2, E])
    with each undirected edge once, ordered by node then adjacency insertion order
    (what `nx.to_pandas_edgelist(convert_node_labels_to_integers(...))` yields)."""
    order, pos, adj = [], {}, {}

    def add(x):
        if x not in pos:
            pos[x] = len(order)
            order.append(x)
            adj[x] = []

    for u, v in shortlist:
        u, v = int(u), int(v)
        add(u)
        add(v)
        if v not in adj[u]:
            adj[u].append(v)
            if u != v:
                adj[v].append(u)
    edges, seen = [], set()
    for u in order:
        for v in adj[u]:
            key = (min(pos[u], pos[v]), max(pos[u], pos[v]))
            if key in seen:
                continue
            seen.add(key)
            edges.append([pos[u], pos[v]])
    ei = np.asarray(edges, dtype=np.int64).reshape(-1, 2).T
    return order, ei


class Data:
    """Minimal stand-in for torch_geometric.data.Data as the reference uses it
    (`data.x` list of token tensors, `data.edge_index` [2, E], `data.num_nodes`)."""

    def __init__(self, x=None, edge_index=None, num_nodes=None, **kwargs):
        self.x = x
        self.edge_index = edge_index
        self.num_nodes = num_nodes if num_nodes is not None else (len(x) if x is not None else None)
        for k, v in kwargs.items():
            setattr(self, k, v)


def adjacency_lists(edgelist):
    """nx.from_edgelist(edgelist): adjacency lists in insertion order, duplicates merged (utils.py:112)."""
    adj = {}
    for u, v in edgelist:
        adj.setdefault(u, [])
        adj.setdefault(v, [])
        if v not in adj[u]:
            adj[u].append(v)
            if u != v:
                adj[v].append(u)
    return adj


def adjacency_csr(adj):
    """(rowptr, col) int64 numpy arrays of the adjacency lists of nodes 0 .. len(adj) - 1, insertion order kept."""
    n = len(adj)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum([len(adj[x]) for x in range(n)])
    col = np.fromiter((y for x in range(n) for y in adj[x]), dtype=np.int64, count=int(rowptr[-1]))
    return rowptr, col


def sample_bundles_device(rowptr, col, roots=None, max_close_neighbors: int = 5, max_total_bundle: int = 20, seed: int = 0,
                          device=None):
    """The bundles of `roots` (default: every node) sampled by ONE launch of `ppk_sample_bundles` (utils.py:155-196
    with one Philox stream per root in place of numpy's global stream).  `rowptr`, `col`: adjacency lists in insertion
    order (adjacency_csr).  Returns a dict of device tensors: shortlist [R, T, 2] / n_short [R] (original labels),
    labels [R, T + 1] / n_nodes [R], edges [R, T, 2] / n_edges [R] (local labels, the order of
    to_pandas_edgelist(convert_node_labels_to_integers(from_edgelist(shortlist))))."""
    from . import _C
    if device is None:
        device = torch.device('cuda')
    rp = torch.as_tensor(rowptr, dtype=torch.int64).to(device).contiguous()
    cl = torch.as_tensor(col, dtype=torch.int64).to(device).contiguous()
    N = rp.numel() - 1
    if roots is not None:
        roots = torch.as_tensor(roots, dtype=torch.int64)
        if roots.numel() and (int(roots.min()) < 0 or int(roots.max()) >= N):
            raise IndexError('root outside [0, num_nodes)')
        roots = roots.to(device).contiguous()
    R = N if roots is None else roots.numel()
    T = int(max_total_bundle)
    i64 = dict(dtype=torch.int64, device=device)
    i32 = dict(dtype=torch.int32, device=device)
    out = dict(shortlist=torch.zeros(R, T, 2, **i64), n_short=torch.zeros(R, **i32), labels=torch.zeros(R, T + 1, **i64),
               n_nodes=torch.zeros(R, **i32), edges=torch.zeros(R, T, 2, **i64), n_edges=torch.zeros(R, **i32))
    status = torch.zeros(1, **i32)
    _C.check(_C.lib.ppk_sample_bundles(_C.p(rp), _C.p(cl), N, _C.p(roots), R, int(max_close_neighbors), T, int(seed) & (2 ** 64 - 1),
                                       _C.p(out['shortlist']), _C.p(out['n_short']), _C.p(out['labels']), _C.p(out['n_nodes']),
                                       _C.p(out['edges']), _C.p(out['n_edges']), _C.p(status), _C.stream()))
    if int(status.item()) != 0:                    # start-up path: one read-back is affordable
        raise IndexError('adjacency lists hold a node id outside [0, num_nodes)')
    return out


def load_data(filename, smiles_language, max_close_neighbors: int = 5, max_total_bundle: int = 20, sampler: str = 'host',
              seed: int = 0) -> List[Data]:
    """Reference utils.py:101-198: one 2-hop bundle per node of the relation graph in
    the JSON file (`smiles2nodes`, `edgelist`).  sampler='host': sampled with the global numpy RNG, statement for
    statement like the reference; sampler='device': all bundles in one kernel launch (`sample_bundles_device`, one Philox
    stream per root keyed by `seed`: same distribution, other numbers than numpy's).
    `smiles_language` needs `smiles_to_token_indexes(str) -> list[int]`."""
    import json
    with open(filename, 'r') as f:
        raw = json.load(f)
    nodes2smiles = {v: k for k, v in raw['smiles2nodes'].items()}
    adj = adjacency_lists(raw['edgelist'])            # nx.from_edgelist: adjacency in insertion order, duplicates merged
    data = []
    if sampler == 'device':
        rowptr, col = adjacency_csr(adj)
        out = {k: v.cpu().numpy() for k, v in sample_bundles_device(rowptr, col, None, max_close_neighbors, max_total_bundle, seed).items()}
        for root in range(len(adj)):
            nn, ne = int(out['n_nodes'][root]), int(out['n_edges'][root])
            labels = out['labels'][root, :nn].tolist()
            x = [torch.tensor(smiles_language.smiles_to_token_indexes(nodes2smiles[l])) for l in labels]
            edge_index = np.ascontiguousarray(out['edges'][root, :ne].T)
            data.append(Data(x=x, edge_index=torch.tensor(edge_index), num_nodes=len(x)))
        return data
    if sampler != 'host':
        raise ValueError(sampler)
    for root in range(len(adj)):
        shortlist = sample_bundle(lambda x: adj[x], root, max_close_neighbors, max_total_bundle)
        labels, edge_index = relabel_bundle(shortlist)
        x = [torch.tensor(smiles_language.smiles_to_token_indexes(nodes2smiles[l])) for l in labels]
        data.append(Data(x=x, edge_index=torch.tensor(edge_index), num_nodes=len(x)))
    return data
