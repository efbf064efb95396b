"""Spawns 2 ranks (gloo, both on cuda:0) that shard the nodes of one relation
graph; checks loss and gradient against the single-rank result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker(rank, world, q):
    try:
        _worker(rank, world, q)
    except Exception as exc:  # report instead of leaving the parent waiting
        q.put((rank, False, repr(exc)))
        raise


def _worker(rank, world, q):
    os.environ['RANK'], os.environ['WORLD_SIZE'] = str(rank), str(world)
    print('rank', rank, 'init', flush=True)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    print('rank', rank, 'pg ok', flush=True)
    torch.cuda.set_device(0)
    from paccmann_polymer_amd.graph_loss_sparse import SparseRelationGraph, graph_loss_sparse, node_ranges
    import networkx as nx
    n, e, zd = 301, 2000, 32
    rng = np.random.default_rng(3)
    gr = nx.gnm_random_graph(n, e, seed=3)
    ei = torch.tensor(np.asarray(list(gr.edges()), dtype=np.int64).T).cuda()
    attr = torch.tensor(rng.integers(1, 4, size=ei.shape[1]).astype(np.float32)).cuda()
    z = torch.tensor((rng.standard_normal((n, zd)) * 0.3).astype(np.float32)).cuda()
    g = SparseRelationGraph(ei, attr, n)
    zs = z.clone().requires_grad_(True)
    single = graph_loss_sparse(zs, g)
    single.backward()
    print('rank', rank, 'single ok', float(single), flush=True)
    starts, counts = node_ranges(n, world)
    ok_loss, err, grads = True, 0.0, {}
    for exchange in ('halo', 'allgather'):
        zl = z[starts[rank]:starts[rank + 1]].clone().requires_grad_(True)
        loss = graph_loss_sparse(zl, g, shard=True, exchange=exchange)
        print('rank', rank, exchange, 'fwd ok', flush=True)
        loss.backward()
        print('rank', rank, exchange, 'bwd ok', flush=True)
        ok_loss = ok_loss and abs(float(loss) - float(single)) <= 1e-6 * abs(float(single))
        ref = zs.grad[starts[rank]:starts[rank + 1]]
        err = max(err, float((zl.grad - ref).norm() / ref.norm()))
        grads[exchange] = zl.grad.clone()
    # the halo moves fewer rows than the all-gather whenever some row is not touched by the other rank's anchors
    plan = g.halo_plan(world)
    moved = sum(int(plan[rank][qq].numel()) for qq in range(world) if qq != rank)
    assert moved <= n - counts[rank], (moved, n, counts)
    ok_loss = ok_loss and torch.allclose(grads['halo'], grads['allgather'], rtol=1e-5, atol=1e-7)
    q.put((rank, ok_loss, err))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    mp.set_start_method('spawn')
    q = mp.Queue()
    ps = [mp.Process(target=worker, args=(r, 2, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=200) for _ in ps]
    for p in ps:
        p.join(60)
    assert all(r[1] for r in res), res
    assert all(isinstance(r[2], float) and r[2] <= 1e-5 for r in res), res
    print('SPARSE_DP_OK', res)
