"""Multi-rank harness for BASELINE.json configs[3]: the sparse graph-regularisation loss (K5) on a 1 M-node / 10 M-edge
relation graph, nodes sharded by range over N GPUs, latents exchanged over xGMI (RCCL), anchors evaluated in
work-balanced ranges, MAX / SUM all-reduces of the normaliser and the sum, partial gradients sent back.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_graph_dist.py --gpus N [--exchange halo|allgather] [--steps K] [--warmup W] [--small]
    python tools/bench_graph_dist.py --gpus 1            (single process: the whole graph on one GPU)

One process per GPU (backend nccl = RCCL).  Rank 0 prints ONE JSON line: per-phase times (max over ranks) of
exchange-forward | kernels-forward (incl. the two scalar all-reduces) | kernels-backward | exchange-backward, the
algorithmic bytes of the slowest rank's kernels, and the bytes each GPU puts on / takes off the wire against
7 x 153 GB/s of xGMI per GPU.

NOTHING this prints is a measurement of configs[3] unless `backend` is "nccl" and `ranks_per_gpu` is 1: with
PPK_BENCH_REHEARSAL=1 the ranks share one GPU over gloo (RCCL refuses two ranks on one device) — that run only
rehearses the control flow (tests/test_gpu_bench_rehearsal.py) and says so in the line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
XGMI_GBS_PER_GPU = 7 * 153.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--exchange', choices=['halo', 'allgather'], default='halo')
    ap.add_argument('--small', action='store_true')
    args = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    rehearsal = os.environ.get('PPK_BENCH_REHEARSAL') == '1'
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    if rehearsal:
        import faulthandler
        faulthandler.dump_traceback_later(150, exit=True)
    device = torch.device('cuda', 0 if rehearsal else local)
    torch.cuda.set_device(device)
    backend = None
    if world > 1:
        backend = 'gloo' if rehearsal else 'nccl'
        dist.init_process_group(backend, rank=rank, world_size=world)

    import bench_graph as BG
    from paccmann_polymer_amd.graph_loss_sparse import (SparseRelationGraph, _AllGatherRows, _HaloRows, _SparseGraphLoss,
                                                        node_ranges)
    rng = np.random.default_rng(1)                                  # the same graph on every rank (it is static: built once)
    N = 100_000 if args.small else 1_000_000
    Z = 256
    und = BG.relation_graph(rng, N, 10)
    E = int(und.shape[1])
    attr = torch.tensor(rng.integers(1, 4, size=E).astype(np.float32)).to(device)
    graph = SparseRelationGraph(torch.tensor(und).to(device), attr, N)
    starts, counts = node_ranges(N, world)
    astarts = graph.anchor_ranges(world)
    a0, a1 = astarts[rank], astarts[rank + 1]
    g = torch.Generator(device='cpu').manual_seed(7)
    z_all = torch.randn(N, Z, generator=g)
    z_local = z_all[starts[rank]:starts[rank + 1]].to(device).requires_grad_(True)
    del z_all
    group = None if world == 1 else dist.group.WORLD

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    phases = ('exchange_fwd', 'kernels_fwd', 'kernels_bwd', 'exchange_bwd')
    acc = {k: 0.0 for k in phases}
    total = 0.0
    loss_val = None
    for it in range(args.warmup + args.steps):
        z_local.grad = None
        sync()
        t = [time.perf_counter()]
        if world == 1:
            z_full = z_local
        elif args.exchange == 'halo':
            z_full = _HaloRows.apply(z_local, graph, group)
        else:
            z_full = _AllGatherRows.apply(z_local, counts, group)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        zf = z_full.detach().requires_grad_(True)
        loss = _SparseGraphLoss.apply(zf, graph, a0, a1, group if world > 1 else False)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        (dzf,) = torch.autograd.grad(loss, zf)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        if world == 1:
            z_local.grad = dzf
        else:
            z_full.backward(dzf)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        if it >= args.warmup:
            for k, (x0, x1) in zip(phases, zip(t[:-1], t[1:])):
                acc[k] += x1 - x0
            total += t[-1] - t[0]
        loss_val = float(loss.detach())
    times = torch.tensor([acc[k] / args.steps for k in phases] + [total / args.steps], dtype=torch.float64)
    if world > 1:
        tt = times.to('cpu' if rehearsal else device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        times = tt.cpu()

    # bytes: SURVEY 8(d) per undirected edge (two latent rows, the int64 pair, the attribute; d / g / coefficient scratch), by the
    # rank's share of the CSR entries; wire: rows received before the kernels and gradient rows sent afterwards (and vice versa)
    rp = graph.rowptr
    frac = float(rp[a1] - rp[a0]) / float(2 * E)
    by_f = E * frac * (2 * 4 * Z + 24) + 3 * E * frac * 4
    by_fb = 2 * by_f + N * Z * 4 / world
    if world == 1:
        wire_in = wire_out = 0
    elif args.exchange == 'halo':
        plan = graph.halo_plan(world)
        wire_in = sum(int(plan[rank][q].numel()) for q in range(world) if q != rank) * Z * 4
        wire_out = sum(int(plan[q][rank].numel()) for q in range(world) if q != rank) * Z * 4
    else:
        wire_in = (N - counts[rank]) * Z * 4
        wire_out = counts[rank] * Z * 4 * (world - 1)
    wires = torch.tensor([by_fb, wire_in + wire_out], dtype=torch.float64)          # fwd: in + out; bwd: the same rows the other way
    if world > 1:
        ww = wires.to('cpu' if rehearsal else device)
        dist.all_reduce(ww, op=dist.ReduceOp.MAX)
        wires = ww.cpu()
    if rank == 0:
        ms = {k: round(float(v) * 1e3, 3) for k, v in zip(phases + ('step',), times)}
        kern_s = float(times[1] + times[2])
        exch_s = float(times[0] + times[3])
        wire_step = 2.0 * float(wires[1])
        rec = dict(
            metric='sparse graph-reg loss forward + backward, node-sharded (BASELINE configs[3])', n_gpus=world, steps=args.steps, warmup=args.warmup,
            ms_per_step=ms['step'], phases_ms=ms, exchange=args.exchange if world > 1 else 'none', backend=backend or 'single process',
            ranks_per_gpu=world if rehearsal else 1,
            workload=f'N={N} nodes, E={E} undirected power-law edges, Z={Z}, anchors in work-balanced ranges',
            edges_per_s=round(E / float(times[4]), 1),
            roofline=dict(bound='hbm', kernel='K5 (slowest rank: dist + terms + finish, coef + dz + dz_fix)', algorithmic_bytes=int(wires[0]),
                          achieved=round(float(wires[0]) / kern_s / 1e9, 1), peak=8000.0, unit='GB/s', frac=round(float(wires[0]) / kern_s / 8e12, 4)),
            wire=dict(bytes_per_gpu_per_step=int(wire_step), xgmi_peak_GBps_per_gpu=XGMI_GBS_PER_GPU,
                      achieved_GBps=round(wire_step / exch_s / 1e9, 1) if exch_s > 0 and world > 1 else None,
                      ideal_ms_at_peak=round(wire_step / (XGMI_GBS_PER_GPU * 1e9) * 1e3, 3)),
            loss=loss_val, data='synthetic')
        rec['is_measurement'] = bool(world == 1 or (backend == 'nccl' and not rehearsal))
        if rehearsal:
            rec['rehearsal'] = 'ranks share ONE GPU over gloo: control flow only, no number here measures configs[3]'
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
