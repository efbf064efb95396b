"""Benchmark of the graph-regularised SMILES-VAE training step (BASELINE.json
configs[1]): batch 2048 per GPU (128 bundles x 16 SMILES, len 20..100), fp32,
StackGRU VAE (vocab 64, E 128, H 512, L 2, stack 50x50, Z 256), graph_gamma 1,
Adam lr 1e-4.  One "step" = encoder + reparameterise + decoder + graph loss,
forward and backward, + optimizer, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
(N > 1: launched by torch.distributed.run, one rank per GPU over RCCL; weak scaling.)
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import faulthandler
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
from torch.nn.utils.rnn import PackedSequence, pack_sequence

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PARAMS = dict(vocab_size=64, embedding_size=128, rnn_cell_size=512, n_layers=2, stack_width=50, stack_depth=50,
              latent_dim=256, dropout=0.0, bidirectional=False, use_stack=True, batch_mode='packed',
              padding_index=0, batch_size=1)
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA (the figure without 2:1 sparsity)
PEAK_HBM_GBS = 8000.0


def synth_batch(rng, n_bundles, bundle_nodes, lo=20, hi=100, vocab=64, return_edges=False):
    """Synthetic SMILES token lists + per-bundle relation graphs (random tree + a few chords, <= 20 edges)."""
    from paccmann_polymer_amd.data_prep import hop_distance_matrix
    seqs, gs, sizes, eis = [], [], [], []
    for _ in range(n_bundles):
        n = bundle_nodes
        edges = [[int(rng.integers(0, v)), v] for v in range(1, n)]
        for _ in range(min(4, max(0, 20 - (n - 1)))):
            a, b = rng.integers(0, n, size=2)
            if a != b:
                edges.append([int(a), int(b)])
        ei = np.asarray(edges, dtype=np.int64).T
        gs.append(hop_distance_matrix(ei, n, order='networkx'))
        eis.append(ei)
        sizes.append(n)
        for _ in range(n):
            L = int(rng.integers(lo, hi + 1))
            seqs.append(torch.tensor(np.concatenate([[2], rng.integers(4, vocab, size=L - 2), [3]]), dtype=torch.int64))
    if return_edges:
        return seqs, gs, sizes, eis
    return seqs, gs, sizes


def to_device_batch(seqs, gs, device):
    enc = pack_sequence(seqs, enforce_sorted=False)
    tgt = pack_sequence([torch.cat([s[1:], torch.zeros(1, dtype=torch.int64)]) for s in seqs], enforce_sorted=False)

    def dev(p):
        return PackedSequence(p.data.to(device), p.batch_sizes, p.sorted_indices.to(device), p.unsorted_indices.to(device))
    g_flat = torch.tensor(np.concatenate([g.ravel() for g in gs]), dtype=torch.float64, device=device)
    return dev(enc), dev(enc), dev(tgt), g_flat


def cpu_baseline(seed=0, bundle_nodes=16):
    """Reference-semantics CPU step on the host cores at BASELINE configs[0] (batch 64, fp32, Adam lr 1e-4), split as
    BASELINE.md section 3 asks: the sequence part (the oracle's per-timestep nn.GRU StackGRU VAE: forward, backward,
    Adam) timed once, and the graph loss (forward + backward w.r.t. z) timed in three restatements on two graphs:
      python loop  — term by term with torch autograd, as the reference's graph_loss does it (oracle.graph_loss_loop_torch)
      vectorised   — closed form in a few torch ops with autograd (oracle.graph_loss_vectorised_torch)
      C port       — oracle/graph_loss_ref.c with its analytic gradient
      1a: one dense 64-node graph (Floyd–Warshall hops of a G(64, 192) random graph); 1b: 4 bundles x 16 nodes.
    `value` is the reference as written on how it batches its graphs (python loop, 1b); the others are in `variants`.
    A bounded sample: one pass of each leg (the 1a python loop alone is ~20 s)."""
    from oracle import stackgru_ref as R
    from oracle import graph_loss_oracle as O
    from paccmann_polymer_amd.data_prep import hop_distance_matrix
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    n_bundles = 64 // bundle_nodes
    seqs, gs, sizes = synth_batch(rng, n_bundles, bundle_nodes)
    B = len(seqs)
    # 1a: G(n = 64, m = 192) random graph, re-drawn until connected
    while True:
        pairs = set()
        while len(pairs) < 192:
            u, v = (int(x) for x in rng.integers(0, B, size=2))
            if u != v:
                pairs.add((min(u, v), max(u, v)))
        g_dense = hop_distance_matrix(np.asarray(sorted(pairs), dtype=np.int64).T, B, order='networkx')
        if np.isfinite(g_dense).all():
            break
    vae = R.TeacherVAERef(R.StackGRUEncoderRef(PARAMS), R.StackGRUDecoderRef(PARAMS))
    opt = torch.optim.Adam(vae.parameters(), lr=1e-4)
    enc = pack_sequence(seqs, enforce_sorted=False)
    tgt = pack_sequence([torch.cat([s[1:], torch.zeros(1, dtype=torch.int64)]) for s in seqs], enforce_sorted=False)
    eps = torch.randn(B, PARAMS['latent_dim'])
    dz_inject = torch.zeros(B, PARAMS['latent_dim'])
    zs = []

    def seq_step():
        opt.zero_grad()
        dec_loss, mu, logvar, z = R.train_step_ref(vae, enc, enc, tgt, eps)
        z2 = z.squeeze(0)
        zs.append(z2.detach().clone())
        (dec_loss + (z2 * dz_inject).sum()).backward()          # graph_gamma = 1: the graph loss enters through dL/dz
        opt.step()
    seq_step()                                                   # warm-up
    t0 = time.perf_counter()
    seq_step()
    seq_step()
    t_seq = (time.perf_counter() - t0) / 2
    z0 = zs[-1]

    def timed(fn):
        t0 = time.perf_counter()
        fn()
        return time.perf_counter() - t0

    def torch_leg(fn, graphs, seg):
        def run():
            z = z0.clone().requires_grad_(True)
            off, total = 0, 0.0
            for g, n in zip(graphs, seg):
                total = total + fn(z[off:off + n], torch.tensor(g, dtype=torch.float32))
                off += n
            total.backward()
        return run

    def c_leg(graphs, seg):
        return lambda: O.graph_loss_segmented_c(z0.numpy(), graphs, seg, acc_fp32=True)
    legs = {
        'python_loop_1b': torch_leg(O.graph_loss_loop_torch, gs, sizes),
        'python_loop_1a': torch_leg(O.graph_loss_loop_torch, [g_dense], [B]),
        'vectorised_1b': torch_leg(O.graph_loss_vectorised_torch, gs, sizes),
        'vectorised_1a': torch_leg(O.graph_loss_vectorised_torch, [g_dense], [B]),
        'c_port_1b': c_leg(gs, sizes),
        'c_port_1a': c_leg([g_dense], [B]),
    }
    variants = {}
    for name, fn in legs.items():
        if not name.startswith('python_loop'):
            fn()                                                 # warm-up of the cheap legs
        t = timed(fn)
        variants[name] = dict(samples_per_s=round(B / (t_seq + t), 3), graph_loss_s=round(t, 4), step_s=round(t_seq + t, 4))
    cores = torch.get_num_threads()
    return dict(value=variants['python_loop_1b']['samples_per_s'], unit='samples/s', cores=cores, kind='port',
                sample=f'configs[0]: batch {B}, len 20..100, one pass per leg; step = sequence part (oracle nn.GRU-per-timestep '
                       f'StackGRU VAE fwd+bwd+Adam, {t_seq:.2f} s, mean of 2) + graph loss fwd+bwd; value = python-loop '
                       f'restatement on 1b ({n_bundles} bundles x {bundle_nodes}); torch threads {cores} (the python loop itself is single-threaded)',
                sequence_part_s=round(t_seq, 4), variants=variants)


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN_bench_pmc_v*.txt, written by tools/collect_profiles.sh: FETCH_SIZE and WRITE_SIZE in KiB summed over
    the run's launches; on gfx950 FETCH_SIZE counts half of the wide streaming reads, so it is doubled —
    MI355X_MICROARCH.md, HBM section).  The counters cannot be collected inside the timed run; the newest committed pass
    is used ONLY when its `# source_hash` line equals the hash of the kernel sources of this tree
    (tools/source_hash.py) — a profile of other code is refused and `traffic` stays null."""
    import glob
    import re
    from tools.source_hash import source_hash
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r*_bench_pmc_v*.txt')),
                   key=lambda f: [int(x) for x in re.findall(r'\d+', os.path.basename(f))])
    if not files:
        return {}
    short = kernel.split('<')[0]
    vals, section, file_hash = {}, None, None
    for line in open(files[-1]):
        f = line.split()
        if len(f) == 3 and f[0] == '#' and f[1] == 'source_hash':
            file_hash = f[2]
        elif len(f) >= 3 and f[0] == 'kernel' and f[-1] in ('FETCH_SIZE', 'WRITE_SIZE'):
            section = f[-1]
        elif line.startswith('kernel'):
            section = None
        elif section and len(f) == 3 and f[0] == short:
            vals[section] = (float(f[2]), int(f[1]))
    here = source_hash()
    if file_hash != here:
        return dict(traffic_note=f'{os.path.basename(files[-1])} was taken from sources {file_hash}, this tree is {here}: refused')
    if 'FETCH_SIZE' not in vals or 'WRITE_SIZE' not in vals:
        return {}
    (fk, fl), (wk, wl) = vals['FETCH_SIZE'], vals['WRITE_SIZE']
    return dict(traffic=round(2.0 * fk * 1024 / fl + wk * 1024 / wl), traffic_unit='HBM bytes per launch',
                traffic_source=os.path.basename(files[-1]), traffic_source_hash=file_hash)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--bundles', type=int, default=128)
    ap.add_argument('--bundle-nodes', type=int, default=16)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dtype', choices=['f32', 'bf16'], default='f32',
                    help='bf16: the MFMA products of the persistent sequence kernels take bf16 operands (ppk_mfma_mode(1), configs[4] variant)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` by hand: start the N ranks as a CHILD job (torch.distributed.run) before this
        # process touches the GPU, and leave with its exit code
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        sys.exit(subprocess.call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
                                  '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # PPK_BENCH_REHEARSAL=1: run the N > 1 code path on a box with fewer GPUs than ranks (ranks share the cards,
    # gloo instead of RCCL, which refuses two ranks on one GPU).  A rehearsal of the control flow, never a
    # measurement: the JSON line says so.
    rehearsal = os.environ.get('PPK_BENCH_REHEARSAL') == '1'
    if rehearsal:
        faulthandler.dump_traceback_later(90, exit=True)       # a rehearsal that hangs says where
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)

    import paccmann_polymer_amd as P
    from paccmann_polymer_amd.paccmann_encoder import Encoder
    from paccmann_polymer_amd.optim import FlatAdam
    from paccmann_polymer_amd.distributed import GradSync, global_token_count, dp_train_step
    from paccmann_polymer_amd.graph_constrained_loss import graph_loss_segmented

    if P._C.lib.ppk_mfma_mode(1 if args.dtype == 'bf16' else 0) != (1 if args.dtype == 'bf16' else 0):
        sys.exit('bench.py: ppk_mfma_mode did not take the requested mode')
    torch.manual_seed(0)                                  # identical weights on every rank
    model = Encoder(PARAMS, None, None, batch_size=1, batch_mode='Packed')
    model.gru_vae.train()
    opt = FlatAdam(model.gru_vae.parameters(), lr=1e-4)
    rng = np.random.default_rng(1234 + rank)              # a different shard per rank (weak scaling)
    seqs, gs, sizes = synth_batch(rng, args.bundles, args.bundle_nodes)
    enc, dec, tgt, g_flat = to_device_batch(seqs, gs, device)
    B = len(seqs)
    model.update_batch_size(B)
    eps = torch.randn(B, PARAMS['latent_dim'], device=device)
    n_tokens = int(enc.data.numel())
    local_valid = n_tokens - B
    total_valid = global_token_count(local_valid)
    sync = GradSync(opt.flat_grad)

    def fwd():
        dec_loss, mu, logvar, z = model.train_step(enc, dec, tgt, eps)
        return dec_loss, graph_loss_segmented(z.squeeze(0), g_flat, sizes)

    def step():
        return dp_train_step(fwd, opt, local_valid, total_valid, 1.0, sync)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, dl, gl = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device='cpu' if rehearsal else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3

    # ---- roofline of the dominant kernel: one extra step with per-launch HIP events on the launch stream
    lib = P._C.lib
    nk = lib.ppk_prof_kind_count()
    lib.ppk_prof_enable(1)
    step()
    torch.cuda.synchronize()
    lib.ppk_prof_enable(0)
    ms = (ctypes.c_double * nk)()
    work = (ctypes.c_double * nk)()
    launches = (ctypes.c_int64 * nk)()
    lib.ppk_prof_collect(ms, work, launches)
    kinds = [dict(kernel=lib.ppk_prof_kind_name(k).decode(), ms=ms[k], work=work[k], launches=int(launches[k]))
             for k in range(nk) if launches[k]]
    kinds.sort(key=lambda d: -d['ms'])
    top = kinds[0]
    is_gemm = top['kernel'].startswith(('k_gemm', 'k_nt', 'k_tn', 'k_seq'))        # fp32-MFMA GEMM families: work = FLOPs
    achieved = top['work'] / (top['ms'] * 1e-3) / (1e12 if is_gemm else 1e9)
    seq_bf16 = args.dtype == 'bf16' and top['kernel'].startswith('k_seq')
    peak = (PEAK_BF16_MFMA_TFLOPS if seq_bf16 else PEAK_FP32_MFMA_TFLOPS) if is_gemm else PEAK_HBM_GBS
    roofline = dict(bound='mfma' if is_gemm else 'hbm', achieved=round(achieved, 2), peak=peak,
                    unit='TFLOP/s' if is_gemm else 'GB/s', frac=round(achieved / peak, 4), traffic=None,
                    kernel=top['kernel'], launches_per_step=top['launches'],
                    profiled_launches_per_step=sum(d['launches'] for d in kinds),
                    avg_launch_us=round(top['ms'] * 1e3 / top['launches'], 2),
                    step_breakdown_ms={d['kernel']: round(d['ms'], 3) for d in kinds})

    if args.bundles == 128 and args.bundle_nodes == 16:
        if args.dtype == 'f32':
            roofline.update(pmc_traffic(top['kernel']))
        else:
            roofline['traffic_note'] = 'the committed PMC passes were taken in fp32 mode: no traffic figure for the bf16-operand variant'


    out = dict(metric='training samples/sec (whole node)', value=round(B * world / (dt / args.steps), 1),
               unit='samples/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=round(ms_per_step, 3), higher_is_better=True, scaling='weak', vs_baseline=None,
               dtype=args.dtype, data='synthetic',
               config=dict(workload=f'configs[1]: graph-regularised SMILES-VAE step, batch {B}/GPU = {args.bundles} bundles x '
                                    f'{args.bundle_nodes} SMILES (len 20..100), StackGRU VAE V64 E128 H512 L2 stack50x50 Z256, '
                                    f'graph_gamma 1, Adam lr 1e-4, ' + ('fp32' if args.dtype == 'f32' else 'bf16 operands / fp32 accumulation in the sequence-kernel MFMA products, fp32 elsewhere (configs[4] variant)'), global_batch=B * world,
                           tokens_per_gpu=n_tokens, parallelism=f'dp{world}'),
               roofline=roofline, final_loss=float(loss))
    if rehearsal:
        faulthandler.cancel_dump_traceback_later()
        out['rehearsal'] = 'ranks share GPUs over gloo: not a measurement'
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
