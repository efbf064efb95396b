"""bench.py's N > 1 path (rendezvous, per-rank shards, token-weighted loss, flat-gradient all-reduce, barrier +
max-over-ranks timing, one JSON line from rank 0) rehearsed with two ranks on the one GPU of the test box
(PPK_BENCH_REHEARSAL=1: gloo, ranks share the card).  The numbers mean nothing; the control flow is the
driver's `python -m torch.distributed.run ... bench.py --gpus N` launch."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('dtype,port', [('f32', '29561'), ('bf16', '29563')])
def test_two_rank_bench_line(dtype, port):
    """dtype bf16 = BASELINE configs[4]'s arithmetic (ppk_mfma_mode(1) on every rank): the launch the driver would use
    with `--gpus 8 --dtype bf16`, at two ranks."""
    env = dict(os.environ, PPK_BENCH_REHEARSAL='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', port, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--dtype', dtype]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                                   # rank 0 only
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 2 and rec['scaling'] == 'weak'
    assert rec['config']['global_batch'] == 4096 and rec['config']['parallelism'] == 'dp2'
    assert rec['value'] > 0 and abs(rec['value'] - 4096 / (rec['ms_per_step'] * 1e-3)) <= 1e-2 * rec['value']
    assert 'cpu_baseline' not in rec and 'rehearsal' in rec
    assert rec['dtype'] == dtype


@pytest.mark.parametrize('exchange,port', [('halo', '29565'), ('allgather', '29567')])
def test_two_rank_configs3_harness(exchange, port):
    """tools/bench_graph_dist.py (BASELINE configs[3]: node-sharded sparse graph loss, latents exchanged between ranks) with
    two ranks sharing the test box's GPU over gloo, reduced graph (--small): the launch line, the per-phase
    timing, the wire-byte accounting and the loss (equal to the single-process loss on the same graph) — the
    line must flag itself as a rehearsal, never a measurement."""
    env = dict(os.environ, PPK_BENCH_REHEARSAL='1')
    script = os.path.join(ROOT, 'tools', 'bench_graph_dist.py')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', port, script, '--gpus', '2', '--steps', '2', '--warmup', '1', '--small', '--exchange', exchange]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['exchange'] == exchange and rec['backend'] == 'gloo'
    assert rec['is_measurement'] is False and 'rehearsal' in rec and rec['ranks_per_gpu'] == 2
    assert set(rec['phases_ms']) == {'exchange_fwd', 'kernels_fwd', 'kernels_bwd', 'exchange_bwd', 'step'}
    assert rec['wire']['bytes_per_gpu_per_step'] > 0 and rec['roofline']['algorithmic_bytes'] > 0
    single = subprocess.run([sys.executable, script, '--gpus', '1', '--steps', '1', '--warmup', '0', '--small'], cwd=ROOT,
                            capture_output=True, text=True, timeout=280)
    assert single.returncode == 0, single.stdout[-2000:] + single.stderr[-2000:]
    one = json.loads([l for l in single.stdout.splitlines() if l.startswith('{')][0])
    assert one['is_measurement'] is True and one['n_gpus'] == 1
    assert abs(rec['loss'] - one['loss']) <= 1e-5 * abs(one['loss']), (rec['loss'], one['loss'])
