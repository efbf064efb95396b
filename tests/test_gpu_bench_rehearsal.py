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
