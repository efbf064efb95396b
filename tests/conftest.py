import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(params=['per_step_launches', 'persistent_launch', 'weights_stationary', 'weights_stationary_looped'])
def seq_schedule(request):
    """Runs a test under every launch schedule of the K1 sequence entry points (ppk_seq_persist_mode 0 / 1 / 2): auto
    mode picks the row-block persistent launch only where the batch fills the machine (H 512: from batch 1024 up), so the parity cases (small batches) would otherwise
    never reach it.  Mode 1 falls back to the per-step launches where a shape is unsupported; mode 2 takes the small-batch
    weights-stationary kernels (ppk_seq_ws.hip) where the shape is supported and behaves like auto elsewhere; mode 3 is mode 2 with the
    looped forward (replicas of up to 256 rows) for batches beyond 32 sequences."""
    from paccmann_polymer_amd import _C
    mode = {'per_step_launches': 0, 'persistent_launch': 1, 'weights_stationary': 2, 'weights_stationary_looped': 3}[request.param]
    assert _C.lib.ppk_seq_persist_mode(mode) == mode
    yield request.param
    _C.lib.ppk_seq_persist_mode(-1)
