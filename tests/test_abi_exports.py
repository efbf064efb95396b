"""CPU-side check that libppk.so loads and exports every symbol declared in
include/ppk.h (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'ppk.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ppk_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert 'ppk_graphreg_dense_fwd' in syms and 'ppk_version' in syms


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, 'paccmann_polymer_amd', 'libppk.so'))
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_binding_table_matches_header():
    from paccmann_polymer_amd import _C
    assert sorted(_C.SIGNATURES) == declared_symbols()
    assert _C.lib.ppk_version().startswith(b'ppk')


def test_no_cpu_fallback():
    import pytest
    import torch
    import paccmann_polymer_amd as P
    with pytest.raises(P._C.PPKError):
        P.graph_loss(torch.zeros(4, 3), torch.zeros(4, 4))


def test_shape_mismatch_is_bare_valueerror():
    import pytest
    import torch
    import paccmann_polymer_amd as P
    with pytest.raises(ValueError):
        P.graph_loss(torch.zeros(4, 3), torch.zeros(5, 5))
