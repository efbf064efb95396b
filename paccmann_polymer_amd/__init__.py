"""MI355X-native training hot path of PaccMann/paccmann_polymer.

Host side mirrors the reference's Python API for the path;
all arithmetic runs in hand-written gfx950 HIP kernels behind the C-ABI of
`libppk.so` (include/ppk.h).
"""
from . import _C  # noqa: F401  (fails loudly if libppk.so is missing)
from .graph_constrained_loss import (  # noqa: F401
    graph_loss, graph_loss_segmented, compute_graph_dists, graph_regularizer_function,
)

__version__ = '0.1.0'
