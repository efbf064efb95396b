"""ctypes binding of libppk.so (include/ppk.h).

There is no CPU fallback: every product op goes through this library, and the
import fails loudly when the library has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or
`make -C paccmann_polymer_amd/csrc`).
"""
import ctypes
import os

# torch must be imported BEFORE libppk.so is loaded: torch bundles its own HIP
# runtime (torch/lib/libamdhip64.so) and both libraries must bind to that one
# copy, otherwise device pointers and streams are not shared.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PPK_LIB_PATH') or os.path.join(_HERE, 'libppk.so')      # PPK_LIB_PATH: instrumented builds (tools/)

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_sz = ctypes.c_size_t
c_f32 = ctypes.c_float
c_f64 = ctypes.c_double
ptr = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/ppk.h declaration by declaration.
SIGNATURES = {
    'ppk_version': (ctypes.c_char_p, []),
    'ppk_last_error': (ctypes.c_char_p, []),
    'ppk_device_count': (c_int, []),
    'ppk_prof_enable': (c_int, [c_int]),
    'ppk_prof_kind_count': (c_int, []),
    'ppk_prof_kind_name': (ctypes.c_char_p, [c_int]),
    'ppk_prof_collect': (c_int, [ptr, ptr, ptr]),
    'ppk_graphreg_dense_workspace_bytes': (c_sz, [c_i64, c_i64, c_i64]),
    'ppk_graphreg_dense_fwd': (c_int, [ptr, c_i64, ptr, c_int, ptr, ptr,
                                       c_i64, c_i64, c_i64, c_i64, c_i64,
                                       ptr, ptr, ptr, c_sz, ptr]),
    'ppk_graphreg_dense_bwd': (c_int, [ptr, c_i64, ptr, ptr,
                                       c_i64, c_i64, c_i64, c_i64, c_i64,
                                       ptr, ptr, c_i64, ptr, c_sz, ptr]),
    'ppk_gemm_f32_workspace_bytes': (c_sz, [c_i64, c_i64, c_i64]),
    'ppk_gemm_f32': (c_int, [c_int, c_int, c_i64, c_i64, c_i64, c_f32, ptr, c_i64, ptr, c_i64, c_f32,
                             ptr, c_i64, ptr, ptr, c_sz, ptr]),
    'ppk_gemm_tn_colsum_f32': (c_int, [c_i64, c_i64, c_i64, ptr, c_i64, ptr, c_i64, ptr, c_i64, ptr, ptr, c_sz, ptr]),
    'ppk_nt_small_grid_limit': (c_i64, [c_i64]),
    'ppk_mfma_mode': (c_int, [c_int]),
    'ppk_seq_persist_mode': (c_int, [c_int]),
    'ppk_sample_bundles': (c_int, [ptr, ptr, c_i64, ptr, c_i64, c_int, c_int, ctypes.c_uint64, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr]),
    'ppk_hop_distances': (c_int, [ptr, ptr, ptr, ptr, ptr, c_i64, c_i64, c_int, ptr, ptr, ptr, ptr]),
    'ppk_stackgru_seq_fwd_workspace_bytes': (c_sz, [c_i64, c_i64, c_i64, c_i64]),
    'ppk_stackgru_seq_fwd': (c_int, [c_i64, ptr, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                                     ptr, ptr, ptr, c_i64, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr,
                                     ptr, ptr, ptr, ptr, ptr, ptr, c_i64, ptr, ptr, ptr, c_sz, ptr]),
    'ppk_dropout_factors': (c_int, [ctypes.c_uint64, c_f32, c_i64, c_i64, c_i64, c_i64, ptr, ptr]),
    'ppk_stackgru_seq_bwd_workspace_bytes': (c_sz, [c_i64, c_i64, c_i64, c_i64, c_i64]),
    'ppk_stackgru_seq_bwd': (c_int, [c_i64, ptr, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                                     ptr, ptr, ptr, ptr, ptr, ptr, ptr, c_i64,
                                     ptr, ptr, c_i64, ptr, ptr, ptr, ptr, ptr, ptr, c_sz, ptr]),
    'ppk_gather_rows': (c_int, [ptr, c_i64, c_i64, ptr, c_i64, c_i64, ptr, c_i64, ptr]),
    'ppk_scatter_rows': (c_int, [ptr, c_i64, ptr, c_i64, c_i64, ptr, c_i64, c_i64, ptr]),
    'ppk_rowsum_by_key_workspace_bytes': (c_sz, [c_i64, c_i64, c_i64]),
    'ppk_rowsum_by_key': (c_int, [ptr, c_i64, c_i64, ptr, c_i64, c_i64, ptr, c_i64, ptr, ptr, c_sz, ptr]),
    'ppk_colsum_workspace_bytes': (c_sz, [c_i64, c_i64]),
    'ppk_colsum': (c_int, [ptr, c_i64, c_i64, c_i64, c_f32, ptr, ptr, c_sz, ptr]),
    'ppk_colsum_noalloc': (c_int, [ptr, c_i64, c_i64, c_i64, c_f32, ptr, ptr]),
    'ppk_ce_workspace_bytes': (c_sz, [c_i64]),
    'ppk_ce_fwd': (c_int, [ptr, c_i64, ptr, c_i64, c_i64, c_i64, ptr, ptr, c_sz, ptr]),
    'ppk_ce_bwd': (c_int, [ptr, c_i64, ptr, c_i64, c_i64, c_i64, ptr, ptr, c_i64, ptr, c_sz, ptr]),
    'ppk_reparam_workspace_bytes': (c_sz, [c_i64]),
    'ppk_reparam_fwd': (c_int, [ptr, ptr, ptr, c_i64, ptr, ptr, ptr, c_sz, ptr]),
    'ppk_reparam_bwd': (c_int, [ptr, ptr, ptr, c_i64, ptr, ptr, ptr, ptr, ptr]),
    'ppk_adam_step': (c_int, [ptr, ptr, ptr, ptr, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i64, ptr]),
    'ppk_csr_build_workspace_bytes': (c_sz, [c_i64, c_i64]),
    'ppk_csr_build': (c_int, [ptr, ptr, c_i64, c_i64, c_int, c_int, c_int, ptr, ptr, ptr, ptr, ptr, ptr, ptr, c_sz, ptr]),
    'ppk_csr_build_rect': (c_int, [ptr, ptr, c_i64, c_i64, c_i64, c_int, c_int, c_int, ptr, ptr, ptr, ptr, ptr, ptr, ptr, c_sz, ptr]),
    'ppk_spmm_csr': (c_int, [ptr, ptr, ptr, c_i64, c_i64, ptr, c_i64, c_i64, ptr, c_i64, ptr]),
    'ppk_spmm_csr_workspace_bytes': (c_sz, [c_i64, c_i64, c_i64]),
    'ppk_spmm_csr_ws': (c_int, [ptr, ptr, ptr, c_i64, c_i64, ptr, c_i64, c_i64, ptr, c_i64, ptr, c_sz, ptr]),
    'ppk_prelu_fwd': (c_int, [ptr, ptr, c_i64, c_i64, ptr, ptr]),
    'ppk_prelu_bwd': (c_int, [ptr, ptr, ptr, c_i64, c_i64, ptr, ptr, ptr]),
    'ppk_dgi_loss_workspace_bytes': (c_sz, [c_i64]),
    'ppk_dgi_loss_fwd': (c_int, [ptr, ptr, c_i64, ptr, ptr, c_sz, ptr]),
    'ppk_dgi_loss_bwd': (c_int, [ptr, ptr, c_i64, ptr, ptr, ptr, ptr]),
    'ppk_graphreg_sparse_workspace_bytes': (c_sz, [c_i64, c_i64, c_i64]),
    'ppk_graphreg_sparse_dist': (c_int, [ptr, c_i64, c_i64, c_i64, ptr, c_i64, ptr, ptr, ptr, c_i64, c_i64, ptr, c_sz, ptr]),
    'ppk_graphreg_sparse_dist_once': (c_int, [ptr, c_i64, c_i64, c_i64, ptr, c_i64, ptr, ptr, ptr, ptr, ptr, ptr, c_sz, ptr]),
    'ppk_graphreg_sparse_max_ptr': (ptr, [ptr, c_i64, c_i64]),
    'ppk_graphreg_sparse_sum_ptr': (ptr, [ptr, c_i64, c_i64]),
    'ppk_graphreg_sparse_terms': (c_int, [c_i64, c_i64, ptr, ptr, c_i64, c_i64, ptr, c_sz, ptr]),
    'ppk_graphreg_sparse_finish': (c_int, [c_i64, c_i64, ptr, ptr, c_sz, ptr]),
    'ppk_graphreg_sparse_bwd': (c_int, [ptr, c_i64, c_i64, c_i64, c_i64, ptr, ptr, ptr, c_i64, c_i64,
                                        ptr, ptr, c_i64, ptr, c_sz, ptr]),
}


PPK_ERR_UNSUP = -4        # include/ppk.h


class PPKError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} is missing: build the HIP library first '
            '(make -C paccmann_polymer_amd/csrc). There is no CPU fallback.'
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()

_ERR_NAMES = {-1: 'shape', -2: 'alignment', -3: 'workspace', -4: 'unsupported', -5: 'hip', -6: 'null pointer'}


def check(rc: int):
    """0 -> ok.  Shape errors map to ValueError as in the reference
    (graph_constrained_loss.py:42-43); everything else to PPKError."""
    if rc == 0:
        return
    msg = lib.ppk_last_error().decode('utf-8', 'replace')
    if rc == -1:
        raise ValueError(msg)
    raise PPKError(f'ppk error {rc} ({_ERR_NAMES.get(rc, "?")}): {msg}')


def p(t):
    """device pointer of a tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise PPKError(
                'paccmann_polymer_amd runs only on MI355X HIP tensors (got a CPU tensor); '
                'there is no CPU fallback'
            )
