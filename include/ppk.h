/*
 * ppk.h — C-ABI of libppk.so: hand-written gfx950 (MI355X) kernels for the
 * training hot path of PaccMann/paccmann_polymer (the topologically
 * regularised SMILES-VAE step).
 *
 * The reference is pure Python and has no FFI of its own; each
 * entry point below names the reference code (file:line under the reference
 * root) whose arithmetic it replaces.  INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are DEVICE pointers unless
 *    the parameter is documented as host;
 *  - the caller allocates every buffer including the workspace
 *    (ppk_*_workspace_bytes); the library never allocates device memory,
 *    keeps no global state and never synchronises the stream;
 *  - every call enqueues on `stream` (a hipStream_t passed as void*);
 *  - return value: 0 on success, a negative PPK_ERR_* otherwise, with a
 *    message available from ppk_last_error() (thread-local);
 *  - all reductions have a fixed order: results are bitwise reproducible.
 */
#ifndef PPK_H
#define PPK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPK_OK            0
#define PPK_ERR_SHAPE    -1
#define PPK_ERR_ALIGN    -2
#define PPK_ERR_WS       -3
#define PPK_ERR_UNSUP    -4
#define PPK_ERR_HIP      -5
#define PPK_ERR_NULL     -6

#define PPK_F32 0
#define PPK_F64 1

typedef void* ppk_stream_t;   /* hipStream_t */

/* ---- library ------------------------------------------------------------ */
const char* ppk_version(void);
const char* ppk_last_error(void);
int         ppk_device_count(void);

/* ---- K4: dense graph-regularisation loss --------------------------------
 * Replaces graph_loss / compute_graph_dists / graph_regularizer_function,
 *   paccmann_polymer/topologically_regularized_models/graph_constrained_loss.py:4-67
 * (called from experiments/chem_reactions/reactions_vae.py:90-91 and
 *  experiments/cora/vae.py:89), in batched form: the batch holds n_seg
 * independent bundles; loss = sum_k graph_loss(z_k, G_k).
 *
 *  z        [n_total, zdim] fp32, row stride ldz (elements)
 *  g        the n_seg matrices G_k (n_k x n_k, row-major) concatenated;
 *           g_dtype PPK_F32 or PPK_F64 (compared exactly as the reference does)
 *  seg_ptr  [n_seg+1] int64 node offsets        (seg_ptr[n_seg] == n_total)
 *  mat_ptr  [n_seg+1] int64 matrix offsets      (mat_ptr[k] = sum_{k'<k} n_k'^2)
 *  max_n    max_k n_k (<= 3072)
 *  loss     [1]     fp32  sum_k loss_k
 *  seg_loss [n_seg] fp32  per-bundle loss
 * The backward call must receive the SAME workspace the forward call filled.
 *  dloss    [1] fp32 upstream gradient;  dz [n_total, zdim] fp32 (overwritten)
 */
size_t ppk_graphreg_dense_workspace_bytes(int64_t n_seg, int64_t n_total, int64_t sum_n2);
int ppk_graphreg_dense_fwd(const float* z, int64_t ldz, const void* g, int g_dtype,
                           const int64_t* seg_ptr, const int64_t* mat_ptr,
                           int64_t n_seg, int64_t n_total, int64_t sum_n2, int64_t max_n, int64_t zdim,
                           float* loss, float* seg_loss, void* workspace, size_t ws_bytes,
                           ppk_stream_t stream);
int ppk_graphreg_dense_bwd(const float* z, int64_t ldz, const int64_t* seg_ptr, const int64_t* mat_ptr,
                           int64_t n_seg, int64_t n_total, int64_t sum_n2, int64_t max_n, int64_t zdim,
                           const float* dloss, float* dz, int64_t lddz,
                           void* workspace, size_t ws_bytes, ppk_stream_t stream);

/* ---- fp32 MFMA GEMM ------------------------------------------------------
 * C[M,N] = alpha * opA(A) . opB(B) + beta * C + bias[N]  (row-major, exact fp32
 * v_mfma_f32_32x32x2_f32).  Replaces the dense contractions the reference
 * reaches through torch.nn.GRU / nn.Linear inside paccmann_chemistry
 * (call sites paccmann_polymer/models/paccmann_encoder.py:166-168) and
 * `x @ weight` of torch_geometric GCNConv (reactions_infomax.py:66).
 *   transA == 0: A is [M,K], lda >= K     transA != 0: A is [K,M], lda >= M
 *   transB == 0: B is [K,N], ldb >= N     transB != 0: B is [N,K], ldb >= K
 * A and B need 16-byte aligned bases and lda/ldb multiples of 4.  bias may be
 * NULL.  The optional workspace (ppk_gemm_f32_workspace_bytes) enables the
 * deterministic split-K path for small M x N with long K.
 */
size_t ppk_gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K);
int ppk_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                 const float* A, int64_t lda, const float* B, int64_t ldb, float beta,
                 float* C, int64_t ldc, const float* bias, void* workspace, size_t ws_bytes,
                 ppk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PPK_H */
