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
 *    (ppk_*_workspace_bytes); the library never allocates device memory and
 *    never synchronises the stream.  Its only process-wide state: the opt-in
 *    timing table (ppk_prof_*), the dispatch knobs ppk_nt_small_grid_limit /
 *    ppk_seq_persist_mode / ppk_mfma_mode (and the PPK_* environment variables
 *    read once behind them), the cached device properties, and the per-device
 *    address of the static array behind ppk_colsum_noalloc;
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

/* Opt-in per-kernel-family timing with HIP events on the launch stream (no
 * reference counterpart; bench.py uses it for the roofline figures).  Process-wide,
 * not thread-safe; off by default.  ppk_prof_collect synchronises the recorded
 * events and fills three arrays of ppk_prof_kind_count() entries. */
int         ppk_prof_enable(int on);
int         ppk_prof_kind_count(void);
const char* ppk_prof_kind_name(int kind);
int         ppk_prof_collect(double* ms, double* work, int64_t* launches);

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

/* ---- K5: sparse (edge-list) graph-regularisation loss ------------------------
 * BUILD-DEFINED generalisation of graph_loss (graph_constrained_loss.py:25-67) for
 * relation graphs too large for the dense O(N^3) form: every compared pair is
 * restricted to an undirected edge set with attribute g[e]; normaliser = max edge
 * distance.  Equals the reference on a complete edge set.
 * (rowptr, col, eid) = CSR (ppk_csr_build, no self loops, no normalisation) of the
 * SYMMETRISED edge list: entry ids e in [0, E) for u->v and E + e for v->u; rev [2E]
 * maps each entry to its reverse entry.  Anchors [a0, a1) let ranks own node ranges:
 * dist -> (MAX-all-reduce *max_ptr) -> terms -> (SUM-all-reduce sum_ptr[0..1]: the sum S and the number of edges
 * whose distance equals the maximum — the gradient through the normaliser is shared evenly by them, as torch's
 * backward of `z_dists.max()`, graph_constrained_loss.py:40, does) -> finish;
 * bwd yields this rank's partial dz [N, zdim] (reduce-scatter across ranks).
 * dist / dist_once with eid == NULL take g as [2E] attributes already in CSR order (permuted once per graph). */
size_t ppk_graphreg_sparse_workspace_bytes(int64_t N, int64_t E, int64_t zdim);
int ppk_graphreg_sparse_dist(const float* z, int64_t ldz, int64_t N, int64_t zdim, const float* g, int64_t E,
                             const int64_t* rowptr, const int64_t* col, const int64_t* eid, int64_t a0, int64_t a1,
                             void* workspace, size_t ws_bytes, ppk_stream_t stream);
/* dist for ALL anchors of a graph held by one GPU (a0 = 0, a1 = N), every undirected edge computed once:
 * up [E] = CSR positions of the entries whose row is the smaller endpoint (ascending), up_row [E] = their rows.
 * Leaves the workspace bit for bit as ppk_graphreg_sparse_dist(..., 0, N, ...) does, with half the gather traffic. */
int ppk_graphreg_sparse_dist_once(const float* z, int64_t ldz, int64_t N, int64_t zdim, const float* g, int64_t E,
                                  const int64_t* col, const int64_t* eid, const int64_t* up, const int64_t* up_row,
                                  const int64_t* rev, void* workspace, size_t ws_bytes, ppk_stream_t stream);
float*  ppk_graphreg_sparse_max_ptr(void* workspace, int64_t N, int64_t E);
double* ppk_graphreg_sparse_sum_ptr(void* workspace, int64_t N, int64_t E);
int ppk_graphreg_sparse_terms(int64_t N, int64_t E, const int64_t* rowptr, const int64_t* col, int64_t a0, int64_t a1,
                              void* workspace, size_t ws_bytes, ppk_stream_t stream);
int ppk_graphreg_sparse_finish(int64_t N, int64_t E, float* loss, void* workspace, size_t ws_bytes, ppk_stream_t stream);
int ppk_graphreg_sparse_bwd(const float* z, int64_t ldz, int64_t N, int64_t zdim, int64_t E,
                            const int64_t* rowptr, const int64_t* col, const int64_t* rev, int64_t a0, int64_t a1,
                            const float* dloss, float* dz, int64_t lddz, void* workspace, size_t ws_bytes, ppk_stream_t stream);

/* ---- fp32 MFMA GEMM ------------------------------------------------------
 * C[M,N] = alpha * opA(A) . opB(B) + beta * C + bias[N]  (row-major, exact fp32
 * v_mfma_f32_32x32x2_f32).  Replaces the dense contractions the reference
 * reaches through torch.nn.GRU / nn.Linear inside paccmann_chemistry
 * (call sites paccmann_polymer/models/paccmann_encoder.py:166-168) and
 * `x @ weight` of torch_geometric GCNConv (reactions_infomax.py:66).
 *   transA == 0: A is [M,K], lda >= K     transA != 0: A is [K,M], lda >= M
 *   transB == 0: B is [K,N], ldb >= N     transB != 0: B is [N,K], ldb >= K
 * 16-byte aligned bases with lda/ldb multiples of 4 take the float4 load path
 * (anything else falls back to scalar loads).  bias may be NULL.  The optional workspace (ppk_gemm_f32_workspace_bytes) enables the
 * deterministic split-K path for small M x N with long K.
 */
size_t ppk_gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K);
int ppk_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                 const float* A, int64_t lda, const float* B, int64_t ldb, float beta,
                 float* C, int64_t ldc, const float* bias, void* workspace, size_t ws_bytes,
                 ppk_stream_t stream);

/* C = A^T . B and colsum[m] = sum_k A[k, m] in ONE pass over A (A [K, M], B [K, N], row-major): a weight gradient
 * dW = dG^T . X together with its bias gradient db = sum_rows dG — the pair torch autograd produces for nn.GRU /
 * nn.Linear parameters in the reference step (reactions_vae.py:94 loss.backward()).  Fixed-order sums (bitwise
 * reproducible).  Returns PPK_ERR_UNSUP with nothing launched when the shape does not take the fused kernel
 * (needs long K >= 8192, N > 192, M % 4 == N % 4 == 0, 16-byte aligned operands): use ppk_gemm_f32 + ppk_colsum.
 * Workspace: ppk_gemm_f32_workspace_bytes(M, N, K). */
int ppk_gemm_tn_colsum_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                           float* C, int64_t ldc, float* colsum, void* workspace, size_t ws_bytes, ppk_stream_t stream);

/* Tuning / test hook of the NT GEMM family behind ppk_gemm_f32 (NT, K % 32 == 0) and K1: launches
 * whose big-tile grid has at most `limit` tiles run on the small-tile kernels (one 32-row tile per
 * workgroup, K split over its four waves) that keep short packed-batch tail steps — and the
 * reference's own <= 21-sequence bundles (reactions_vae.py:52-59) — off the single-CU latency floor.
 * limit == -1 restores the defaults (224 tiles for the 64 x 64 family, 128 for the gate-layout family, from MI355X kernel traces);
 * limit < -1 only queries; returns the value in force. */
int64_t ppk_nt_small_grid_limit(int64_t limit);

/* Launch schedule of K1 (ppk_stackgru_seq_fwd / _bwd).  Three schedules compute the same step: the whole time loop as ONE
 * persistent launch per pass over row-block groups (64 rows x H / 32 workgroups, synchronised among their own workgroups
 * only, weights streamed), one launch per layer-step, and — for small and medium batches, first of all the reference's own
 * regime of one <= 21-SMILES bundle per optimizer step (reactions_vae.py:52-59) — ONE launch per pass of replicas of <= 16 (32)
 * sequences, each H / 16 workgroups that keep their rows of every weight matrix in registers for the whole launch
 * (ppk_seq_ws.hip).  mode:
 *   -1  auto (default; PPK_SEQ_PERSIST in the environment): the weights-stationary launch wherever the shape is instantiated
 *       and one launch fits the device (below: H 512 up to 256 sequences, H 256 up to 512; PPK_WS_MAX_ROWS in the environment
 *       lowers that); the row-block persistent launch where it is faster on MI355X — when the batch's
 *       row-block groups make one machine-full that is at least 3/8 full at H >= 384
 *       (9/10 at H >= 256) or two that are at least 9/10 full at H >= 512: e.g. H 512 from batch 1024 to 4096, H 256 at batch
 *       4096, never at H 128 (tools/persist_threshold_shapes.py, profiles/r03_persist_shapes.txt) —, per-step launches elsewhere;
 *    0  never a persistent launch of either kind;
 *    1  row-block persistent at ANY batch, wherever the shape is supported (H % 32 == 0, ks / ldcs at their minimal paddings
 *       roundup32(W) / roundup32(3 + W), one group's workgroups resident together on the current device); an
 *       unsupported shape takes the per-step launches.  Never the weights-stationary kernels;
 *    2  the weights-stationary kernels wherever one launch of <= 32-row replicas fits (never the looped forward), auto's choice elsewhere;
 *    3  like 2, and the looped weights-stationary forward (replicas of up to 256 rows; forward only) for every larger batch it fits —
 *       auto takes that forward only where it was measured faster (rows per replica x H <= 32768: H 512 batch 257 - 512, H 256 batch
 *       513 - 1536); at batch 2048 / H 512 it is slower than the row-block kernel (profiles/r03_ws_replicas.txt).
 * Weights-stationary shapes: ceil(B / 16) replicas (fewer, of up to 32 rows, when replicas x H / 16 would exceed the CUs), L <= 2 (3 at H <= 256), H in {32, 64, 128, 256, 512} with ks == ldcs-minimal paddings, ldcs <= 64
 * (H >= 256: ks == 64 only), the owned rows' stacks within the workgroup's LDS (32 / (H / 16) rows x D x W floats: 2 images
 * forward, 3 backward).  fp32 operands whatever ppk_mfma_mode says.  Measured (MI355X, H 512, L 2,
 * bundles of 6 - 21 sequences of 20 - 100 tokens, whole training step): 15.7 ms -> 8.0 ms (profiles/r03_small_batch_ws.txt); batch 64: 17.2 -> 9.0 ms, 256: 19.9 -> 12.9 ms
 * (profiles/r03_ws_replicas.txt).
 * Any other value only queries.  Process-global like ppk_nt_small_grid_limit.  Returns the value in force.
 * A persistent launch (either kind) whose workgroups lose a hand-off (bounded wait, ~3 s; another process holding the GPU can cause
 * it) fills every output of the call with NaN: the entry point still returns PPK_OK (it never synchronises). */
int ppk_seq_persist_mode(int mode);

/* Arithmetic of the MFMA products of K1's persistent sequence kernels (BASELINE.json configs[4], "bf16-MFMA variant
 * with stated tolerance"; no reference counterpart — the reference computes in fp32 throughout).
 *   0 (default): fp32 operands, v_mfma_f32_32x32x2_f32 — the parity path (results match the fp64 oracle to 1e-5).
 *   1: operands rounded to bf16 (round to nearest even) in registers, v_mfma_f32_32x32x8_bf16, fp32 accumulation;
 *      everything outside the products (gate math, stack, hand-offs, memory) stays fp32.  The long-K weight-gradient
 *      GEMM behind ppk_gemm_tn_colsum_f32 / ppk_gemm_f32 (TN, k_tn128) follows the same mode.  Tolerance vs mode 0 on the
 *      training step: loss 2e-3 relative, gradients 5e-2 of the largest entry per tensor.  STATED TOLERANCE, against the fp64
 *      oracle on the full model at BASELINE configs[0] size (tests/test_gpu_bf16.py::test_bf16_step_vs_fp64_oracle_configs0):
 *      losses 1e-2 relative, latents 2e-2, every gradient tensor within 1e-1 of its largest entry, cosine >= 0.99
 *      (measured: losses 1e-4, worst gradient 3e-2).
 * In K1 it takes effect only where the persistent kernels run: in auto mode (ppk_seq_persist_mode -1) that is where the
 * rule above picks them (H 512: from batch 1024 up) — elsewhere mode 1 changes nothing in K1 unless ppk_seq_persist_mode(1)
 * forces the persistent launch; the small-batch weights-stationary kernels (<= 32 sequences) always multiply fp32 operands.  Any other
 * value only queries.  Process-global.  Returns the value in force. */
int ppk_mfma_mode(int mode);

/* ---- K1: stack-augmented GRU sequence (packed, length-sorted batch) -------
 * Replaces the per-timestep loop of paccmann_chemistry StackGRU / nn.GRU that
 * the reference drives from paccmann_polymer/models/paccmann_encoder.py:166
 * (TeacherVAE.encode) and :168 (TeacherVAE.decode); semantics SURVEY.md
 * Appendix B.  batch_sizes_host is a HOST array [T] (PackedSequence.batch_sizes);
 * w_ih / w_hh / b_ih / b_hh are HOST arrays of L device pointers
 * (nn.GRU weight_ih_l{k} [3H, H] and bias_ih_l{k} for k >= 1 — entry 0 is not read —,
 * weight_hh_l{k} [3H, H], bias_hh_l{k}; gate order r|z|n).  The layer-0 input product
 * is split: table [V, 3H] = embedding.weight . weight_ih_l0[:, :E]^T + bias_ih_l0 (a
 * lookup per token, hoisted out of the time loop) and wx0s [3H, ks] = weight_ih_l0[:, E:E+W]
 * zero padded to ks = W rounded up to 32.  wcs [3+W, H] / bcs [3+W] are
 * stack_controls_layer rows 0..2 followed by stack_input_layer; h0 [L, B, H]; S0 [B, D, W] = stack entering
 * step 0 (NULL = zeros: training starts from the empty stack; the single-step decoder call of
 * paccmann_encoder.py:115-131 -> TeacherVAE.generate carries it from token to token).
 * H must be a multiple of 32 (the host side pads with zero units); ldcs = 3+W rounded up to 32.
 * Per step: stack update, then one MFMA GEMM launch per layer with the GRU gate math in
 * its epilogue (plus the recurrent products needed by the next step).
 * Saved activations (caller-allocated, consumed by the backward):
 *   XS [NT, ks] (stack top), Hs / HP [L, NT, H], gates [L, NT, 4H], S [NT, D, W], CS [NT, ldcs].
 * Training-mode dropout of nn.GRU(dropout = p) (the reference trains with gru_vae.train(), reactions_vae.py:44, on
 * params whose `dropout` key comes from the pretrained JSON, paccmann_encoder.py:41-45): torch applies it to the
 * output of every layer but the last, on the way to the next layer only (the recurrent path keeps the undropped
 * state).  drop [L-1, NT, H] holds the factor of every (layer, packed row, unit): 0 or 1 / (1 - p)
 * (ppk_dropout_factors draws them; a test injects its own); Hdrop [L-1, NT, H] receives Hs * drop, the input
 * of the layer above (saved: it is the operand of dW_ih[l >= 1]).  drop == NULL: no dropout, Hdrop unused.
 */
size_t ppk_stackgru_seq_fwd_workspace_bytes(int64_t B, int64_t H, int64_t L, int64_t W);
int ppk_stackgru_seq_fwd(int64_t T, const int64_t* batch_sizes_host, int64_t B, int64_t NT,
                         int64_t V, int64_t H, int64_t L, int64_t W, int64_t D,
                         const int64_t* tokens, const float* table, const float* wx0s, int64_t ks,
                         const float* const* w_ih, const float* const* w_hh,
                         const float* const* b_ih, const float* const* b_hh,
                         const float* wcs, const float* bcs, const float* h0, const float* S0,
                         float* XS, float* Hs, float* HP, float* gates, float* S,
                         float* CS, int64_t ldcs, const float* drop, float* Hdrop,
                         void* workspace, size_t ws_bytes, ppk_stream_t stream);
/* drop factors for ppk_stackgru_seq_fwd / _bwd: out[l, row, u] = keep ? 1 / (1 - p) : 0 with keep drawn by Philox4x32-10
 * keyed by (seed, l, row, u / 4): a pure function of its index, independent of launch geometry and schedule.
 * Units >= H_valid (zero padding of the hidden size) get factor 0.  0 <= p < 1. */
int ppk_dropout_factors(uint64_t seed, float p, int64_t layers, int64_t NT, int64_t H, int64_t H_valid, float* out,
                        ppk_stream_t stream);
/* dHtop [NT, H] = gradient at the top layer's output of every step.  Outputs:
 * dXS [NT, ks] (gradient of the stack top fed to layer 0), DGI / DGH [L, NT, 3H] (gate
 * pre-activation gradients of the input / hidden side), DCS [NT, ldcs], dh0 [L, B, H].
 * Weight gradients are single GEMMs over these: dW_ih[l>=1] = DGI^T Hs[l-1] (Hdrop[l-1] under dropout), dW_ih0[:, E:] = DGI0^T XS,
 * dtable = rows of DGI0 summed per token, dW_hh = DGH^T HP, dWcs = DCS^T HP[L-1].
 * The backward takes the weights TRANSPOSED (host arrays of L device pointers:
 * w_ih_t[0] [W, 3H] = stack columns of weight_ih_l0, w_ih_t[l>=1] [H, 3H], w_hh_t[l] [H, 3H];
 * wcs_t [H, ldcs] zero padded) so that the per-step data-gradient GEMMs read k-contiguous operands.
 * drop: the factors the forward ran with (NULL: none). */
size_t ppk_stackgru_seq_bwd_workspace_bytes(int64_t B, int64_t H, int64_t L, int64_t W, int64_t D);
int ppk_stackgru_seq_bwd(int64_t T, const int64_t* batch_sizes_host, int64_t B, int64_t NT,
                         int64_t H, int64_t L, int64_t W, int64_t D,
                         const float* const* w_ih_t, const float* const* w_hh_t, const float* wcs_t,
                         const float* HP, const float* gates, const float* S, const float* CS, int64_t ldcs,
                         const float* dHtop, float* dXS, int64_t ks, float* DGI, float* DGH, float* DCS,
                         float* dh0, const float* drop, void* workspace, size_t ws_bytes, ppk_stream_t stream);

/* ---- row gather / scatter, column sums, embedding gradient ----------------
 * (index_select / index_put / sum(0) / nn.Embedding backward in the reference's
 * torch graph.)  All deterministic. */
int ppk_gather_rows(const float* src, int64_t ld_src, int64_t src_rows, const int64_t* idx, int64_t n_rows, int64_t width,
                    float* dst, int64_t ld_dst, ppk_stream_t stream);
int ppk_scatter_rows(const float* src, int64_t ld_src, const int64_t* idx, int64_t n_rows, int64_t width,
                     float* dst, int64_t ld_dst, int64_t dst_rows, ppk_stream_t stream);
size_t ppk_colsum_workspace_bytes(int64_t rows, int64_t cols);
int ppk_colsum(const float* A, int64_t lda, int64_t rows, int64_t cols, float beta, float* out,
               void* workspace, size_t ws_bytes, ppk_stream_t stream);
/* ppk_colsum WITHOUT a workspace and without any allocation.  No counterpart in the reference: it exists for callers
 * that cannot allocate device memory — code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  The partial sums live in a 1 MiB file-scope device array of the library (one copy per GPU, held until the
 * process ends, reached through hipGetSymbolAddress on the current device at the first call on that device).
 * Limit: ceil(rows / 512) * cols <= 262144, otherwise PPK_ERR_UNSUP with nothing launched.  Bit-identical to ppk_colsum.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (streams, threads) race. */
int ppk_colsum_noalloc(const float* A, int64_t lda, int64_t rows, int64_t cols, float beta, float* out, ppk_stream_t stream);

/* out[k, :] = sum of the rows r of x with keys[r] == k (k in [0, n_keys)), in a fixed order (ascending r inside 64-row
 * chunks, a key's chunks ascending): bitwise reproducible, sort-free (stable counting sort on the device).
 * The embedding-table gradient of K1: autograd's nn.Embedding backward for the reference (paccmann_encoder.py:166,168)
 * scatters per token with float atomics; here the rows of the layer-0 gate gradients are summed per token id.
 * n_keys <= 8192; F, ldx, ldo multiples of 4; status [1] (device int) becomes non-zero if a key was out of range
 * (those rows are skipped). */
size_t ppk_rowsum_by_key_workspace_bytes(int64_t n_rows, int64_t n_keys, int64_t F);
int ppk_rowsum_by_key(const int64_t* keys, int64_t n_rows, int64_t n_keys, const float* x, int64_t ldx, int64_t F,
                      float* out, int64_t ldo, int* status, void* workspace, size_t ws_bytes, ppk_stream_t stream);

/* ---- K2: cross-entropy over packed logits ---------------------------------
 * nn.CrossEntropyLoss(ignore_index=0) of TeacherVAE.decode (paccmann_encoder.py:168):
 * mean over rows whose target != ignore_index.  The workspace carries the
 * per-row logsumexp and the valid count from fwd to bwd. */
size_t ppk_ce_workspace_bytes(int64_t rows);
int ppk_ce_fwd(const float* logits, int64_t ldl, const int64_t* targets, int64_t rows, int64_t V, int64_t ignore_index,
               float* loss, void* workspace, size_t ws_bytes, ppk_stream_t stream);
int ppk_ce_bwd(const float* logits, int64_t ldl, const int64_t* targets, int64_t rows, int64_t V, int64_t ignore_index,
               const float* dloss, float* dlogits, int64_t ldd, const void* workspace, size_t ws_bytes, ppk_stream_t stream);

/* ---- K3: reparameterise + KL ------------------------------------------------
 * TeacherVAE.reparameterize (paccmann_encoder.py:167; in-tree twin
 * experiments/cora/vae.py:31-34) with injected eps, fused with the KL term
 * -0.5 sum(1 + logvar - mu^2 - exp(logvar)) (cora/vae.py:50).  kl / dz / dkl may be NULL. */
size_t ppk_reparam_workspace_bytes(int64_t n);
int ppk_reparam_fwd(const float* mu, const float* logvar, const float* eps, int64_t n, float* z, float* kl,
                    void* workspace, size_t ws_bytes, ppk_stream_t stream);
int ppk_reparam_bwd(const float* mu, const float* logvar, const float* eps, int64_t n, const float* dz, const float* dkl,
                    float* dmu, float* dlogvar, ppk_stream_t stream);

/* ---- flat Adam --------------------------------------------------------------
 * optimizer.step() of reactions_vae.py:96 (OPTIMIZER_FACTORY['adam'] = torch.optim.Adam)
 * over one flat fp32 parameter buffer; step is 1-based. */
int ppk_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, ppk_stream_t stream);

/* ---- batched hop-distance matrices of bundle graphs ------
 * Replaces, per bundle, reactions_vae.py:79-83 (nx.from_edgelist + add_nodes_from + floyd_warshall_numpy, inf -> 0):
 * undirected all-pairs hop counts as fp64, 0 for unreachable pairs and the diagonal, rows in networkx node INSERTION
 * order (networkx_order != 0; Quirk C) or in index order.  esrc / edst [E_total]: LOCAL labels 0 .. n_k - 1 of the
 * concatenated edge lists; edge_ptr / seg_ptr / mat_ptr [n_seg + 1]: device prefix arrays over edges, nodes, n_k^2.
 * G [mat_ptr[n_seg]]; order_out [seg_ptr[n_seg]] (may be NULL): original label of each row; status [1] device int,
 * non-zero when a label was out of range (those edges are skipped).  Bundles of at most 64 nodes (max_n). */
int ppk_hop_distances(const int64_t* esrc, const int64_t* edst, const int64_t* edge_ptr, const int64_t* seg_ptr,
                      const int64_t* mat_ptr, int64_t n_seg, int64_t max_n, int networkx_order,
                      double* G, int64_t* order_out, int* status, ppk_stream_t stream);

/* ---- device-side bundle sampler ---------------------------
 * Replaces, for R root nodes at once, experiments/chem_reactions/utils.py:155-196 (the 2-hop bundle of load_data):
 * max_close_neighbors distinct neighbours of the root (the head of a uniform permutation of its neighbour list), then
 * uniformly chosen neighbours of those, round-robin, until max_total_bundle [root-or-close, neighbour] pairs or
 * total_far draws; then the relabelling of nx.from_edgelist -> convert_node_labels_to_integers -> to_pandas_edgelist
 * (nodes by first appearance, every undirected edge once, by node then adjacency insertion order).
 * (rowptr, col): adjacency lists of the relation graph in networkx insertion order, N nodes.  roots [R] or NULL (= 0..R-1).
 * The reference's single numpy Mersenne-Twister stream through all roots is replaced by one Philox4x32-10 stream per root
 * (key = seed, counter = (draw index, 0, root)): same distribution, other numbers; oracle/bundle_sampler_ref.py replays the
 * draws through the host restatement.  Outputs, per root r: shortlist [R, max_total, 2] original labels + n_short [R];
 * labels [R, max_total + 1] original label of local node p + n_nodes [R]; edges [R, max_total, 2] local labels + n_edges [R].
 * status [1] device int: bit 0 when a root was outside [0, N), bit 1 when a neighbour id of a root was (those roots get empty
 * bundles; nothing is read through a bad id).  max_total <= 32. */
int ppk_sample_bundles(const int64_t* rowptr, const int64_t* col, int64_t N, const int64_t* roots, int64_t R,
                       int max_close, int max_total, uint64_t seed, int64_t* shortlist, int* n_short,
                       int64_t* labels, int* n_nodes, int64_t* edges, int* n_edges, int* status, ppk_stream_t stream);

/* ---- K6: CSR build (stable sort of edge_index by target) --------------------
 * gcn_norm + the grouping that torch_scatter.scatter_add performs implicitly inside
 * torch_geometric GCNConv (reference reactions_infomax.py:60,66).
 * Row i of the CSR lists the edges whose target is i in ORIGINAL edge-list order,
 * the appended self loop last.  rowptr [N+1], col / eid / val [E + N] (nnz = rowptr[N]).
 * eid is the original edge id (E + i for node i's self loop).  status [1] (device int)
 * becomes non-zero if an endpoint was out of range (those edges are skipped). */
size_t ppk_csr_build_workspace_bytes(int64_t N, int64_t E);
int ppk_csr_build(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int drop_self_loops, int add_self_loops,
                  int normalize, const float* edge_weight, int64_t* rowptr, int64_t* col, int64_t* eid, float* val,
                  int* status, void* workspace, size_t ws_bytes, ppk_stream_t stream);
/* The same grouping for a rectangular incidence: dst in [0, N) is the CSR row, src in [0, N_src) indexes another table.
 * Used by the embedding-table gradient of K1 (rows of the packed batch grouped by token: the one-hot product that
 * autograd's nn.Embedding backward performs for the reference, paccmann_encoder.py:166,168, as a segmented reduction in
 * a fixed order).  Self loops and normalisation need N_src == N. */
int ppk_csr_build_rect(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int64_t N_src, int drop_self_loops,
                       int add_self_loops, int normalize, const float* edge_weight, int64_t* rowptr, int64_t* col, int64_t* eid,
                       float* val, int* status, void* workspace, size_t ws_bytes, ppk_stream_t stream);
/* ---- K7: out[i, :] = sum_{j in row i} val[j] * x[col[j], :] --------------------
 * One wavefront per row, fixed order, no atomics (GCNConv.propagate's gather +
 * scatter_add).  The backward is the same call on the CSR of the reversed edges. */
int ppk_spmm_csr(const int64_t* rowptr, const int64_t* col, const float* val, int64_t N, int64_t nnz,
                 const float* x, int64_t ldx, int64_t F, float* out, int64_t ldo, ppk_stream_t stream);
/* The same product for graphs with hub rows (power-law relation graphs): rows beyond 1024 entries are cut into
 * 1024-entry chunks, one wavefront per chunk over the whole grid, and a row's partial sums are added in chunk order
 * (fixed association: bitwise reproducible; short rows exactly as ppk_spmm_csr).  ppk_spmm_csr reduces such a row
 * with one workgroup.  Workspace: ppk_spmm_csr_workspace_bytes(N, nnz, F). */
size_t ppk_spmm_csr_workspace_bytes(int64_t N, int64_t nnz, int64_t F);
int ppk_spmm_csr_ws(const int64_t* rowptr, const int64_t* col, const float* val, int64_t N, int64_t nnz,
                    const float* x, int64_t ldx, int64_t F, float* out, int64_t ldo, void* workspace, size_t ws_bytes,
                    ppk_stream_t stream);

/* ---- PReLU and the DeepGraphInfomax loss (reactions_infomax.py:61,67,131-132,251-264) */
int ppk_prelu_fwd(const float* x, const float* alpha, int64_t rows, int64_t C, float* y, ppk_stream_t stream);
int ppk_prelu_bwd(const float* x, const float* alpha, const float* dy, int64_t rows, int64_t C, float* dx, float* t,
                  ppk_stream_t stream);
size_t ppk_dgi_loss_workspace_bytes(int64_t N);
int ppk_dgi_loss_fwd(const float* pos_logit, const float* neg_logit, int64_t N, float* loss, void* workspace, size_t ws_bytes,
                     ppk_stream_t stream);
int ppk_dgi_loss_bwd(const float* pos_logit, const float* neg_logit, int64_t N, const float* dloss, float* dpos, float* dneg,
                     ppk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PPK_H */
