// Internal C++ interfaces shared between the translation units of libppk.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace ppk {

// C[M,N] = alpha * opA(A) . opB(B) + beta * C + bias[N]   (row-major, fp32 MFMA)
//   transA == 0: A is [M,K] (lda >= K)      transA != 0: A is [K,M] (lda >= M)
//   transB == 0: B is [K,N] (ldb >= N)      transB != 0: B is [N,K] (ldb >= K)
int gemm(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
         const float* A, int64_t lda, const float* B, int64_t ldb, float beta,
         float* C, int64_t ldc, const float* bias, void* ws, size_t ws_bytes, hipStream_t stream);
size_t gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);

// ---- NT GEMMs with LDS-DMA staging and fused GRU epilogues (ppk_gemm_nt.hip) ----
//   acc[M,N] = A[M,K] . B[N,K]^T ; K % 32 == 0, lda/ldb % 4 == 0, 16-byte aligned operands.
//   Only the first a_rows rows of A exist: rows >= a_rows contribute acc = 0 (their loads are clamped).
enum NtEpi { NT_STORE = 0, NT_GRU_FWD = 1, NT_GRU_BWD = 2 };
constexpr int NT_MAX_GROUP = 4;
struct NtProblem {
    const float* A; const float* B; long long lda, ldb;
    int M, N, K, a_rows;
    int epi;
    int H;                       // hidden size (gate layout: N == 3H ; NT_GRU_BWD: N == H)
    // NT_STORE: C = acc + bias[col] + beta * C
    float* C; long long ldc; const float* bias; float beta;
    // NT_GRU_FWD (gate layout): gi = acc + bias (+ table[tok[row]]) ; gh [M,3H] holds W_hh h + b_hh
    const float* gh; const float* hprev; const int64_t* tok; const float* table;
    int V;                       // rows of `table`: token ids are clamped to [0, V) in the kernel (the host validates them)
    float* hnew; float* hp_out; float* gates_out;          // [M,H], [M,H], [M,4H] = r, z, n, gh_n
    // inter-layer dropout of nn.GRU(dropout = p) in training mode (NULL: none).  drop [M,H] holds 0 or 1 / (1 - p):
    //   NT_GRU_FWD: hdrop = hnew * drop is what the layer above reads;  NT_GRU_BWD: dh = acc * drop + carry (+ ext)
    const float* drop; float* hdrop;
    // NT_GRU_BWD: dh = acc + carry (+ ext) ; writes dgi, dgh [M,3H] and carry = dh * z
    float* carry; const float* ext; const float* gates_in; const float* hp_in; float* dgi; float* dgh;
};
bool gemm_nt_fast_ok(int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb);
int gemm_nt64(const NtProblem* ps, int count, hipStream_t stream);   // 64 x 64 tiles: NT_STORE, NT_GRU_BWD
int gemm_ntF(const NtProblem* ps, int count, hipStream_t stream);    // 128 x (3 x 32) gate-layout tiles: NT_STORE, NT_GRU_FWD


// ---- persistent StackGRU sequence kernels (ppk_seq_persist.hip): the whole time loop in one launch,
//      row-block groups synchronised only among their own workgroups ----
// auto mode (ppk_seq_persist_mode(-1)): which schedule runs a batch of B rows at hidden size H.  A group (64 rows) is
// H / 32 workgroups and a launch carries at most one machine-full of whole groups; the persistent launch pays its
// per-timestep hand-off chain once per machine-full, the per-step launches pay per tile.  Measured on MI355X over
// H 128 .. 1024, L 1 .. 3, B 512 .. 4096 (tools/persist_threshold_shapes.py, profiles/r03_persist_shapes.txt; training
// step, persistent / per-step): the persistent launch wins (0.84 .. 0.99) with ONE machine-full that is at least 3/8 full
// at H >= 384 (or at least 9/10 full at H >= 256), and with two machine-fulls that are at least 9/10 full at H >= 512; it loses
// (1.01 .. 2.0) at H = 128, with a half-empty second machine-full, and from three machine-fulls up.
// H 512, L 2 (tools/persist_threshold.py; per-step / persistent, ms): B 16: 16.0 / 21.6, 64: 17.2 / 29.5, 256: 19.9 / 27.8,
// 512: 27.3 / 29.3, 1024: 35.4 / 33.0, 2048: 52.9 / 45.5
bool seq_auto_persistent(int64_t B, int64_t H, int groups_per_launch, int capacity);
constexpr int SEQP_MAXT = 512, SEQP_MAXL = 8;
int mfma_mode(int mode = -1);        // 0 fp32 operands (default), 1 bf16 operands; other values only query
int seq_persist_mode(int mode);      // -1 auto (default), 0 never, 1 row-block persistent whenever supported, 2 = auto + the small-batch
                                     // weights-stationary kernels whenever supported; other values only query
bool seq_fwd_persist_ok(int64_t T, int64_t B, int64_t NT, int64_t H, int64_t L, int64_t W, int64_t D, int64_t ks, int64_t ldcs);
size_t seq_fwd_persist_ws_bytes(int64_t B, int64_t H, int64_t L, int64_t W);
int seq_fwd_persist(int64_t T, const int64_t* bs, int64_t B, int64_t NT, int64_t V, int64_t H, int64_t L, int64_t W, int64_t D,
                    const int64_t* tokens, const float* table, const float* wx0s, int64_t ks,
                    const float* const* w_ih, const float* const* w_hh, const float* const* b_ih, const float* const* b_hh,
                    const float* wcs, const float* bcs, const float* h0, const float* S0,
                    float* XS, float* Hs, float* HP, float* gates, float* S, float* CS, int64_t ldcs,
                    const float* drop, float* Hdrop, void* ws, size_t ws_bytes, hipStream_t stream);

// ---- small-batch (<= 32 rows) weights-stationary forward (ppk_seq_ws.hip): H / 16 workgroups keep their gate rows of
//      every matrix in registers for the whole launch; same arguments, outputs and workspace as seq_fwd_persist ----
size_t seq_fwd_ws_ws_bytes(int64_t B, int64_t H, int64_t L, int64_t W);      // 0 where these kernels do not apply
size_t seq_bwd_ws_ws_bytes(int64_t B, int64_t H, int64_t L, int64_t W);
bool seq_fwd_ws_ok(int64_t T, int64_t B, int64_t NT, int64_t H, int64_t L, int64_t W, int64_t D, int64_t ks, int64_t ldcs);
int seq_fwd_ws(int64_t T, const int64_t* bs, int64_t B, int64_t NT, int64_t V, int64_t H, int64_t L, int64_t W, int64_t D,
               const int64_t* tokens, const float* table, const float* wx0s, int64_t ks,
               const float* const* w_ih, const float* const* w_hh, const float* const* b_ih, const float* const* b_hh,
               const float* wcs, const float* bcs, const float* h0, const float* S0,
               float* XS, float* Hs, float* HP, float* gates, float* S, float* CS, int64_t ldcs,
               const float* drop, float* Hdrop, void* ws, size_t ws_bytes, hipStream_t stream);

// looped forward (mode 3, experiment): replicas of up to 256 rows walked 32 at a time, weights in registers
size_t seq_fwd_wsl_ws_bytes(int64_t B, int64_t H, int64_t L, int64_t W);
bool seq_fwd_wsl_ok(int64_t T, int64_t B, int64_t NT, int64_t H, int64_t L, int64_t W, int64_t D, int64_t ks, int64_t ldcs);
int seq_fwd_wsl(int64_t T, const int64_t* bs, int64_t B, int64_t NT, int64_t V, int64_t H, int64_t L, int64_t W, int64_t D,
                const int64_t* tokens, const float* table, const float* wx0s, int64_t ks,
                const float* const* w_ih, const float* const* w_hh, const float* const* b_ih, const float* const* b_hh,
                const float* wcs, const float* bcs, const float* h0, const float* S0,
                float* XS, float* Hs, float* HP, float* gates, float* S, float* CS, int64_t ldcs,
                const float* drop, float* Hdrop, void* ws, size_t ws_bytes, hipStream_t stream);
bool seq_bwd_ws_ok(int64_t T, int64_t B, int64_t NT, int64_t H, int64_t L, int64_t W, int64_t D, int64_t ks, int64_t ldcs);
int seq_bwd_ws(int64_t T, const int64_t* bs, int64_t B, int64_t NT, int64_t H, int64_t L, int64_t W, int64_t D,
               const float* const* w_ih_t, const float* const* w_hh_t, const float* wcs_t,
               const float* HP, const float* gates, const float* S, const float* CS, int64_t ldcs,
               const float* dHtop, float* dXS, int64_t ks, float* DGI, float* DGH, float* DCS, float* dh0,
               const float* drop, void* ws, size_t ws_bytes, hipStream_t stream);

// shared host helpers of the two persistent kernels (ppk_seq_persist.hip)
int seq_poison(const int* status, float* const* ptrs, const long long* counts, int n, hipStream_t stream);
int seq_capacity(const void* kernel, int slot);          // slot: 0 / 1 forward fp32 / bf16, 2 / 3 backward
int seq_groups_per_launch(int capacity, int ntiles, int xt, int tpx);
void seq_placement(int ntiles, int& xt, int& tpx);
double seq_row_share(const int* bs, int T, int G, int MT, int pair, int g0, int g1);

bool seq_bwd_persist_ok(int64_t T, int64_t B, int64_t NT, int64_t H, int64_t L, int64_t W, int64_t D, int64_t ks, int64_t ldcs);
size_t seq_bwd_persist_ws_bytes(int64_t B, int64_t H, int64_t L, int64_t W, int64_t D);
int seq_bwd_persist(int64_t T, const int64_t* bs, int64_t B, int64_t NT, int64_t H, int64_t L, int64_t W, int64_t D,
                    const float* const* w_ih_t, const float* const* w_hh_t, const float* wcs_t,
                    const float* HP, const float* gates, const float* S, const float* CS, int64_t ldcs,
                    const float* dHtop, float* dXS, int64_t ks, float* DGI, float* DGH, float* DCS, float* dh0,
                    const float* drop, void* ws, size_t ws_bytes, hipStream_t stream);

// ---- opt-in per-kernel timing (HIP events on the launch stream) -------------
// Off by default; bench.py switches it on for ONE extra step after the timed
// region to price each kernel family against its roofline.
enum ProfKind {
    PK_GEMM_S_XX = 0, PK_GEMM_S_XK, PK_GEMM_S_KX, PK_GEMM_S_KK,      // 64x64 tiles;  A,B layout: K = k-contiguous, X = x-contiguous
    PK_GEMM_L_XX, PK_GEMM_L_XK, PK_GEMM_L_KX, PK_GEMM_L_KK,          // 128x128 tiles
    PK_STACK_FWD, PK_STACK_BWD,
    PK_GRAPHREG_TERMS, PK_GRAPHREG_DZ, PK_SPMM, PK_GRAPHREG_SPARSE,
    PK_GEMM_NT64, PK_GEMM_NTF, PK_GEMM_NT32, PK_GEMM_NTFS, PK_GEMM_TN128, PK_SEQ_FWD, PK_SEQ_BWD,
    PK_COUNT
};
struct ProfScope {
    ProfScope(int kind, double work, hipStream_t stream);   // work: FLOPs or bytes of this launch
    ~ProfScope();
    int idx;
    hipStream_t stream;
};

}  // namespace ppk
