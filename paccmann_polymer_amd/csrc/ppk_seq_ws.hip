// K1 for SMALL AND MEDIUM batches — first of all the reference's own regime, one bundle of <= 21 SMILES per optimizer
// step (experiments/chem_reactions/reactions_vae.py:52-59) — as ONE persistent launch per pass with the weights STATIONARY
// IN REGISTERS (paccmann_chemistry StackGRU reached from paccmann_polymer/models/paccmann_encoder.py:166,168; semantics
// SURVEY.md Appendix B).  Forward (k_seq_fwd_ws), backward (k_seq_bwd_ws), and a looped forward for larger batches
// (k_seq_fwd_wsl).
//
// With few rows the per-step launches and the row-block persistent kernel (ppk_seq_persist.hip) both spend a timestep
// streaming every weight (9.8 MB at H 512, L 2) past one or two 32-row tiles: ~35 us per timestep on 16 CUs.  Here the
// hidden units are dealt 16 to a workgroup (H / 16 workgroups of 512 threads, one per CU) and a workgroup keeps ITS gate
// rows of every matrix — r, z, n of its 16 units: 48 x (ks + (2L-1) H) floats = 307 KB at H 512, L 2 — in the VGPRs of its
// eight waves for the whole launch (150 registers per lane), as A operands of v_mfma_f32_16x16x4_f32:
//     D[unit][row] += W[unit][k] . x[row][k]       (M = 16 gate rows, N = 16 batch rows, K = 4)
// The eight waves split K; a lane's B operands are 16 floats of one activation row, loaded straight from global memory
// into registers (no LDS staging: the activations are 2 KB per row).  Only activations move: per timestep a workgroup
// reads the rows' stack tops and hidden states (<= 32 x 4.4 KB) and writes its 16 units of them.
// The eight K-chunk partial tiles are summed in wave order through LDS (fixed order: bitwise reproducible), the GRU
// cell runs on the sums, and the stack heads are split over the workgroups as in ppk_seq_persist.hip (every workgroup
// multiplies its 16 units with its slice of [W_c; W_s], the row's owner adds the partials in workgroup order).  The
// owner of a row keeps that row's stack in LDS for the whole launch (global S is write-only here).
// A batch of more than 16 sequences is dealt round-robin to independent REPLICAS of <= 16 (32) rows, each with its own
// H / 16 workgroups, counters and partial buffers (struct Rep): one launch while replicas x H / 16 <= CUs.
//
// Order inside a timestep (L >= 2):  W_hh0 h_0(t-1)  |wait heads| stack top of the owned rows |signal x| rest of the rows
//   |wait x| W_ih0 x(t), cell 0 |signal h_0|  W_hh1 h_1(t-1)  |wait h_0| W_ih1 h_0(t), cell 1, heads partial |signal h_1, heads|
// — the products that need nothing new sit in front of the waits, what only later kernels read is stored behind the
// signals.  Three hand-offs per timestep: sc1 write-through stores, every wave drains, barrier, one agent-scope add; one
// lane polls (sc1), barrier, sc1 loads to registers (ws_wait below: why no acquire).
#include "ppk_common.h"
#include "ppk_kernels.h"
#include "ppk_seq_common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

using namespace ppk_seq;

constexpr int MAXT = ppk::SEQP_MAXT;
constexpr int NTH = 512;                       // 8 waves: two per SIMD, 256 registers each
constexpr int WROWS = 32;                      // batch rows of a launch (two 16-row MFMA tiles)
constexpr int STACK_FLOATS = 12288;            // LDS for the owned rows' stacks (two images per row)
constexpr int WS_MAXL = 3;                      // three layers: H <= 256 (the weight registers: 3 (ks + (2L-1) H) / 32 per lane)

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WsArgs {
    int T, B, W, D, V, ldcs, DW, R;
    long long NT;
    const int64_t* tokens;
    const float* table; const float* wx0s;
    const float* w_ih[WS_MAXL]; const float* w_hh[WS_MAXL]; const float* b_ih[WS_MAXL]; const float* b_hh[WS_MAXL];
    const float* wcs_t; const float* bcs; const float* h0; const float* S0;
    float* XS; float* Hs; float* HP; float* gates; float* S; float* CS;
    const float* drop; float* HD;
    float* hpart;                              // [R][nwg][32][ldcs]: head partials, rewritten every step
    int* ctr; int* status;
    long long* dbg;
    int bs[MAXT];
};

__shared__ __align__(16) float w_part[8 * 4 * WROWS * 16];      // K-chunk partials [wave][r, z, n_i, n_h][row][unit]; head-partial staging
__shared__ __align__(16) float w_red[4 * WROWS * 16];           // their sums
__shared__ __align__(16) float w_hown[WS_MAXL * WROWS * 16];    // this workgroup's 16 units of every layer's hidden state
__shared__ __align__(16) float w_stack[STACK_FLOATS];           // owned rows: [row k][image 0 / 1][DW4]
__shared__ __align__(16) float w_act[16 * 64];                  // softmax(3) | tanh(W) of the owned rows (at most 16 per workgroup: H = 32)
__shared__ __align__(16) float w_top[64];                       // its new stack top, zero padded to ks
__shared__ __align__(16) float w_pf[6 * 512];                   // cell operands requested ahead by LDS-DMA: [operand][cell thread][4]
__shared__ int w_off[MAXT + 1];
__shared__ int w_flag;
// PPK_WS_TIMING build (tools/ws_timing.py): thread 0 adds the shader-clock cycles of every phase into LDS slots
#ifdef PPK_WS_TIMING
__shared__ long long w_tacc[16];
__shared__ long long w_tlast;
long long* g_ws_dbg = nullptr;
#define WSTAMP(id) do { if (threadIdx.x == 0) { const long long now_ = __builtin_readcyclecounter(); w_tacc[id] += now_ - w_tlast; w_tlast = now_; } } while (0)
#define WSTAMP_INIT() do { if (threadIdx.x == 0) { for (int k_ = 0; k_ < 16; ++k_) w_tacc[k_] = 0; w_tlast = __builtin_readcyclecounter(); } } while (0)
#define WSTAMP_DUMP(dbg, j) do { if (threadIdx.x == 0 && (dbg)) for (int k_ = 0; k_ < 16; ++k_) (dbg)[(long long)(j) * 16 + k_] = w_tacc[k_]; } while (0)
#else
#define WSTAMP(id) do { } while (0)
#define WSTAMP_INIT() do { } while (0)
#define WSTAMP_DUMP(dbg, j) do { } while (0)
#endif

// Replicas: a batch of more than 16 (32) sequences is dealt round-robin (in length order: every replica gets the same
// length profile) to R independent replicas of H / 16 workgroups; replica `rep` owns the sorted rows rep, rep + R, ...
// Each has its own counters and partial buffers and never talks to another; all hold the same weights.
struct Rep { int rep, R; };
__device__ __forceinline__ int rows_of(int bs_t, Rep r) { const int v = bs_t - r.rep; return v <= 0 ? 0 : (v + r.R - 1) / r.R; }      // active local rows
__device__ __forceinline__ long long prow(long long base, int row, Rep r) { return base + r.rep + (long long)r.R * row; }             // packed row of a local row

__device__ __forceinline__ f32x2 ld2_sc1(const float* base, long long byte_off)
{
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(make_rsrc(base), (int)byte_off, 0, CPOL_SC1));
}
__device__ __forceinline__ float ld1_sc1(const float* base, long long byte_off)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(make_rsrc(base), (int)byte_off, 0, CPOL_SC1));
}

// The wait of these kernels.  One 512-thread workgroup per CU, every handed-off byte written by 16-byte sc1 stores behind
// group_signal (every wave drains, barrier, one agent-scope add) and read by sc1 buffer loads INTO REGISTERS after the
// polling wave's barrier: MI355X_MICROARCH.md lists this form (one workgroup per CU, hipMalloc memory, sc1 poll of the
// counter, sc1 loads of 4 / 8 / 16 bytes) as measured valid without the agent acquire, which costs 1.3 - 1.7 us per wait —
// three per timestep here.  -DPPK_WS_ACQUIRE keeps it.  Polls are PPK_WS_POLL_SLEEP x 64 cycles apart (32 pollers on
// three counter lines: the interval is latency, not contention) and bounded like group_wait's.
#ifndef PPK_WS_POLL_SLEEP
#define PPK_WS_POLL_SLEEP 2
#endif
__device__ __forceinline__ bool ws_wait(int* ctr, int target, int* status, volatile int* flag)
{
    if (threadIdx.x == 0) {
        int ok = 0;
        constexpr int POLLS = SPIN_LIMIT == 1 ? 1 : SPIN_LIMIT * (32 / PPK_WS_POLL_SLEEP);      // ~3 s; the give-up-at-once test build polls once
        for (int it = 0; it < POLLS; ++it) {
            if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
            if ((it & 1023) == 1023 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            __builtin_amdgcn_s_sleep(PPK_WS_POLL_SLEEP);
        }
        if (!ok) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef PPK_WS_ACQUIRE
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        *flag = ok;
    }
    __syncthreads();
    const bool ok = *flag != 0;
    __syncthreads();
    return ok;
}

// K index (inside a wave's chunk of 4 KK values) that MFMA m contracts in k-group g.  With 16-byte loads (KK % 4 == 0)
// the four k-groups of a batch row read ADJACENT 16-byte pieces in one load instruction — 64 contiguous bytes per row and
// instruction (16 line requests per wave instruction instead of 32, each line asked for once per two instructions instead
// of four times): k = 16 x + 4 g + c for m = 4 x + c.  The weight registers use the same map.
template <int N> struct IntK { static constexpr int v = N; };
template <int KK> __device__ __forceinline__ constexpr int wk(int g, int m)
{
    return KK % 4 == 0 ? 16 * (m >> 2) + 4 * g + (m & 3) : g * KK + m;
}
// this lane's KK floats of an activation row (handed-off data: L1-bypassing loads); byte_off = the wave's chunk of the row
template <int KK>
__device__ __forceinline__ void load_b(float (&b)[KK], const float* base, long long byte_off, int g)
{
    if constexpr (KK % 4 == 0) {
#pragma unroll
        for (int x = 0; x < KK / 4; ++x) {
            const f32x4 v = ld4_sc1(base, byte_off + 64 * x + 16 * g);
            b[4 * x] = v[0]; b[4 * x + 1] = v[1]; b[4 * x + 2] = v[2]; b[4 * x + 3] = v[3];
        }
    } else if constexpr (KK == 2) {
        const f32x2 v = ld2_sc1(base, byte_off + 8 * g);
        b[0] = v[0]; b[1] = v[1];
    } else {
        static_assert(KK == 1, "K / 32 is 1, 2 or a multiple of 4");
        b[0] = ld1_sc1(base, byte_off + 4 * g);
    }
}

// acc[rt][0], acc[rt][1], acc[rt][NSEL] += W_q[16 units x K] . X[rows x K]^T over this wave's K chunk (K = 32 KK):
// lane (n = lane & 15, g = lane >> 4) holds k = chunk + g KK + m of unit n (weights) and of batch row 16 rt + n (X)
template <int KK, int NSEL, int NR>
__device__ __forceinline__ void ws_product(f32x4 (&acc)[NR][4], const float (&w)[3][KK], const float* X, long long row0, int bt, int nrt,
                                           int wave, int lane, Rep rp)
{
    const int n = lane & 15, g = lane >> 4;
    constexpr int K = 32 * KK;
    float b0[KK], b1[KK];
    {
        const int r = n < bt ? n : bt - 1;
        load_b<KK>(b0, X, (prow(row0, r, rp) * K + wave * 4 * KK) * 4, g);
    }
    if (NR > 1 && nrt > 1) {
        const int r = 16 + n < bt ? 16 + n : bt - 1;
        load_b<KK>(b1, X, (prow(row0, r, rp) * K + wave * 4 * KK) * 4, g);
    }
#pragma unroll
    for (int m = 0; m < KK; ++m) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0][m], b0[m], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1][m], b0[m], acc[0][1], 0, 0, 0);
        acc[0][NSEL] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2][m], b0[m], acc[0][NSEL], 0, 0, 0);
    }
    if (NR > 1 && nrt > 1) {
#pragma unroll
        for (int m = 0; m < KK; ++m) {
            acc[NR - 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0][m], b1[m], acc[NR - 1][0], 0, 0, 0);
            acc[NR - 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1][m], b1[m], acc[NR - 1][1], 0, 0, 0);
            acc[NR - 1][NSEL] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2][m], b1[m], acc[NR - 1][NSEL], 0, 0, 0);
        }
    }
}

// the three gate tiles of one row tile: acc[0], acc[1], acc[NSEL] += W_q . b
template <int KK, int NSEL>
__device__ __forceinline__ void mma_gates(f32x4 (&acc)[4], const float (&w)[3][KK], const float (&b)[KK])
{
#pragma unroll
    for (int m = 0; m < KK; ++m) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0][m], b[m], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1][m], b[m], acc[1], 0, 0, 0);
        acc[NSEL] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2][m], b[m], acc[NSEL], 0, 0, 0);
    }
}

// this lane's KK weights (K map: wk) of gate row `grow` (a [rows x K] row-major matrix), K chunk of `wave`
template <int KK>
__device__ __forceinline__ void load_w(float (&w)[KK], const float* Wm, long long grow, int wave, int lane)
{
    const float* p = Wm + grow * (32 * KK) + wave * 4 * KK;
#pragma unroll
    for (int m = 0; m < KK; ++m) w[m] = p[wk<KK>(lane >> 4, m)];
}

// ---- stack update of the rows this workgroup owns (rho = j + k nwg), for the stack ENTERING step t ----------------------
// In two parts around the hand-off.  ws_stack_top: heads = the workgroups' partials added in workgroup order + bias ->
// softmax(3) / tanh(W), then ONLY the new stack top (W elements) -> XS, handed to the group.  ws_stack_rest, behind the
// signal: the push / pop / no-op blend of the whole row's image in LDS into its other image, the new row to S and the
// activations to CS (both read by later kernels only) — off the group's critical path.
__device__ __forceinline__ void ws_stack_top(const WsArgs& a, int t, int j, int nwg, int ks, Rep rp)
{
    const int tid = threadIdx.x, bt = rows_of(a.bs[t], rp), ldcs = a.ldcs, W = a.W, DW = a.DW, DW4 = (DW + 3) & ~3, nout = 3 + W;
    const long long base = w_off[t];
    float* hp = w_part;                                // [nwg][ldcs]
    const float* hpart = a.hpart + (long long)rp.rep * nwg * WROWS * ldcs;
    for (int k = 0; j + k * nwg < bt; ++k) {           // workgroup-uniform
        const int rho = j + k * nwg;
        float* act = w_act + 64 * k;
        for (int e = tid; e < nwg * (ldcs >> 2); e += NTH) {
            const int jj = e / (ldcs >> 2), o = (e - jj * (ldcs >> 2)) * 4;
            st4(hp + jj * ldcs + o, ld4_sc1(hpart, (((long long)jj * WROWS + rho) * ldcs + o) * 4));
        }
        __syncthreads();
        if (tid < ldcs) {
            float v = 0.f;
            for (int jj = 0; jj < nwg; ++jj) v += hp[jj * ldcs + tid];
            act[tid] = tid < nout ? v + a.bcs[tid] : 0.f;
        }
        __syncthreads();
        if (tid == 0) {
            const float p0 = act[0], p1 = act[1], p2 = act[2];
            const float m = fmaxf(p0, fmaxf(p1, p2));
            const float e0 = expf(p0 - m), e1 = expf(p1 - m), e2 = expf(p2 - m);
            const float inv = 1.0f / (e0 + e1 + e2);
            act[0] = e0 * inv; act[1] = e1 * inv; act[2] = e2 * inv;
        } else if (tid >= 3 && tid < nout) act[tid] = tanhf(act[tid]);
        __syncthreads();
        if (tid < ks) {
            const float* sp = w_stack + (2 * k + ((t & 1) ^ 1)) * DW4;
            float v = 0.f;
            if (tid < W) v = act[2] * sp[tid] + act[0] * act[3 + tid] + act[1] * (tid + W < DW ? sp[tid + W] : 0.f);
            w_top[tid] = v;
        }
        __syncthreads();
        if (tid < (ks >> 2)) st4_sc1(a.XS, (prow(base, rho, rp) * ks + 4 * tid) * 4, ld4(w_top + 4 * tid));      // handed to the group
    }
}
__device__ __forceinline__ void ws_stack_rest(const WsArgs& a, int t, int j, int nwg, Rep rp)
{
    const int tid = threadIdx.x, bt = rows_of(a.bs[t], rp), ldcs = a.ldcs, W = a.W, DW = a.DW, DW4 = (DW + 3) & ~3, nout = 3 + W;
    const long long base = w_off[t];
    const bool vec = (DW & 3) == 0;
    for (int k = 0; j + k * nwg < bt; ++k) {           // workgroup-uniform
        const int rho = j + k * nwg;
        const float* act = w_act + 64 * k;
        const float a_push = act[0], a_pop = act[1], a_noop = act[2];
        const float* sp = w_stack + (2 * k + ((t & 1) ^ 1)) * DW4;
        float* sn = w_stack + (2 * k + (t & 1)) * DW4;
        float* sg = a.S + prow(base, rho, rp) * DW;
        auto upd = [&](int e) {
            const float up = e < W ? act[3 + e] : sp[e - W];
            const float down = e + W < DW ? sp[e + W] : 0.f;
            return a_noop * sp[e] + a_push * up + a_pop * down;
        };
        if (vec) {
            for (int e4 = tid; e4 < (DW >> 2); e4 += NTH) {
                f32x4 v;
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] = upd(4 * e4 + x);
                st4(sn + 4 * e4, v);
                st4(sg + 4 * e4, v);
            }
        } else {
            for (int e = tid; e < DW; e += NTH) {
                const float v = upd(e);
                sn[e] = v;
                sg[e] = v;
            }
        }
        if (tid < nout) a.CS[prow(base, rho, rp) * ldcs + tid] = act[tid];
    }
}

// ---- heads partial of this workgroup's 16 units of the top layer's hidden state (w_hown) ---------------------------
//   hpart[j][row][o] = sum_c h[row][16 j + c] wcs_t[16 j + c][o]        wave = (row tile, 16-column block of the outputs)
__device__ __forceinline__ void ws_heads_partial(const WsArgs& a, const float* hpart, const float (&wc)[4], const float* hown, int j, int bt, int wave, int lane)
{
    // row tile 0 on waves 4 - 7: waves 0 and 1 are still issuing the cell's write-through stores
    const int n = lane & 15, g = lane >> 4, ot = wave & 3, rt = (wave >> 2) ^ 1;
    if (16 * ot < a.ldcs && 16 * rt < bt) {
        const f32x4 hv = ld4(hown + (16 * rt + n) * 16 + 4 * g);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[m], hv[m], acc, 0, 0, 0);
        if (16 * rt + n < bt) st4_sc1(hpart, (((long long)j * WROWS + 16 * rt + n) * a.ldcs + 16 * ot + 4 * g) * 4, acc);
    }
}

template <int H, int KS, int L, int NR>
__global__ __launch_bounds__(NTH) void k_seq_fwd_ws(const WsArgs a)
{
    constexpr int KH = H / 32, KX = KS / 32, NWG = H / 16;
    const int rep = blockIdx.x / NWG, j = blockIdx.x - rep * NWG;
    const Rep rp{rep, a.R};
    const int Bl = rows_of(a.B, rp);                    // rows of this replica
    const float* hpart = a.hpart + (long long)rep * NWG * WROWS * a.ldcs;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const long long H3 = 3LL * H;
    if (tid == 0) {
        int s = 0;
        for (int t = 0; t < a.T; ++t) { w_off[t] = s; s += a.bs[t]; }
        w_off[a.T] = s;
    }
    // ---- the workgroup's weights: resident in registers from here on
    float wh[L][3][KH], wi0[3][KX], wi[L > 1 ? L - 1 : 1][3][KH], wc[4];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const long long grow = (long long)q * H + 16 * j + n;
        load_w<KX>(wi0[q], a.wx0s, grow, wave, lane);
#pragma unroll
        for (int l = 0; l < L; ++l) load_w<KH>(wh[l][q], a.w_hh[l], grow, wave, lane);
#pragma unroll
        for (int l = 1; l < L; ++l) load_w<KH>(wi[l - 1][q], a.w_ih[l], grow, wave, lane);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) wc[m] = 16 * (wave & 3) < a.ldcs ? a.wcs_t[(long long)(16 * j + 4 * g + m) * a.ldcs + 16 * (wave & 3) + n] : 0.f;
    // own units of the initial hidden state; the owned rows' initial stacks; zero padding of the stack top
    for (int e = tid; e < L * WROWS * 4; e += NTH) {
        const int l = e / (WROWS * 4), row = (e >> 2) & (WROWS - 1), u4 = e & 3;
        st4(w_hown + (l * WROWS + row) * 16 + 4 * u4,
            row < Bl ? ld4(a.h0 + ((long long)l * a.B + prow(0, row, rp)) * H + 16 * j + 4 * u4) : f32x4{0.f, 0.f, 0.f, 0.f});
    }
    {
        const int DW4 = (a.DW + 3) & ~3;
        for (int k = 0; j + k * NWG < Bl; ++k) {
            float* img = w_stack + (2 * k + 1) * DW4;                 // the image step 0 reads
            for (int e = tid; e < a.DW; e += NTH) img[e] = a.S0 ? a.S0[prow(0, j + k * NWG, rp) * a.DW + e] : 0.f;
        }
    }
    if (tid < 64) w_top[tid] = 0.f;
    __syncthreads();
    WSTAMP_INIT();
    int* ctr = a.ctr + (long long)rep * (L + 1) * CTR_STRIDE;
    // "step -1": the heads of the initial hidden state of the top layer
    ws_heads_partial(a, hpart, wc, w_hown + (L - 1) * WROWS * 16, j, Bl, wave, lane);
    group_signal(ctr + (L - 1) * CTR_STRIDE);

    // epilogue of a layer: dump the K-chunk partials, sum them in wave order, GRU cell on the sums (one element per
    // thread: the transcendental functions are the long part), then ONLY the handed-off stores; what later kernels read
    // (HP, gates) is stored by epilogue_tail behind the signal
    float* cellv = w_part;                             // [hp, r, z, n, hn][32][16]: the partials are consumed by then
    auto epilogue = [&](f32x4 (&acc)[NR][4], int l, int t, int bt, int nrt) {
        const long long base = w_off[t];
#pragma unroll
        for (int rt = 0; rt < NR; ++rt)
            if (rt < nrt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) st4(w_part + ((wave * 4 + q) * WROWS + 16 * rt + n) * 16 + 4 * g, acc[rt][q]);
            }
        __syncthreads();
        {
            const int q = tid >> 7, row = (tid >> 2) & 31, u4 = tid & 3;
            if (row < bt) {
                const float* p = w_part + (q * WROWS + row) * 16 + 4 * u4;
                f32x4 v = ld4(p);
#pragma unroll
                for (int w = 1; w < 8; ++w) v += ld4(p + w * 4 * WROWS * 16);
                st4(w_red + (q * WROWS + row) * 16 + 4 * u4, v);
            }
        }
        wait_vmcnt<0>();                               // the LDS-DMA requests of prefetch() have landed
        __syncthreads();
        if ((tid >> 4) < bt) {
            const int e = tid;                         // row * 16 + unit
            const float v0 = w_red[e] + (w_pf[e] + w_pf[3 * 512 + e]);
            const float v1 = w_red[512 + e] + (w_pf[512 + e] + w_pf[4 * 512 + e]);
            const float v2 = w_red[1024 + e] + w_pf[1024 + e];
            const float v3 = w_red[1536 + e] + w_pf[5 * 512 + e];
            const float hp = w_hown[l * 512 + e];
            const float rr = sigmoidf_(v0), zz = sigmoidf_(v1);
            const float nn = tanhf(v2 + rr * v3);
            w_hown[l * 512 + e] = (1.0f - zz) * nn + zz * hp;
            cellv[e] = hp; cellv[512 + e] = rr; cellv[1024 + e] = zz; cellv[1536 + e] = nn; cellv[2048 + e] = v3;
        }
        __syncthreads();
        if (tid < 4 * WROWS && (tid >> 2) < bt) {
            const long long pr = (long long)l * a.NT + prow(base, tid >> 2, rp);
            const int u = 16 * j + 4 * (tid & 3);
            const f32x4 ho = ld4(w_hown + l * 512 + 4 * tid);
            st4_sc1(a.Hs, (pr * H + u) * 4, ho);                                             // handed to the group
            if (a.drop && l < L - 1) st4_sc1(a.HD, (pr * H + u) * 4, ho * ld4(a.drop + pr * H + u));
        }
    };
    auto epilogue_tail = [&](int l, int t, int bt) {
        if (tid < 4 * WROWS && (tid >> 2) < bt) {
            const long long pr = (long long)l * a.NT + prow(w_off[t], tid >> 2, rp);
            const int u = 16 * j + 4 * (tid & 3);
            st4(a.HP + pr * H + u, ld4(cellv + 4 * tid));
            float* gt = a.gates + pr * 4 * H + u;
            st4(gt, ld4(cellv + 512 + 4 * tid)); st4(gt + H, ld4(cellv + 1024 + 4 * tid));
            st4(gt + 2 * H, ld4(cellv + 1536 + 4 * tid)); st4(gt + 3 * H, ld4(cellv + 2048 + 4 * tid));
        }
    };
    // operands of the cell threads that depend on nothing handed off (per-token table row or input bias, recurrent bias):
    // requested before the wait by LDS-DMA, so that no registers are held across the products
    auto prefetch = [&](int l, int t, int bt) {
        if (tid < 4 * WROWS && (tid >> 2) < bt) {
            float* dst = w_pf + 256 * wave;
            const int row = tid >> 2, u = 16 * j + 4 * (tid & 3);
            const float* tp;
            if (l == 0) {
                const long long tok = a.tokens[prow(w_off[t], row, rp)];
                tp = a.table + (tok < 0 ? 0 : (tok < a.V ? tok : a.V - 1)) * H3 + u;
            } else tp = a.b_ih[l] + u;
            const float* bp = a.b_hh[l] + u;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                __builtin_amdgcn_global_load_lds(tp + q * H, PPK_LDS_PTR(dst + q * 512), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(bp + q * H, PPK_LDS_PTR(dst + (3 + q) * 512), 16, 0, 0);
            }
        }
    };

    // ---- two layers, at most 16 rows, no dropout: the operands of the products that need nothing new never wait for
    // memory.  W_hh0 . h_0(t) (for step t + 1) reuses the registers that fed W_ih1 . h_0(t); the rows of h_1(t-1) are
    // requested together with the stack tops and used behind the next signal.  (With two row tiles the extra 32 - 48
    // registers spill; with dropout layer 1 reads the dropped-out copy, not h_0.)
    if constexpr (L == 2 && NR == 1) {
        if (!a.drop) {
            f32x4 acc[1][4];
            float bx[KX], bh[KH], b[KH];
            auto zero1 = [&]() {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[0][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            };
            const long long lo = (wave * 4 * KH) * 4;                       // this wave's chunk of a K = H row, bytes
            zero1();
            load_b<KH>(b, a.h0, prow(0, n < Bl ? n : Bl - 1, rp) * H * 4 + lo, g);
            mma_gates<KH, 3>(acc[0], wh[0], b);
            for (int t = 0; t < a.T; ++t) {
                const int bt = rows_of(a.bs[t], rp);
                if (bt == 0) break;                                  // this replica's longest sequence has ended
                const long long base = w_off[t], pbase = t > 0 ? w_off[t - 1] : 0;
                const int rx = n < bt ? n : bt - 1;
                WSTAMP(0);
                if (!ws_wait(ctr + CTR_STRIDE, NWG * (t + 1), a.status, &w_flag)) return;
                WSTAMP(1);
                ws_stack_top(a, t, j, NWG, KS, rp);
                WSTAMP(2);
                group_signal(ctr + 2 * CTR_STRIDE);
                WSTAMP(3);
                ws_stack_rest(a, t, j, NWG, rp);
                prefetch(0, t, bt);
                if (!ws_wait(ctr + 2 * CTR_STRIDE, NWG * (t + 1), a.status, &w_flag)) return;
                WSTAMP(4);
                load_b<KX>(bx, a.XS, (prow(base, rx, rp) * KS + wave * 4 * KX) * 4, g);
                if (t > 0) load_b<KH>(bh, a.Hs + a.NT * H, prow(pbase, rx, rp) * H * 4 + lo, g);
                else load_b<KH>(bh, a.h0 + (long long)a.B * H, prow(0, rx, rp) * H * 4 + lo, g);
                mma_gates<KX, 2>(acc[0], wi0, bx);
                WSTAMP(5);
                epilogue(acc, 0, t, bt, 1);
                WSTAMP(6);
                group_signal(ctr);
                WSTAMP(7);
                epilogue_tail(0, t, bt);
                zero1();
                mma_gates<KH, 3>(acc[0], wh[1], bh);
                prefetch(1, t, bt);
                WSTAMP(8);
                if (!ws_wait(ctr, NWG * (t + 1), a.status, &w_flag)) return;
                WSTAMP(9);
                load_b<KH>(b, a.Hs, prow(base, rx, rp) * H * 4 + lo, g);
                mma_gates<KH, 2>(acc[0], wi[0], b);
                WSTAMP(10);
                epilogue(acc, 1, t, bt, 1);
                WSTAMP(11);
                ws_heads_partial(a, hpart, wc, w_hown + WROWS * 16, j, bt, wave, lane);
                WSTAMP(12);
                group_signal(ctr + CTR_STRIDE);
                WSTAMP(13);
                epilogue_tail(1, t, bt);
                zero1();
                if (t + 1 < a.T) mma_gates<KH, 3>(acc[0], wh[0], b);     // rows of step t + 1 are a prefix of this step's
            }
            WSTAMP_DUMP(a.dbg, j);
            return;
        }
    }

    for (int t = 0; t < a.T; ++t) {
        const int bt = rows_of(a.bs[t], rp), nrt = (bt + 15) >> 4;
        if (bt == 0) break;                                          // this replica's longest sequence has ended
        const long long base = w_off[t], pbase = t > 0 ? w_off[t - 1] : 0;
        f32x4 acc[NR][4];
        auto zero = [&]() {
#pragma unroll
            for (int rt = 0; rt < NR; ++rt)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[rt][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        };
        auto hidden_product = [&](int l) {             // W_hh[l] . h_l(t-1): every unit of it was handed off one phase ago
            if (t > 0) ws_product<KH, 3, NR>(acc, wh[l], a.Hs + (long long)l * a.NT * H, pbase, bt, nrt, wave, lane, rp);
            else ws_product<KH, 3, NR>(acc, wh[l], a.h0 + (long long)l * a.B * H, 0, bt, nrt, wave, lane, rp);
        };
        zero();
        if constexpr (L > 1) hidden_product(0);
        WSTAMP(0);
        // the stack entering this step: heads of the top layer's hidden state of the step before
        if (!ws_wait(ctr + (L - 1) * CTR_STRIDE, NWG * (t + 1), a.status, &w_flag)) return;
        WSTAMP(1);
        ws_stack_top(a, t, j, NWG, KS, rp);
        WSTAMP(2);
        group_signal(ctr + L * CTR_STRIDE);
        WSTAMP(3);
        ws_stack_rest(a, t, j, NWG, rp);
        if constexpr (L == 1) hidden_product(0);
        // layer 0: + W_ih0[:, E:] . stack top, + table[token]
        prefetch(0, t, bt);
        if (!ws_wait(ctr + L * CTR_STRIDE, NWG * (t + 1), a.status, &w_flag)) return;
        WSTAMP(4);
        ws_product<KX, 2, NR>(acc, wi0, a.XS, base, bt, nrt, wave, lane, rp);
        WSTAMP(5);
        epilogue(acc, 0, t, bt, nrt);
        if constexpr (L == 1) ws_heads_partial(a, hpart, wc, w_hown, j, bt, wave, lane);
        WSTAMP(6);
        group_signal(ctr);
        WSTAMP(7);
        epilogue_tail(0, t, bt);
#pragma unroll
        for (int l = 1; l < L; ++l) {
            zero();
            hidden_product(l);
            prefetch(l, t, bt);
            WSTAMP(8);
            if (!ws_wait(ctr + (l - 1) * CTR_STRIDE, NWG * (t + 1), a.status, &w_flag)) return;
            WSTAMP(9);
            ws_product<KH, 2, NR>(acc, wi[l - 1], (a.drop ? a.HD : a.Hs) + (long long)(l - 1) * a.NT * H, base, bt, nrt, wave, lane, rp);
            WSTAMP(10);
            epilogue(acc, l, t, bt, nrt);
            WSTAMP(11);
            if (l == L - 1) ws_heads_partial(a, hpart, wc, w_hown + l * WROWS * 16, j, bt, wave, lane);
            WSTAMP(12);
            group_signal(ctr + l * CTR_STRIDE);
            WSTAMP(13);
            epilogue_tail(l, t, bt);
        }
    }
    WSTAMP_DUMP(a.dbg, j);
}

// =====================================================================================================================
// Looped forward: replicas of up to 256 rows (8 per workgroup's stack ownership), the rows of a replica walked 32 at a
// time inside every layer phase — the weights stay in registers, three hand-offs per timestep for the whole replica.
// B = 2048 at H 512 is 8 replicas x 256 rows on 256 CUs: no weight is streamed at all.  The owned rows' stacks do not fit
// LDS here: the previous row comes from global S (this workgroup's own stores of the step before).
constexpr int LROWS = 256;                     // rows of a replica
constexpr int LOWN = 8;                        // stack rows a workgroup owns (rows of a replica <= 8 x H / 16)
__shared__ __align__(16) float l_hown[WS_MAXL * LROWS * 16];    // this workgroup's 16 units of every layer's hidden state
__shared__ __align__(16) float l_sp[LOWN * 128];                // first 2W elements of the owned rows' previous stack rows
__shared__ __align__(16) float l_act[LOWN * 64];
__shared__ __align__(16) float l_top[LOWN * 64];

template <int H, int KS, int L>
__global__ __launch_bounds__(NTH) void k_seq_fwd_wsl(const WsArgs a)
{
    constexpr int KH = H / 32, KX = KS / 32, NWG = H / 16, NR = 2;
    const int rep = blockIdx.x / NWG, j = blockIdx.x - rep * NWG;
    const Rep rp{rep, a.R};
    const int Bl = rows_of(a.B, rp);
    float* hpart = a.hpart + (long long)rep * NWG * LROWS * a.ldcs;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const long long H3 = 3LL * H;
    const int ldcs = a.ldcs, W = a.W, DW = a.DW, nout = 3 + W;
    if (tid == 0) {
        int s = 0;
        for (int t = 0; t < a.T; ++t) { w_off[t] = s; s += a.bs[t]; }
        w_off[a.T] = s;
    }
    float wh[L][3][KH], wi0[3][KX], wi[L > 1 ? L - 1 : 1][3][KH], wc[4];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const long long grow = (long long)q * H + 16 * j + n;
        load_w<KX>(wi0[q], a.wx0s, grow, wave, lane);
#pragma unroll
        for (int l = 0; l < L; ++l) load_w<KH>(wh[l][q], a.w_hh[l], grow, wave, lane);
#pragma unroll
        for (int l = 1; l < L; ++l) load_w<KH>(wi[l - 1][q], a.w_ih[l], grow, wave, lane);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) wc[m] = 16 * (wave & 3) < ldcs ? a.wcs_t[(long long)(16 * j + 4 * g + m) * ldcs + 16 * (wave & 3) + n] : 0.f;
    for (int e = tid; e < L * LROWS * 4; e += NTH) {
        const int l = e / (LROWS * 4), row = (e >> 2) & (LROWS - 1), u4 = e & 3;
        st4(l_hown + (l * LROWS + row) * 16 + 4 * u4,
            row < Bl ? ld4(a.h0 + ((long long)l * a.B + prow(0, row, rp)) * H + 16 * j + 4 * u4) : f32x4{0.f, 0.f, 0.f, 0.f});
    }
    __syncthreads();
    int* ctr = a.ctr + (long long)rep * (L + 1) * CTR_STRIDE;

    // heads partial of the rows r0 .. r0 + 31 (w_hown layout [row][16 units]); wave = (16-column block of the outputs, row tile)
    auto heads_pair = [&](int l, int r0, int bt) {
        const int ot = wave & 3, rt = (wave >> 2) ^ 1, row = r0 + 16 * rt + n;
        if (16 * ot < ldcs && r0 + 16 * rt < bt) {
            const f32x4 hv = ld4(l_hown + (l * LROWS + row) * 16 + 4 * g);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[m], hv[m], acc, 0, 0, 0);
            if (row < bt) st4_sc1(hpart, (((long long)j * LROWS + row) * ldcs + 16 * ot + 4 * g) * 4, acc);
        }
    };
    for (int r0 = 0; r0 < Bl; r0 += 32) heads_pair(L - 1, r0, Bl);
    group_signal(ctr + (L - 1) * CTR_STRIDE);

    float* cellv = w_part;
    // rows r0 .. r0 + 31 of layer l: partial dump, sum in wave order, GRU cell, stores
    auto epilogue = [&](f32x4 (&acc)[NR][4], int l, int t, int r0, int bt, int nrt) {
        const long long base = w_off[t];
#pragma unroll
        for (int rt = 0; rt < NR; ++rt)
            if (rt < nrt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) st4(w_part + ((wave * 4 + q) * WROWS + 16 * rt + n) * 16 + 4 * g, acc[rt][q]);
            }
        __syncthreads();
        {
            const int q = tid >> 7, row = (tid >> 2) & 31, u4 = tid & 3;
            if (r0 + row < bt) {
                const float* p = w_part + (q * WROWS + row) * 16 + 4 * u4;
                f32x4 v = ld4(p);
#pragma unroll
                for (int w = 1; w < 8; ++w) v += ld4(p + w * 4 * WROWS * 16);
                st4(w_red + (q * WROWS + row) * 16 + 4 * u4, v);
            }
        }
        wait_vmcnt<0>();
        __syncthreads();
        if (r0 + (tid >> 4) < bt) {
            const int e = tid;
            const float v0 = w_red[e] + (w_pf[e] + w_pf[3 * 512 + e]);
            const float v1 = w_red[512 + e] + (w_pf[512 + e] + w_pf[4 * 512 + e]);
            const float v2 = w_red[1024 + e] + w_pf[1024 + e];
            const float v3 = w_red[1536 + e] + w_pf[5 * 512 + e];
            float* hl = l_hown + (l * LROWS + r0) * 16 + e;
            const float hp = *hl;
            const float rr = sigmoidf_(v0), zz = sigmoidf_(v1);
            const float nn = tanhf(v2 + rr * v3);
            *hl = (1.0f - zz) * nn + zz * hp;
            cellv[e] = hp; cellv[512 + e] = rr; cellv[1024 + e] = zz; cellv[1536 + e] = nn; cellv[2048 + e] = v3;
        }
        __syncthreads();
        if (tid < 4 * WROWS && r0 + (tid >> 2) < bt) {
            const long long pr = (long long)l * a.NT + prow(base, r0 + (tid >> 2), rp);
            const int u = 16 * j + 4 * (tid & 3);
            const f32x4 ho = ld4(l_hown + (l * LROWS + r0) * 16 + 4 * tid);
            st4_sc1(a.Hs, (pr * H + u) * 4, ho);                                             // handed to the group
            if (a.drop && l < L - 1) st4_sc1(a.HD, (pr * H + u) * 4, ho * ld4(a.drop + pr * H + u));
            st4(a.HP + pr * H + u, ld4(cellv + 4 * tid));
            float* gt = a.gates + pr * 4 * H + u;
            st4(gt, ld4(cellv + 512 + 4 * tid)); st4(gt + H, ld4(cellv + 1024 + 4 * tid));
            st4(gt + 2 * H, ld4(cellv + 1536 + 4 * tid)); st4(gt + 3 * H, ld4(cellv + 2048 + 4 * tid));
        }
        __syncthreads();                               // cellv / w_pf are free for the next 32 rows
    };
    auto prefetch = [&](int l, int t, int r0, int bt) {
        if (tid < 4 * WROWS && r0 + (tid >> 2) < bt) {
            float* dst = w_pf + 256 * wave;
            const int row = r0 + (tid >> 2), u = 16 * j + 4 * (tid & 3);
            const float* tp;
            if (l == 0) {
                const long long tok = a.tokens[prow(w_off[t], row, rp)];
                tp = a.table + (tok < 0 ? 0 : (tok < a.V ? tok : a.V - 1)) * H3 + u;
            } else tp = a.b_ih[l] + u;
            const float* bp = a.b_hh[l] + u;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                __builtin_amdgcn_global_load_lds(tp + q * H, PPK_LDS_PTR(dst + q * 512), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(bp + q * H, PPK_LDS_PTR(dst + (3 + q) * 512), 16, 0, 0);
            }
        }
    };
    // acc += W . X[rows r0 .. r0 + 31]: both row tiles' operands requested before the first MFMA
    auto product = [&](f32x4 (&acc)[NR][4], auto& w, const float* X, long long rowbase, int r0, int bt, int nrt, auto kk, auto nsel) {
        constexpr int KK = decltype(kk)::v, NSEL = decltype(nsel)::v;
        constexpr int K = 32 * KK;
        float b0[KK], b1[KK];
        {
            const int r = r0 + n < bt ? r0 + n : bt - 1;
            load_b<KK>(b0, X, (prow(rowbase, r, rp) * K + wave * 4 * KK) * 4, g);
        }
        if (nrt > 1) {
            const int r = r0 + 16 + n < bt ? r0 + 16 + n : bt - 1;
            load_b<KK>(b1, X, (prow(rowbase, r, rp) * K + wave * 4 * KK) * 4, g);
        }
        mma_gates<KK, NSEL>(acc[0], w, b0);
        if (nrt > 1) mma_gates<KK, NSEL>(acc[1], w, b1);
    };

    for (int t = 0; t < a.T; ++t) {
        const int bt = rows_of(a.bs[t], rp);
        if (bt == 0) break;
        const long long base = w_off[t], pbase = t > 0 ? w_off[t - 1] : 0;
        // ---- stack entering this step: the owned rows rho = j + k NWG (k < 8)
        {
            const int k = tid >> 6, e = tid & 63, rho = j + k * NWG;
            // first 2W elements of the previous rows: requested ahead of the wait (this workgroup's own stores)
            if (rho < bt) {
                const float* src = t > 0 ? a.S + prow(pbase, rho, rp) * DW : (a.S0 ? a.S0 + prow(0, rho, rp) * DW : nullptr);
                const int lim = 2 * W < DW ? 2 * W : DW;
                l_sp[k * 128 + e] = src && e < lim ? src[e] : 0.f;
                l_sp[k * 128 + 64 + e] = src && e + 64 < lim ? src[e + 64] : 0.f;
            }
        }
        if (!ws_wait(ctr + (L - 1) * CTR_STRIDE, NWG * (t + 1), a.status, &w_flag)) return;
        {
            // heads of the owned rows: thread = (row k, group of workgroups grp, 4 columns o4); partials added in workgroup order
            const int k = tid >> 6, grp = (tid >> 4) & 3, o4 = tid & 15, rho = j + k * NWG;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (rho < bt && 4 * o4 < ldcs) {
                constexpr int NP = (NWG + 3) / 4;
                f32x4 p[NP];
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const int jj = grp + 4 * i < NWG ? grp + 4 * i : NWG - 1;
                    p[i] = ld4_sc1(hpart, (((long long)jj * LROWS + rho) * ldcs + 4 * o4) * 4);
                }
#pragma unroll
                for (int i = 0; i < NP; ++i) if (grp + 4 * i < NWG) v += p[i];
            }
            st4(w_part + ((grp * LOWN + k) * 64 + 4 * o4), v);
            __syncthreads();
            const int o = tid & 63;
            if (rho < bt) {
                float s4 = 0.f;
                if (o < nout) s4 = (((w_part[(0 * LOWN + k) * 64 + o] + w_part[(1 * LOWN + k) * 64 + o]) + w_part[(2 * LOWN + k) * 64 + o]) + w_part[(3 * LOWN + k) * 64 + o]) + a.bcs[o];
                l_act[k * 64 + o] = s4;
            }
            __syncthreads();
            if (rho < bt) {
                float* act = l_act + k * 64;
                if (o == 0) {
                    const float p0 = act[0], p1 = act[1], p2 = act[2];
                    const float m = fmaxf(p0, fmaxf(p1, p2));
                    const float e0 = expf(p0 - m), e1 = expf(p1 - m), e2 = expf(p2 - m);
                    const float inv = 1.0f / (e0 + e1 + e2);
                    act[0] = e0 * inv; act[1] = e1 * inv; act[2] = e2 * inv;
                } else if (o >= 3 && o < nout) act[o] = tanhf(act[o]);
            }
            __syncthreads();
            if (rho < bt) {
                const float* act = l_act + k * 64;
                const float* sp = l_sp + k * 128;
                float tv = 0.f;
                if (o < W) tv = act[2] * sp[o] + act[0] * act[3 + o] + act[1] * (o + W < DW ? sp[o + W] : 0.f);
                l_top[k * 64 + o] = tv;
            }
            __syncthreads();
            if (tid < LOWN * 16) {
                const int k2 = tid >> 4, x4 = tid & 15, rho2 = j + k2 * NWG;
                if (rho2 < bt && 4 * x4 < KS) st4_sc1(a.XS, (prow(base, rho2, rp) * KS + 4 * x4) * 4, ld4(l_top + k2 * 64 + 4 * x4));
            }
        }
        group_signal(ctr + L * CTR_STRIDE);
        {
            // behind the signal: the whole rows (one wave per owned row), S and CS for the backward
            const int k = wave, rho = j + k * NWG;
            if (rho < bt) {
                const float* act = l_act + k * 64;
                const float a_push = act[0], a_pop = act[1], a_noop = act[2];
                const float* sp = t > 0 ? a.S + prow(pbase, rho, rp) * DW : (a.S0 ? a.S0 + prow(0, rho, rp) * DW : nullptr);
                float* sn = a.S + prow(base, rho, rp) * DW;
                for (int e0 = 0; e0 < DW; e0 += 64 * 8) {
                    float c[8], u[8], d[8];
#pragma unroll
                    for (int x = 0; x < 8; ++x) {
                        const int e = e0 + 64 * x + lane;
                        c[x] = sp && e < DW ? sp[e] : 0.f;
                        u[x] = e < W ? act[3 + (e < W ? e : 0)] : (sp && e < DW ? sp[e - W] : 0.f);
                        d[x] = sp && e + W < DW ? sp[e + W] : 0.f;
                    }
#pragma unroll
                    for (int x = 0; x < 8; ++x) {
                        const int e = e0 + 64 * x + lane;
                        if (e < DW) sn[e] = a_noop * c[x] + a_push * u[x] + a_pop * d[x];
                    }
                }
                if (lane < nout) a.CS[prow(base, rho, rp) * ldcs + lane] = act[lane];
            }
        }
        // ---- layers
#pragma unroll
        for (int l = 0; l < L; ++l) {
            if (!ws_wait(ctr + (l == 0 ? L : l - 1) * CTR_STRIDE, NWG * (t + 1), a.status, &w_flag)) return;
            for (int r0 = 0; r0 < bt; r0 += 32) {
                const int nrt = bt - r0 > 16 ? 2 : 1;
                prefetch(l, t, r0, bt);
                f32x4 acc[NR][4];
#pragma unroll
                for (int rt = 0; rt < NR; ++rt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[rt][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (t > 0) product(acc, wh[l], a.Hs + (long long)l * a.NT * H, pbase, r0, bt, nrt, IntK<KH>{}, IntK<3>{});
                else product(acc, wh[l], a.h0 + (long long)l * a.B * H, 0, r0, bt, nrt, IntK<KH>{}, IntK<3>{});
                if (l == 0) product(acc, wi0, a.XS, base, r0, bt, nrt, IntK<KX>{}, IntK<2>{});
                else product(acc, wi[l > 0 ? l - 1 : 0], (a.drop ? a.HD : a.Hs) + (long long)(l - 1) * a.NT * H, base, r0, bt, nrt, IntK<KH>{}, IntK<2>{});
                epilogue(acc, l, t, r0, bt, nrt);
                if (l == L - 1) heads_pair(l, r0, bt);
            }
            group_signal(ctr + l * CTR_STRIDE);
        }
    }
}

// [3+W, H] -> [H, ldcs] (zero padded)
__global__ __launch_bounds__(256) void k_ws_wcs_transpose(const float* __restrict__ wcs, int nout, int H, int ldcs, float* __restrict__ out)
{
    const long long total = (long long)H * ldcs;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int k = (int)(e / ldcs), o = (int)(e - (long long)k * ldcs);
        out[e] = o < nout ? wcs[(long long)o * H + k] : 0.f;
    }
}


// =====================================================================================================================
// Backward (the autograd graph the reference walks at experiments/chem_reactions/reactions_vae.py:94-95, loss.backward()),
// same decomposition: workgroup j owns hidden units 16 j .. 16 j + 15 of the OUTPUT of every product and keeps its rows of
// the TRANSPOSED matrices in registers: w_ih_t[l >= 1][u][3H], w_hh_t[l][u][3H] (48 registers each at H 512), its rows of
// wcs_t and its 48 columns of the stack columns of W_ih0 (for the dXS partial).  Per step t (descending):
//   TOP     accD = DCS_{t+1} . wcs_t[u]                        |wait stack backward of t + 1|
//   l = L-1 .. 0:  gate backward of layer l on accD + carry (+ dHtop): dgi_l, dgh_l handed to the group
//           l >= 1: |wait dgi_l| accD = dgi_l . W_ih[l] (-> dh_{l-1}), accC = dgh_l . W_hh[l] (-> carry_l)
//   dXS partial over the workgroup's own 48 gate columns |signal| ; the owner of a row adds the partials, runs the
//   stack-update backward on the row's gradient image in LDS (it never leaves the CU) and hands DCS_t to the group;
//   carry_0 += dgh_0 . W_hh[0] runs behind that hand-off.
// K order of the 3H contractions: [gate][unit], a wave takes units [w H/8, (w+1) H/8) of every gate, so the r and z
// thirds (dgi == dgh) are loaded once.
struct WsBwdArgs {
    int T, B, W, D, ldcs, DW, R;
    long long NT;
    const float* w_ih_t[WS_MAXL]; const float* w_hh_t[WS_MAXL]; const float* wcs_t;
    const float* HP; const float* gates; const float* S; const float* CS; const float* dHtop; const float* drop;
    float* dXS; float* DGI; float* DGH; float* DCS; float* dh0;
    float* xpart;                              // [R][nwg][32][ks]: dXS partials, rewritten every step
    int* ctr; int* status;
    long long* dbg;
    int bs[MAXT];
};

constexpr int BSTACK_FLOATS = 15360;           // owned rows: two images of the stack gradient + the previous stack row
__shared__ __align__(16) float w_bpart[8 * 2 * WROWS * 16];      // K-chunk partials [wave][accD, accC][row][unit]; dXS-partial staging
__shared__ __align__(16) float w_carry[WS_MAXL * WROWS * 16];   // dh z of this workgroup's units (+ dgh . W_hh): the recurrent gradient
__shared__ __align__(16) float w_g0[WROWS * 48];                // dgi_0 of this workgroup's units: [row][gate][unit]
__shared__ __align__(16) float w_bstack[BSTACK_FLOATS];
__shared__ __align__(16) float w_dc[64];
__shared__ float w_rsum[(16 * 8 + 2) * 4];                    // dot-product partials: [owned row][wave][3] before the wait, [wave 0 / 1][3] behind it

template <int KK>
__device__ __forceinline__ void mma_acc(f32x4& acc, const float (&w)[KK], const float (&b)[KK])
{
#pragma unroll
    for (int m = 0; m < KK; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[m], b[m], acc, 0, 0, 0);
}

// stack-update backward of the rows this workgroup owns at step t (k-th step of the launch).  Only the first W
// elements of the row's incoming gradient change with dXS (the stack top fed layer 0), so everything else — the three dot
// products over the elements i >= W and the outgoing gradient of the elements i >= 2W — is computed BEFORE the wait for
// the group's dXS partials, next to the LDS-DMA requests for the previous stack rows; behind the wait only 2W elements
// and the DCS row remain.
__device__ __forceinline__ void ws_stack_bwd_pre(const WsBwdArgs& a, int t, int kstep, int j, int nwg, Rep rp)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bt = rows_of(a.bs[t], rp), ldcs = a.ldcs, W = a.W, DW = a.DW, DW4 = (DW + 3) & ~3, nout = 3 + W, n4 = DW >> 2;
    const long long base = w_off[t], pbase = t > 0 ? w_off[t - 1] : 0;
    const bool vec = (DW & 3) == 0;
    const int par = kstep & 1;
    // previous stack rows of the owned rows (forward output: plain memory) and their saved activations
    for (int k = 0; j + k * nwg < bt; ++k) {
        float* spl = w_bstack + (3 * k + 2) * DW4;
        const float* src = a.S + prow(pbase, j + k * nwg, rp) * DW;
        if (t == 0) { for (int e = tid; e < DW; e += NTH) spl[e] = 0.f; }
        else if (vec) {
            for (int pi = wave; pi * 64 < n4; pi += 8) {
                const int e = pi * 64 + lane;
                if (e < n4) __builtin_amdgcn_global_load_lds(src + 4 * e, PPK_LDS_PTR(spl + pi * 256), 16, 0, 0);
            }
        } else { for (int e = tid; e < DW; e += NTH) spl[e] = src[e]; }
        if (tid < ldcs) w_act[64 * k + tid] = tid < nout ? a.CS[prow(base, j + k * nwg, rp) * ldcs + tid] : 0.f;
    }
    wait_vmcnt<0>();
    __syncthreads();
    for (int k = 0; j + k * nwg < bt; ++k) {
        const float* gin = w_bstack + (3 * k + (par ^ 1)) * DW4;
        float* gout = w_bstack + (3 * k + par) * DW4;
        const float* sp = w_bstack + (3 * k + 2) * DW4;
        const float* act = w_act + 64 * k;
        const float a_push = act[0], a_pop = act[1], a_noop = act[2];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int i = W + tid; i < DW; i += NTH) {
            const float gv = gin[i];
            const float down = i + W < DW ? sp[i + W] : 0.f;
            s0 = fmaf(gv, sp[i], s0);
            s1 = fmaf(gv, sp[i - W], s1);
            s2 = fmaf(gv, down, s2);
            if (i >= 2 * W) {
                float o = a_noop * gv;                 // gradient wrt the previous stack
                if (i + W < DW) o = fmaf(a_push, gin[i + W], o);
                o = fmaf(a_pop, gin[i - W], o);
                gout[i] = o;
            }
        }
        s0 = ppk_wave_sum(s0); s1 = ppk_wave_sum(s1); s2 = ppk_wave_sum(s2);
        if (lane == 0) { float* r = w_rsum + (k * 8 + wave) * 4; r[0] = s0; r[1] = s1; r[2] = s2; }
    }
}
__device__ __forceinline__ void ws_stack_bwd_post(const WsBwdArgs& a, int t, int kstep, int j, int nwg, int ks, Rep rp)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bt = rows_of(a.bs[t], rp), ldcs = a.ldcs, W = a.W, DW = a.DW, DW4 = (DW + 3) & ~3, nout = 3 + W;
    const long long base = w_off[t];
    const int par = kstep & 1;
    float* xp = w_bpart;                                // [nwg][ks]
    const float* xpart = a.xpart + (long long)rp.rep * nwg * WROWS * ks;
    for (int k = 0; j + k * nwg < bt; ++k) {           // workgroup-uniform
        const int rho = j + k * nwg;
        float* gin = w_bstack + (3 * k + (par ^ 1)) * DW4;
        float* gout = w_bstack + (3 * k + par) * DW4;
        const float* sp = w_bstack + (3 * k + 2) * DW4;
        const float* act = w_act + 64 * k;
        for (int e = tid; e < nwg * (ks >> 2); e += NTH) {
            const int jj = e / (ks >> 2), o = (e - jj * (ks >> 2)) * 4;
            st4(xp + jj * ks + o, ld4_sc1(xpart, (((long long)jj * WROWS + rho) * ks + o) * 4));
        }
        __syncthreads();
        if (tid < ks) {
            float v = 0.f;
            for (int jj = 0; jj < nwg; ++jj) v += xp[jj * ks + tid];
            a.dXS[prow(base, rho, rp) * ks + tid] = v;
            if (tid < W) gin[tid] += v;                // the stack top fed layer 0
        }
        __syncthreads();
        const float a_push = act[0], a_pop = act[1], a_noop = act[2];
        if (tid < 128) {                               // the 2W <= 128 elements that depend on dXS (waves 0 and 1)
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            const int i = tid;
            if (i < 2 * W && i < DW) {
                const float gv = gin[i];
                if (i < W) {
                    const float down = i + W < DW ? sp[i + W] : 0.f;
                    s0 = gv * sp[i];
                    s1 = gv * act[3 + i];
                    s2 = gv * down;
                }
                float o = a_noop * gv;
                if (i + W < DW) o = fmaf(a_push, gin[i + W], o);
                if (i >= W) o = fmaf(a_pop, gin[i - W], o);
                gout[i] = o;
            }
            s0 = ppk_wave_sum(s0); s1 = ppk_wave_sum(s1); s2 = ppk_wave_sum(s2);
            if (lane == 0) { float* r = w_rsum + (16 * 8 + wave) * 4; r[0] = s0; r[1] = s1; r[2] = s2; }
        }
        __syncthreads();
        if (tid < ldcs) {
            float t3[3];
            const float* r = w_rsum + k * 8 * 4;
            const float* h = w_rsum + 16 * 8 * 4;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                t3[q] = (((r[q] + r[4 + q]) + (r[8 + q] + r[12 + q])) + ((r[16 + q] + r[20 + q]) + (r[24 + q] + r[28 + q]))) + (h[q] + h[4 + q]);
            const float s_noop = t3[0], s_push = t3[1], s_pop = t3[2];
            const float dot = a_push * s_push + a_pop * s_pop + a_noop * s_noop;
            float v = 0.f;
            if (tid == 0) v = a_push * (s_push - dot);
            else if (tid == 1) v = a_pop * (s_pop - dot);
            else if (tid == 2) v = a_noop * (s_noop - dot);
            else if (tid < nout) { const float sv = act[tid]; v = a_push * gin[tid - 3] * (1.0f - sv * sv); }
            w_dc[tid] = v;
        }
        __syncthreads();
        if (tid < (ldcs >> 2)) st4_sc1(a.DCS, (prow(base, rho, rp) * ldcs + 4 * tid) * 4, ld4(w_dc + 4 * tid));      // handed to the group
    }
}

template <int H, int KS, int L, int NR>
__global__ __launch_bounds__(NTH) void k_seq_bwd_ws(const WsBwdArgs a)
{
    constexpr int KH = H / 32, NWG = H / 16, H3 = 3 * H;
    const int rep = blockIdx.x / NWG, j = blockIdx.x - rep * NWG;
    const Rep rp{rep, a.R};
    const int Bl = rows_of(a.B, rp);                    // rows of this replica
    const float* xpart = a.xpart + (long long)rep * NWG * WROWS * KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    if (tid == 0) {
        int s = 0;
        for (int t = 0; t < a.T; ++t) { w_off[t] = s; s += a.bs[t]; }
        w_off[a.T] = s;
    }
    // ---- the workgroup's rows of the transposed weights: resident in registers from here on
    float wiT[L > 1 ? L - 1 : 1][3][KH], whT[L][3][KH], wc[2], wxa[3][4];
    {
        const long long urow = 16 * j + n;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const float* p = a.w_hh_t[l] + urow * H3 + q * H + wave * 4 * KH;
#pragma unroll
                for (int m = 0; m < KH; ++m) whT[l][q][m] = p[wk<KH>(g, m)];
            }
#pragma unroll
            for (int l = 1; l < L; ++l) {
                const float* p = a.w_ih_t[l] + urow * H3 + q * H + wave * 4 * KH;
#pragma unroll
                for (int m = 0; m < KH; ++m) wiT[l - 1][q][m] = p[wk<KH>(g, m)];
            }
            // dXS partial: A[w][k], k = this workgroup's unit 4 g + mm of gate q; wave = (16-column block of w, row tile)
            const int wcol = 16 * (wave & 3) + n;
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) wxa[q][mm] = wcol < a.W ? a.w_ih_t[0][(long long)wcol * H3 + q * H + 16 * j + 4 * g + mm] : 0.f;
        }
        // TOP: K = ldcs split over the eight waves: ldcs / 32 values per lane
        const int kc = a.ldcs >> 5;
#pragma unroll
        for (int m = 0; m < 2; ++m) wc[m] = m < kc ? a.wcs_t[urow * a.ldcs + wave * 4 * kc + g * kc + m] : 0.f;
    }
    for (int e = tid; e < WS_MAXL * WROWS * 16; e += NTH) w_carry[e] = 0.f;
    for (int e = tid; e < BSTACK_FLOATS; e += NTH) w_bstack[e] = 0.f;
    __syncthreads();
    WSTAMP_INIT();
    int* ctr = a.ctr + (long long)rep * (L + 1) * CTR_STRIDE;

    // the gate threads' operands (saved gates, previous hidden, incoming top gradient): by LDS-DMA, ahead of the waits
    auto prefetch = [&](int l, int t, int bt) {
        if (tid < 4 * WROWS && (tid >> 2) < bt) {
            float* dst = w_pf + 256 * wave;
            const long long row = (long long)l * a.NT + prow(w_off[t], tid >> 2, rp);
            const int u = 16 * j + 4 * (tid & 3);
            const float* gt = a.gates + row * 4 * H + u;
#pragma unroll
            for (int q = 0; q < 4; ++q) __builtin_amdgcn_global_load_lds(gt + q * H, PPK_LDS_PTR(dst + q * 512), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(a.HP + row * H + u, PPK_LDS_PTR(dst + 4 * 512), 16, 0, 0);
            if (l == L - 1) __builtin_amdgcn_global_load_lds(a.dHtop + prow(w_off[t], tid >> 2, rp) * H + u, PPK_LDS_PTR(dst + 5 * 512), 16, 0, 0);
        }
    };
    // sums of the K-chunk partials (slot 0: accD, slot 1: accC) into w_red
    auto reduce = [&](f32x4 (&accD)[NR], f32x4 (&accC)[NR], int nslot, int nrt, int bt) {
#pragma unroll
        for (int rt = 0; rt < NR; ++rt)
            if (rt < nrt) {
                st4(w_bpart + ((wave * 2 + 0) * WROWS + 16 * rt + n) * 16 + 4 * g, accD[rt]);
                if (nslot > 1) st4(w_bpart + ((wave * 2 + 1) * WROWS + 16 * rt + n) * 16 + 4 * g, accC[rt]);
            }
        __syncthreads();
        {
            const int q = tid >> 7, row = (tid >> 2) & 31, u4 = tid & 3;
            if (q < nslot && row < bt) {
                const float* p = w_bpart + (q * WROWS + row) * 16 + 4 * u4;
                f32x4 v = ld4(p);
#pragma unroll
                for (int w = 1; w < 8; ++w) v += ld4(p + w * 2 * WROWS * 16);
                st4(w_red + (q * WROWS + row) * 16 + 4 * u4, v);
            }
        }
        wait_vmcnt<0>();
        __syncthreads();
    };

    // carry_0 += dgh_0 . W_hh[0] is needed one step later only, and its 96 KB of operand rows are more than one hand-off
    // window hides: with two or more layers the r and z thirds are contracted behind the DCS signal of the step (accK
    // stays in registers), the n third behind the first signal of the NEXT step, where the partial tiles are summed
    f32x4 accK[NR];
#pragma unroll
    for (int rt = 0; rt < NR; ++rt) accK[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    bool have_k = false;
    auto carry0_part = [&](int q0, int q1, int rows, long long rowbase, int ntile) {
#pragma unroll
        for (int rt = 0; rt < NR; ++rt)
            if (rt < ntile) {
                const int r = 16 * rt + n < rows ? 16 * rt + n : rows - 1;
                const long long bo = (prow(rowbase, r, rp) * H3 + wave * 4 * KH) * 4;
                float b0[KH], b1[KH], b2[KH];
                if (q0 <= 0 && 0 < q1) load_b<KH>(b0, a.DGH, bo, g);
                if (q0 <= 1 && 1 < q1) load_b<KH>(b1, a.DGH, bo + 4LL * H, g);
                if (q0 <= 2 && 2 < q1) load_b<KH>(b2, a.DGH, bo + 8LL * H, g);
                if (q0 <= 0 && 0 < q1) mma_acc<KH>(accK[rt], whT[0][0], b0);
                if (q0 <= 1 && 1 < q1) mma_acc<KH>(accK[rt], whT[0][1], b1);
                if (q0 <= 2 && 2 < q1) mma_acc<KH>(accK[rt], whT[0][2], b2);
            }
    };
    auto carry0_finish = [&](int rows, int ntile) {
        reduce(accK, accK, 1, ntile, rows);
        if (tid < 4 * WROWS && (tid >> 2) < rows) {
            float* cp = w_carry + (tid >> 2) * 16 + 4 * (tid & 3);
            st4(cp, ld4(cp) + ld4(w_red + (tid >> 2) * 16 + 4 * (tid & 3)));
        }
        __syncthreads();
#pragma unroll
        for (int rt = 0; rt < NR; ++rt) accK[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    int kstep = 0;                                     // steps this replica has run (its rows start at its own longest sequence)
    for (int t = a.T - 1; t >= 0; --t) {
        const int bt = rows_of(a.bs[t], rp), nn = t + 1 < a.T ? rows_of(a.bs[t + 1], rp) : 0, nrt = (bt + 15) >> 4, nrtn = (nn + 15) >> 4;
        if (bt == 0) continue;
        const long long base = w_off[t], nbase = w_off[t + 1];
        f32x4 accD[NR], accC[NR];
        auto zero = [&](f32x4 (&x)[NR]) {
#pragma unroll
            for (int rt = 0; rt < NR; ++rt) x[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        };
        zero(accD);
        zero(accC);
        // ---- TOP: DCS_{t+1} . [W_c ; W_s] for the rows that were still active at t + 1
        prefetch(L - 1, t, bt);
        if (nn > 0) {
            if (!ws_wait(ctr + L * CTR_STRIDE, NWG * kstep, a.status, &w_flag)) return;
            WSTAMP(0);
            const int kc = a.ldcs >> 5;
#pragma unroll
            for (int rt = 0; rt < NR; ++rt)
                if (rt < nrtn) {
                    const int r = 16 * rt + n < nn ? 16 * rt + n : nn - 1;
                    const long long bo = (prow(nbase, r, rp) * a.ldcs + wave * 4 * kc + g * kc) * 4;
                    if (kc == 2) {
                        const f32x2 v = ld2_sc1(a.DCS, bo);
                        accD[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[0], v[0], accD[rt], 0, 0, 0);
                        accD[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[1], v[1], accD[rt], 0, 0, 0);
                    } else {
                        accD[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[0], ld1_sc1(a.DCS, bo), accD[rt], 0, 0, 0);
                    }
                }
        }
#pragma unroll
        for (int l = L - 1; l >= 0; --l) {
            // ---- gate backward of layer l: dh = accD (+ carry, + dHtop for the top layer)
            WSTAMP(1);
            reduce(accD, accC, l < L - 1 ? 2 : 1, nrt, bt);
            WSTAMP(2);
            if (tid < 4 * WROWS) {
                const int row = tid >> 2, u4 = tid & 3, u = 16 * j + 4 * u4;
                if (row < bt) {
                    const f32x4 rr = ld4(w_pf + 4 * tid), zz = ld4(w_pf + 512 + 4 * tid), nv = ld4(w_pf + 2 * 512 + 4 * tid),
                                hn = ld4(w_pf + 3 * 512 + 4 * tid), hp = ld4(w_pf + 4 * 512 + 4 * tid);
                    f32x4 dsum = ld4(w_red + row * 16 + 4 * u4);
                    if (l == L - 1) {
                        if (row >= nn) dsum = f32x4{0.f, 0.f, 0.f, 0.f};         // no stack read this row's hidden at t + 1
                        dsum += ld4(w_pf + 5 * 512 + 4 * tid);
                    } else {
                        // carry_{l+1} += dgh_{l+1} . W_hh[l+1]
                        float* cp = w_carry + ((l + 1) * WROWS + row) * 16 + 4 * u4;
                        st4(cp, ld4(cp) + ld4(w_red + (WROWS + row) * 16 + 4 * u4));
                        // layer l + 1 read h_l * drop_l: its input gradient comes back through the same factors
                        if (a.drop) dsum *= ld4(a.drop + ((long long)l * a.NT + prow(base, row, rp)) * H + u);
                    }
                    float* cl = w_carry + (l * WROWS + row) * 16 + 4 * u4;
                    const f32x4 dh = dsum + ld4(cl);
                    f32x4 dn, dz, dr, dnr, c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        dn[k] = dh[k] * (1.0f - zz[k]) * (1.0f - nv[k] * nv[k]);
                        dz[k] = dh[k] * (hp[k] - nv[k]) * zz[k] * (1.0f - zz[k]);
                        dr[k] = dn[k] * hn[k] * rr[k] * (1.0f - rr[k]);
                        dnr[k] = dn[k] * rr[k];
                        c[k] = dh[k] * zz[k];
                    }
                    const long long o = ((((long long)l * a.NT + prow(base, row, rp)) * 3) * H + u) * 4;
                    st4_sc1(a.DGI, o, dr); st4_sc1(a.DGI, o + 4LL * H, dz); st4_sc1(a.DGI, o + 8LL * H, dn);
                    st4_sc1(a.DGH, o, dr); st4_sc1(a.DGH, o + 4LL * H, dz); st4_sc1(a.DGH, o + 8LL * H, dnr);
                    st4(cl, c);
                    if (l == 0) { st4(w_g0 + row * 48 + 4 * u4, dr); st4(w_g0 + row * 48 + 16 + 4 * u4, dz); st4(w_g0 + row * 48 + 32 + 4 * u4, dn); }
                }
            }
            __syncthreads();
            WSTAMP(3);
            if (l == 0) break;
            group_signal(ctr + l * CTR_STRIDE);
            WSTAMP(4);
            // ---- P_l: accD = dgi_l . W_ih[l] (-> dh_{l-1}), accC = dgh_l . W_hh[l] (-> carry_l)
            prefetch(l - 1, t, bt);
            if (l == L - 1 && have_k) {                    // inside the hand-off window: the rest of the previous step's carry_0
                carry0_part(2, 3, nn, nbase, nrtn);
                carry0_finish(nn, nrtn);
                have_k = false;
            }
            zero(accD);
            zero(accC);
            if (!ws_wait(ctr + l * CTR_STRIDE, NWG * (kstep + 1), a.status, &w_flag)) return;
            WSTAMP(5);
            {
                const float* gi = a.DGI + (long long)l * a.NT * H3;
                const float* gh = a.DGH + (long long)l * a.NT * H3;
#pragma unroll
                for (int rt = 0; rt < NR; ++rt)
                    if (rt < nrt) {
                        const int r = 16 * rt + n < bt ? 16 * rt + n : bt - 1;
                        const long long bo = (prow(base, r, rp) * H3 + wave * 4 * KH) * 4;
                        // every load of the row tile goes out before its first MFMA: one memory round trip, not four
                        float b0[KH], b1[KH], b2[KH], b3[KH];
                        load_b<KH>(b0, gi, bo, g);                          // dr
                        load_b<KH>(b1, gi, bo + 4LL * H, g);                // dz          (dgi == dgh for both)
                        load_b<KH>(b2, gi, bo + 8LL * H, g);                // dn with W_ih
                        load_b<KH>(b3, gh, bo + 8LL * H, g);                // dn r with W_hh
                        mma_acc<KH>(accD[rt], wiT[l - 1][0], b0);
                        mma_acc<KH>(accC[rt], whT[l][0], b0);
                        mma_acc<KH>(accD[rt], wiT[l - 1][1], b1);
                        mma_acc<KH>(accC[rt], whT[l][1], b1);
                        mma_acc<KH>(accD[rt], wiT[l - 1][2], b2);
                        mma_acc<KH>(accC[rt], whT[l][2], b3);
                    }
            }
        }
        // ---- layer 0 is done: dXS partial over this workgroup's 48 gate columns, hand-off, stack backward, hand-off
        {
            const int ot = wave & 3, rt = wave >> 2;
            if (16 * ot < KS && 16 * rt < bt) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const f32x4 gv = ld4(w_g0 + (16 * rt + n) * 48 + 16 * q + 4 * g);
#pragma unroll
                    for (int mm = 0; mm < 4; ++mm) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wxa[q][mm], gv[mm], acc, 0, 0, 0);
                }
                if (16 * rt + n < bt) st4_sc1(xpart, (((long long)j * WROWS + 16 * rt + n) * KS + 16 * ot + 4 * g) * 4, acc);
            }
        }
        WSTAMP(6);
        group_signal(ctr);
        WSTAMP(7);
        ws_stack_bwd_pre(a, t, kstep, j, NWG, rp);
        if (!ws_wait(ctr, NWG * (kstep + 1), a.status, &w_flag)) return;
        WSTAMP(11);
        ws_stack_bwd_post(a, t, kstep, j, NWG, KS, rp);
        WSTAMP(8);
        group_signal(ctr + L * CTR_STRIDE);
        WSTAMP(9);
        // ---- carry_0 += dgh_0 . W_hh[0]: only this workgroup's next step needs it — it fills the DCS hand-off.  (Requesting
        // its 96 KB of rows before the owners finish their stack rows put those loads in front of the owners' dXS partials:
        // 44.4 -> 48.6 k cycles per timestep, not kept.)
        if constexpr (L > 1) {
            carry0_part(0, 2, bt, base, nrt);
            have_k = true;
        } else {
            carry0_part(0, 3, bt, base, nrt);
            carry0_finish(bt, nrt);
        }
        WSTAMP(10);
        ++kstep;
    }
    // the stack of step 0 read the INITIAL hidden state of the top layer: dh0[L-1] += DCS_0 . [W_c ; W_s]
    {
        if (have_k) {
            carry0_part(2, 3, Bl, 0, (Bl + 15) >> 4);
            carry0_finish(Bl, (Bl + 15) >> 4);
        }
        if (!ws_wait(ctr + L * CTR_STRIDE, NWG * kstep, a.status, &w_flag)) return;
        f32x4 accD[NR];
        const int kc = a.ldcs >> 5, nrt = (Bl + 15) >> 4;
#pragma unroll
        for (int rt = 0; rt < NR; ++rt) {
            accD[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (rt < nrt) {
                const int r = 16 * rt + n < Bl ? 16 * rt + n : Bl - 1;
                const long long bo = (prow(0, r, rp) * a.ldcs + wave * 4 * kc + g * kc) * 4;
                if (kc == 2) {
                    const f32x2 v = ld2_sc1(a.DCS, bo);
                    accD[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[0], v[0], accD[rt], 0, 0, 0);
                    accD[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[1], v[1], accD[rt], 0, 0, 0);
                } else {
                    accD[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[0], ld1_sc1(a.DCS, bo), accD[rt], 0, 0, 0);
                }
            }
        }
        reduce(accD, accD, 1, nrt, Bl);
        for (int e = tid; e < L * WROWS * 4; e += NTH) {
            const int l = e / (WROWS * 4), row = (e >> 2) & (WROWS - 1), u4 = e & 3;
            if (row < Bl) {
                f32x4 v = ld4(w_carry + (l * WROWS + row) * 16 + 4 * u4);
                if (l == L - 1) v += ld4(w_red + row * 16 + 4 * u4);
                st4(a.dh0 + ((long long)l * a.B + prow(0, row, rp)) * H + 16 * j + 4 * u4, v);
            }
        }
    }
    WSTAMP_DUMP(a.dbg, j);
}

typedef void (*WsBwdKernel)(const WsBwdArgs);
template <int H, int KS> WsBwdKernel pick_bl(int L, int nr)
{
    if (L == 1) return nr == 1 ? k_seq_bwd_ws<H, KS, 1, 1> : k_seq_bwd_ws<H, KS, 1, 2>;
    if (L == 2) return nr == 1 ? k_seq_bwd_ws<H, KS, 2, 1> : k_seq_bwd_ws<H, KS, 2, 2>;
    if constexpr (H <= 256) { if (L == 3) return nr == 1 ? k_seq_bwd_ws<H, KS, 3, 1> : k_seq_bwd_ws<H, KS, 3, 2>; }
    return nullptr;
}
WsBwdKernel pick_bwd(int H, int ks, int L, int nr)
{
    if (ks == 64) {
        if (H == 512) return pick_bl<512, 64>(L, nr);
        if (H == 256) return pick_bl<256, 64>(L, nr);
        if (H == 128) return pick_bl<128, 64>(L, nr);
        if (H == 64) return pick_bl<64, 64>(L, nr);
        if (H == 32) return pick_bl<32, 64>(L, nr);
    } else if (ks == 32) {
        if (H == 128) return pick_bl<128, 32>(L, nr);
        if (H == 64) return pick_bl<64, 32>(L, nr);
        if (H == 32) return pick_bl<32, 32>(L, nr);
    }
    return nullptr;
}

typedef void (*WsKernel)(const WsArgs);
template <int H, int KS> WsKernel pick_l(int L, int nr)
{
    if (L == 1) return nr == 1 ? k_seq_fwd_ws<H, KS, 1, 1> : k_seq_fwd_ws<H, KS, 1, 2>;
    if (L == 2) return nr == 1 ? k_seq_fwd_ws<H, KS, 2, 1> : k_seq_fwd_ws<H, KS, 2, 2>;
    if constexpr (H <= 256) { if (L == 3) return nr == 1 ? k_seq_fwd_ws<H, KS, 3, 1> : k_seq_fwd_ws<H, KS, 3, 2>; }
    return nullptr;
}
template <int H, int KS> WsKernel pick_ll(int L)
{
    if (L == 1) return k_seq_fwd_wsl<H, KS, 1>;
    if (L == 2) return k_seq_fwd_wsl<H, KS, 2>;
    if constexpr (H <= 256) { if (L == 3) return k_seq_fwd_wsl<H, KS, 3>; }
    return nullptr;
}
WsKernel pick_fwdl(int H, int ks, int L)
{
    if (ks == 64) {
        if (H == 512) return pick_ll<512, 64>(L);
        if (H == 256) return pick_ll<256, 64>(L);
        if (H == 128) return pick_ll<128, 64>(L);
        if (H == 64) return pick_ll<64, 64>(L);
        if (H == 32) return pick_ll<32, 64>(L);
    } else if (ks == 32) {
        if (H == 128) return pick_ll<128, 32>(L);
        if (H == 64) return pick_ll<64, 32>(L);
        if (H == 32) return pick_ll<32, 32>(L);
    }
    return nullptr;
}
// nr: 16-row tiles of the batch (1: at most 16 sequences, 2: at most 32)
WsKernel pick_fwd(int H, int ks, int L, int nr)
{
    if (ks == 64) {
        if (H == 512) return pick_l<512, 64>(L, nr);
        if (H == 256) return pick_l<256, 64>(L, nr);
        if (H == 128) return pick_l<128, 64>(L, nr);
        if (H == 64) return pick_l<64, 64>(L, nr);
        if (H == 32) return pick_l<32, 64>(L, nr);
    } else if (ks == 32) {
        if (H == 128) return pick_l<128, 32>(L, nr);
        if (H == 64) return pick_l<64, 32>(L, nr);
        if (H == 32) return pick_l<32, 32>(L, nr);
    }
    return nullptr;
}

// auto mode takes these kernels up to this many sequences per call (PPK_WS_MAX_ROWS in the environment; default: whenever one launch fits)
int g_ws_max_rows = -1;
int ws_auto_max_rows()
{
    if (g_ws_max_rows < 0) { const char* e = getenv("PPK_WS_MAX_ROWS"); g_ws_max_rows = e ? atoi(e) : (1 << 30); }
    return g_ws_max_rows;
}

// How a batch is dealt to replicas on the current device: as many replicas of at most 16 rows (one row tile: the faster
// kernels) as the CUs allow — a replica is H / 16 workgroups, one per CU, all resident at once —, else replicas of at most
// 32 rows.  false: the batch does not fit one launch.
bool ws_plan(int64_t B, int64_t H, int& R, int& nr)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    const int64_t nwg = H / 16, rmax = cus / nwg;
    if (B < 1 || rmax < 1) return false;
    int64_t r = std::min<int64_t>((B + 15) / 16, rmax);
    const int64_t rows = (B + r - 1) / r;
    if (rows > WROWS) return false;
    R = (int)r;
    nr = rows > 16 ? 2 : 1;
    return true;
}

// replicas of the looped forward: every CU gets a workgroup; rows of a replica <= 256 and <= 8 x H / 16 (stack ownership)
bool wsl_plan(int64_t B, int64_t H, int& R);
bool wsl_plan_ok(int64_t B, int64_t H) { int R; return wsl_plan(B, H, R); }
bool wsl_plan(int64_t B, int64_t H, int& R)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    const int64_t nwg = H / 16, rmax = cus / nwg;
    if (B < 1 || rmax < 1) return false;
    const int64_t r = std::min<int64_t>((B + 31) / 32, rmax), rows = (B + r - 1) / r;
    if (rows > LROWS || rows > LOWN * nwg) return false;
    R = (int)r;
    return true;
}

}  // namespace

namespace ppk {

// auto mode: wherever one launch of replicas fits the device these kernels are ~2 x faster than either other schedule
// (H 512: up to 256 sequences, H 256: up to 512; tools/ws_replica_shapes.py, profiles/r03_ws_replicas.txt)
bool seq_auto_ws(int64_t B, int64_t H, int R)
{
    (void)H;
    return R >= 1 && B <= ws_auto_max_rows();
}

size_t seq_fwd_ws_ws_bytes(int64_t B, int64_t H, int64_t L, int64_t W)
{
    int R = 1, nr = 1;
    if (H % 16 || L > WS_MAXL || !ws_plan(B, H, R, nr)) return 0;
    const int64_t ldcs = (3 + W + 31) / 32 * 32, nwg = H / 16;
    return ppk_align_up(sizeof(float) * (size_t)H * (size_t)ldcs, 256) + ppk_align_up(sizeof(int) * ((size_t)R * (size_t)(L + 1) * CTR_STRIDE + 64), 256)
           + ppk_align_up(sizeof(float) * (size_t)R * (size_t)nwg * WROWS * (size_t)ldcs, 256);
}

bool seq_fwd_ws_ok(int64_t T, int64_t B, int64_t NT, int64_t H, int64_t L, int64_t W, int64_t D, int64_t ks, int64_t ldcs)
{
    const int mode = seq_persist_mode(-2);
    if (mode != 2 && mode != 3 && mode != -1) return false;
    if (T < 1 || T > MAXT || B < 1 || L < 1 || L > WS_MAXL || H % 16 || L * NT * 4 * H * 4 >= (1LL << 31)) return false;      // byte offsets of the buffer accesses
    if (ldcs != (3 + W + 31) / 32 * 32 || ks != (W + 31) / 32 * 32 || ldcs > 64) return false;
    if (mode == 3 && B > WROWS && wsl_plan_ok(B, H)) return false;      // mode 3: the looped forward takes the batches beyond one 32-row replica
    int R, nr;
    if (!ws_plan(B, H, R, nr)) return false;            // every workgroup resident at once, one per CU
    if (mode == -1 && !seq_auto_ws(B, H, R)) return false;
    if (!pick_fwd((int)H, (int)ks, (int)L, nr)) return false;
    const int64_t nwg = H / 16, nown = (WROWS + nwg - 1) / nwg, DW4 = (D * W + 3) / 4 * 4;
    return nown * 2 * DW4 <= STACK_FLOATS;
}

int seq_fwd_ws(int64_t T, const int64_t* bs, int64_t B, int64_t NT, int64_t V, int64_t H, int64_t L, int64_t W, int64_t D,
               const int64_t* tokens, const float* table, const float* wx0s, int64_t ks,
               const float* const* w_ih, const float* const* w_hh, const float* const* b_ih, const float* const* b_hh,
               const float* wcs, const float* bcs, const float* h0, const float* S0,
               float* XS, float* Hs, float* HP, float* gates, float* S, float* CS, int64_t ldcs,
               const float* drop, float* Hdrop, void* ws, size_t ws_bytes, hipStream_t stream)
{
    if (ws_bytes < seq_fwd_ws_ws_bytes(B, H, L, W)) PPK_FAIL(PPK_ERR_WS, "seq_fwd_ws: workspace too small");
    int R = 1, nr = 1;
    if (!ws_plan(B, H, R, nr)) PPK_FAIL(PPK_ERR_UNSUP, "seq_fwd_ws: the batch does not fit one launch");
    WsKernel kern = pick_fwd((int)H, (int)ks, (int)L, nr);
    if (!kern) PPK_FAIL(PPK_ERR_UNSUP, "seq_fwd_ws: unsupported shape");
    WsArgs a{};
    a.T = (int)T; a.B = (int)B; a.W = (int)W; a.D = (int)D; a.V = (int)V; a.ldcs = (int)ldcs; a.DW = (int)(D * W); a.R = R;
    a.NT = NT;
    a.tokens = tokens; a.table = table; a.wx0s = wx0s;
    for (int l = 0; l < (int)L; ++l) { a.w_ih[l] = w_ih[l]; a.w_hh[l] = w_hh[l]; a.b_ih[l] = b_ih[l]; a.b_hh[l] = b_hh[l]; }
    a.bcs = bcs; a.h0 = h0; a.S0 = S0;
    a.XS = XS; a.Hs = Hs; a.HP = HP; a.gates = gates; a.S = S; a.CS = CS;
    a.drop = L > 1 ? drop : nullptr; a.HD = Hdrop;
#ifdef PPK_WS_TIMING
    a.dbg = g_ws_dbg;
#endif
    for (int t = 0; t < (int)T; ++t) a.bs[t] = (int)bs[t];
    const int nwg = (int)(H / 16);
    char* cur = (char*)ws;
    float* wcs_t = (float*)ppk_carve(cur, sizeof(float) * (size_t)H * (size_t)ldcs);
    const size_t nctr = (size_t)R * (size_t)(L + 1) * CTR_STRIDE + 64;
    int* ctr = (int*)ppk_carve(cur, sizeof(int) * nctr);
    a.wcs_t = wcs_t; a.ctr = ctr; a.status = ctr + (size_t)R * (size_t)(L + 1) * CTR_STRIDE;
    a.hpart = (float*)ppk_carve(cur, sizeof(float) * (size_t)R * (size_t)nwg * WROWS * (size_t)ldcs);
    PPK_HIP(hipMemsetAsync(ctr, 0, sizeof(int) * nctr, stream));
    k_ws_wcs_transpose<<<dim3((unsigned)std::min<int64_t>((H * ldcs + 255) / 256, 1024)), dim3(256), 0, stream>>>(wcs, (int)(3 + W), (int)H, (int)ldcs, wcs_t);
    PPK_CHECK_LAUNCH("k_ws_wcs_transpose");
    {
        const double flops = 2.0 * (double)NT * 3.0 * (double)H * ((double)ks + (2.0 * (double)L - 1.0) * (double)H);
        ProfScope prof(PK_SEQ_FWD, flops, stream);
        kern<<<dim3((unsigned)(R * nwg)), dim3(NTH), 0, stream>>>(a);
        PPK_CHECK_LAUNCH("k_seq_fwd_ws");
    }
    float* ptrs[5] = {Hs, CS, XS, S, a.drop ? Hdrop : nullptr};
    const long long counts[5] = {(long long)L * NT * H, NT * ldcs, NT * ks, NT * D * W, (long long)(L - 1) * NT * H};
    return seq_poison(a.status, ptrs, counts, 5, stream);
}

size_t seq_bwd_ws_ws_bytes(int64_t B, int64_t H, int64_t L, int64_t W)
{
    int R = 1, nr = 1;
    if (H % 16 || L > WS_MAXL || !ws_plan(B, H, R, nr)) return 0;
    const int64_t ks = (W + 31) / 32 * 32, nwg = H / 16;
    return ppk_align_up(sizeof(int) * ((size_t)R * (size_t)(L + 1) * CTR_STRIDE + 64), 256) + ppk_align_up(sizeof(float) * (size_t)R * (size_t)nwg * WROWS * (size_t)ks, 256);
}

bool seq_bwd_ws_ok(int64_t T, int64_t B, int64_t NT, int64_t H, int64_t L, int64_t W, int64_t D, int64_t ks, int64_t ldcs)
{
    const int mode = seq_persist_mode(-2);
    if (mode != 2 && mode != 3 && mode != -1) return false;
    if (T < 1 || T > MAXT || B < 1 || L < 1 || L > WS_MAXL || H % 16) return false;
    if (ldcs != (3 + W + 31) / 32 * 32 || ks != (W + 31) / 32 * 32 || ldcs > 64) return false;
    if (NT * 3 * H * 4 * L >= (1LL << 31)) return false;              // byte offsets of the buffer accesses into DGI / DGH
    int R, nr;
    if (!ws_plan(B, H, R, nr)) return false;
    if (mode == -1 && !seq_auto_ws(B, H, R)) return false;
    if (!pick_bwd((int)H, (int)ks, (int)L, nr)) return false;
    const int64_t nwg = H / 16, nown = (WROWS + nwg - 1) / nwg, DW4 = (D * W + 3) / 4 * 4;
    return nown * 3 * DW4 <= BSTACK_FLOATS;
}

int seq_bwd_ws(int64_t T, const int64_t* bs, int64_t B, int64_t NT, int64_t H, int64_t L, int64_t W, int64_t D,
               const float* const* w_ih_t, const float* const* w_hh_t, const float* wcs_t,
               const float* HP, const float* gates, const float* S, const float* CS, int64_t ldcs,
               const float* dHtop, float* dXS, int64_t ks, float* DGI, float* DGH, float* DCS, float* dh0,
               const float* drop, void* ws, size_t ws_bytes, hipStream_t stream)
{
    if (ws_bytes < seq_bwd_ws_ws_bytes(B, H, L, W)) PPK_FAIL(PPK_ERR_WS, "seq_bwd_ws: workspace too small");
    int R = 1, nr = 1;
    if (!ws_plan(B, H, R, nr)) PPK_FAIL(PPK_ERR_UNSUP, "seq_bwd_ws: the batch does not fit one launch");
    WsBwdKernel kern = pick_bwd((int)H, (int)ks, (int)L, nr);
    if (!kern) PPK_FAIL(PPK_ERR_UNSUP, "seq_bwd_ws: unsupported shape");
    WsBwdArgs a{};
    a.T = (int)T; a.B = (int)B; a.W = (int)W; a.D = (int)D; a.ldcs = (int)ldcs; a.DW = (int)(D * W); a.R = R;
    a.NT = NT;
    for (int l = 0; l < (int)L; ++l) { a.w_ih_t[l] = w_ih_t[l]; a.w_hh_t[l] = w_hh_t[l]; }
    a.wcs_t = wcs_t;
    a.HP = HP; a.gates = gates; a.S = S; a.CS = CS; a.dHtop = dHtop;
    a.drop = L > 1 ? drop : nullptr;
    a.dXS = dXS; a.DGI = DGI; a.DGH = DGH; a.DCS = DCS; a.dh0 = dh0;
#ifdef PPK_WS_TIMING
    a.dbg = g_ws_dbg ? g_ws_dbg + 16 * 64 : nullptr;
#endif
    for (int t = 0; t < (int)T; ++t) a.bs[t] = (int)bs[t];
    const int nwg = (int)(H / 16);
    char* cur = (char*)ws;
    const size_t nctr = (size_t)R * (size_t)(L + 1) * CTR_STRIDE + 64;
    int* ctr = (int*)ppk_carve(cur, sizeof(int) * nctr);
    a.ctr = ctr; a.status = ctr + (size_t)R * (size_t)(L + 1) * CTR_STRIDE;
    a.xpart = (float*)ppk_carve(cur, sizeof(float) * (size_t)R * (size_t)nwg * WROWS * (size_t)ks);
    PPK_HIP(hipMemsetAsync(ctr, 0, sizeof(int) * nctr, stream));
    {
        const double flops = 2.0 * (double)NT * (double)H * (3.0 * (double)H * (2.0 * (double)L - 1.0) + (double)ldcs) + 2.0 * (double)NT * 3.0 * (double)H * (double)ks;
        ProfScope prof(PK_SEQ_BWD, flops, stream);
        kern<<<dim3((unsigned)(R * nwg)), dim3(NTH), 0, stream>>>(a);
        PPK_CHECK_LAUNCH("k_seq_bwd_ws");
    }
    float* ptrs[5] = {dXS, DGI, DGH, DCS, dh0};
    const long long counts[5] = {NT * ks, (long long)L * NT * 3 * H, (long long)L * NT * 3 * H, NT * ldcs, (long long)L * B * H};
    return seq_poison(a.status, ptrs, counts, 5, stream);
}

// ---- looped forward (mode 3: beyond the batches one launch of <= 32-row replicas holds) ----------------------------------
size_t seq_fwd_wsl_ws_bytes(int64_t B, int64_t H, int64_t L, int64_t W)
{
    int R = 1;
    if (H % 16 || L > WS_MAXL || !wsl_plan(B, H, R)) return 0;
    const int64_t ldcs = (3 + W + 31) / 32 * 32, nwg = H / 16;
    return ppk_align_up(sizeof(float) * (size_t)H * (size_t)ldcs, 256) + ppk_align_up(sizeof(int) * ((size_t)R * (size_t)(L + 1) * CTR_STRIDE + 64), 256)
           + ppk_align_up(sizeof(float) * (size_t)R * (size_t)nwg * LROWS * (size_t)ldcs, 256);
}

bool seq_fwd_wsl_ok(int64_t T, int64_t B, int64_t NT, int64_t H, int64_t L, int64_t W, int64_t D, int64_t ks, int64_t ldcs)
{
    const int mode = seq_persist_mode(-2);
    if ((mode != 3 && mode != -1) || B <= WROWS) return false;
    if (T < 1 || T > MAXT || L < 1 || L > WS_MAXL || H % 16 || L * NT * 4 * H * 4 >= (1LL << 31)) return false;
    if (ldcs != (3 + W + 31) / 32 * 32 || ks != (W + 31) / 32 * 32 || ldcs > 64 || 2 * W > 128) return false;
    int R;
    if (!wsl_plan(B, H, R)) return false;
    // auto: where it was measured faster than both other schedules — rows per replica x H <= 64 x 512 (H 512: batch 257 .. 512,
    // training step 0.91 - 0.92 of the faster other one; 0.98 at 768, 1.03 at 1024, 1.12 at 2048.  H 256: batch 513 .. 1536,
    // 0.90 - 0.92.  H 128: up to 2048, 0.69 - 1.0: profiles/r03_ws_replicas.txt)
    if (mode == -1 && (B + R - 1) / R * H > 64 * 512) return false;
    (void)D;
    return pick_fwdl((int)H, (int)ks, (int)L) != nullptr;
}

int seq_fwd_wsl(int64_t T, const int64_t* bs, int64_t B, int64_t NT, int64_t V, int64_t H, int64_t L, int64_t W, int64_t D,
                const int64_t* tokens, const float* table, const float* wx0s, int64_t ks,
                const float* const* w_ih, const float* const* w_hh, const float* const* b_ih, const float* const* b_hh,
                const float* wcs, const float* bcs, const float* h0, const float* S0,
                float* XS, float* Hs, float* HP, float* gates, float* S, float* CS, int64_t ldcs,
                const float* drop, float* Hdrop, void* ws, size_t ws_bytes, hipStream_t stream)
{
    if (ws_bytes < seq_fwd_wsl_ws_bytes(B, H, L, W)) PPK_FAIL(PPK_ERR_WS, "seq_fwd_wsl: workspace too small");
    int R = 1;
    if (!wsl_plan(B, H, R)) PPK_FAIL(PPK_ERR_UNSUP, "seq_fwd_wsl: the batch does not fit one launch");
    WsKernel kern = pick_fwdl((int)H, (int)ks, (int)L);
    if (!kern) PPK_FAIL(PPK_ERR_UNSUP, "seq_fwd_wsl: unsupported shape");
    WsArgs a{};
    a.T = (int)T; a.B = (int)B; a.W = (int)W; a.D = (int)D; a.V = (int)V; a.ldcs = (int)ldcs; a.DW = (int)(D * W); a.R = R;
    a.NT = NT;
    a.tokens = tokens; a.table = table; a.wx0s = wx0s;
    for (int l = 0; l < (int)L; ++l) { a.w_ih[l] = w_ih[l]; a.w_hh[l] = w_hh[l]; a.b_ih[l] = b_ih[l]; a.b_hh[l] = b_hh[l]; }
    a.bcs = bcs; a.h0 = h0; a.S0 = S0;
    a.XS = XS; a.Hs = Hs; a.HP = HP; a.gates = gates; a.S = S; a.CS = CS;
    a.drop = L > 1 ? drop : nullptr; a.HD = Hdrop;
    for (int t = 0; t < (int)T; ++t) a.bs[t] = (int)bs[t];
    const int nwg = (int)(H / 16);
    char* cur = (char*)ws;
    float* wcs_t = (float*)ppk_carve(cur, sizeof(float) * (size_t)H * (size_t)ldcs);
    const size_t nctr = (size_t)R * (size_t)(L + 1) * CTR_STRIDE + 64;
    int* ctr = (int*)ppk_carve(cur, sizeof(int) * nctr);
    a.wcs_t = wcs_t; a.ctr = ctr; a.status = ctr + (size_t)R * (size_t)(L + 1) * CTR_STRIDE;
    a.hpart = (float*)ppk_carve(cur, sizeof(float) * (size_t)R * (size_t)nwg * LROWS * (size_t)ldcs);
    PPK_HIP(hipMemsetAsync(ctr, 0, sizeof(int) * nctr, stream));
    k_ws_wcs_transpose<<<dim3((unsigned)std::min<int64_t>((H * ldcs + 255) / 256, 1024)), dim3(256), 0, stream>>>(wcs, (int)(3 + W), (int)H, (int)ldcs, wcs_t);
    PPK_CHECK_LAUNCH("k_ws_wcs_transpose");
    {
        const double flops = 2.0 * (double)NT * 3.0 * (double)H * ((double)ks + (2.0 * (double)L - 1.0) * (double)H);
        ProfScope prof(PK_SEQ_FWD, flops, stream);
        kern<<<dim3((unsigned)(R * nwg)), dim3(NTH), 0, stream>>>(a);
        PPK_CHECK_LAUNCH("k_seq_fwd_wsl");
    }
    float* ptrs[5] = {Hs, CS, XS, S, a.drop ? Hdrop : nullptr};
    const long long counts[5] = {(long long)L * NT * H, NT * ldcs, NT * ks, NT * D * W, (long long)(L - 1) * NT * H};
    return seq_poison(a.status, ptrs, counts, 5, stream);
}

}  // namespace ppk

#ifdef PPK_WS_TIMING
// diagnosis builds only (tools/ws_timing.py): [2 passes][64 workgroups][16] int64 cycle sums per phase (forward, then backward)
extern "C" void ppk_dbg_ws_timing_buffer(long long* p) { g_ws_dbg = p; }
#endif
