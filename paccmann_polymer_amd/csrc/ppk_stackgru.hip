// K1 — stack-augmented GRU sequence (forward + hand-written backward) over a
// packed, length-sorted batch.
//
// Replaces the per-timestep Python loop of paccmann_chemistry's StackGRU
// (embedding -> stack update from hidden[-1] -> nn.GRU with seq_len 1) that the
// reference reaches from
//   paccmann_polymer/models/paccmann_encoder.py:166  (TeacherVAE.encode)
//   paccmann_polymer/models/paccmann_encoder.py:168  (TeacherVAE.decode)
// (semantics: SURVEY.md Appendix B; CPU restatement: oracle/stackgru_ref.py).
//
// Layout (MI355X-first: 288 GB of HBM, so every activation the backward needs is
// kept, nothing is recomputed):  rows of step t live at packed offset
// off[t] .. off[t]+B_t in sorted order, so "hidden of the previous step" for the
// active prefix is simply the row block of step t-1 — no copies, no masking.
//   Hs    [L, NT, H]  hidden after the step          HP [L, NT, H] hidden before it
//   gates [L, NT, 4H] r, z, n, (W_hn h + b_hn)
//   S     [NT, D, W]  stack after the step           CS [NT, ldcs] softmax(3) | tanh(W)
//   XS    [NT, ks]    stack top fed to layer 0 (ks = W rounded up to 32, zero padded)
// Three launch schedules behind the same entry points (ppk_seq_persist_mode):
//  * weights-stationary (ppk_seq_ws.hip): small and medium batches, replicas of <= 32 rows whose workgroups keep their rows of
//    every weight matrix in registers — taken in auto mode wherever one launch fits the device;
//  * persistent (ppk_seq_persist.hip / ppk_seq_persist_bwd.hip): the whole time loop in ONE launch per pass,
//    row-block groups synchronised among their own workgroups only — taken in auto mode where the groups fill the
//    machine (H 512: from batch 1024 up; ppk::seq_auto_persistent);
//  * per-step launches (this file): the gate contractions as fp32-MFMA GEMMs (ppk_gemm_nt.hip) with the GRU gate
//    math fused into their epilogues, 2 + L launches per timestep — faster where the groups' workgroups
//    cannot fill the machine or need three or more machine-fulls ("Small batches").
// Per step (forward):   stack update -> {cell 0 (K = ks, + table[token]) ; W_hh[L-1] h of the previous step}
// -> for l >= 1 {cell l ; W_hh[l-1] h for the NEXT step} -> stack heads for the next step.
// The embedding half of the layer-0 input product is hoisted out of the time loop:
// table[v] = emb[v] . W_ih0[:, :E]^T + b_ih0 is a [V, 3H] lookup.
#include "ppk_common.h"
#include "ppk_kernels.h"
#include <algorithm>
#include <vector>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// row image <-> global: 16-byte accesses when the row (D*W floats) keeps 16-byte alignment, else scalar
__device__ __forceinline__ void row_to_lds(float* dst, const float* src, int n, bool vec)
{
    if (vec) {
        for (int j = threadIdx.x; j < (n >> 2); j += 256) reinterpret_cast<f32x4*>(dst)[j] = reinterpret_cast<const f32x4*>(src)[j];
    } else {
        for (int i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
    }
}

// one block per active row: stack update (Appendix B) + stack top into X.
// The previous stack row is staged in LDS once (the update reads every element three times: as
// itself, as the element above and as the element below) and the new row leaves as 16-byte stores.
__global__ __launch_bounds__(256) void k_stack_fwd(const float* __restrict__ csraw, long long ldcs,
                                                   const float* __restrict__ Sprev,   // null at t == 0
                                                   float* __restrict__ Snew, float* __restrict__ CS,
                                                   float* __restrict__ Xtop, long long ldx, int D, int W)
{
    extern __shared__ __align__(16) float sm[];      // previous row [D*W] (padded to 4), then [0..2] controls, [3..3+W) s_in
    const long long b = blockIdx.x;
    const int DW = D * W, DW4 = (DW + 3) & ~3;
    const bool vec = (DW & 3) == 0;
    float* spl = sm;
    float* act = sm + DW4;
    const float* raw = csraw + b * ldcs;
    if (threadIdx.x == 0) {
        float p0 = raw[0], p1 = raw[1], p2 = raw[2];
        float m = fmaxf(p0, fmaxf(p1, p2));
        float e0 = expf(p0 - m), e1 = expf(p1 - m), e2 = expf(p2 - m);
        float inv = 1.0f / (e0 + e1 + e2);
        act[0] = e0 * inv; act[1] = e1 * inv; act[2] = e2 * inv;
    }
    for (int w = threadIdx.x; w < W; w += 256) act[3 + w] = tanhf(raw[3 + w]);
    const long long base = b * (long long)DW;
    if (Sprev) row_to_lds(spl, Sprev + base, DW, vec);
    else for (int i = threadIdx.x; i < DW; i += 256) spl[i] = 0.f;
    __syncthreads();
    const float a_push = act[0], a_pop = act[1], a_noop = act[2];
    float* cs = CS + b * ldcs;
    for (int i = threadIdx.x; i < 3 + W; i += 256) cs[i] = act[i];
    float* sn = Snew + base;
    auto upd = [&](int i) {
        const int d = i / W, w = i - d * W;
        const float up = d == 0 ? act[3 + w] : spl[i - W];
        const float down = d < D - 1 ? spl[i + W] : 0.f;
        return a_noop * spl[i] + a_push * up + a_pop * down;
    };
    if (vec) {
        for (int j = threadIdx.x; j < (DW >> 2); j += 256) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = upd(4 * j + k);
            reinterpret_cast<f32x4*>(sn)[j] = v;
#pragma unroll
            for (int k = 0; k < 4; ++k) if (4 * j + k < W) Xtop[b * ldx + 4 * j + k] = v[k];
        }
    } else {
        for (int i = threadIdx.x; i < DW; i += 256) {
            const float v = upd(i);
            sn[i] = v;
            if (i < W) Xtop[b * ldx + i] = v;
        }
    }
    for (int w = W + threadIdx.x; w < ldx; w += 256) Xtop[b * ldx + w] = 0.f;      // zero padding of the K operand
}

__device__ __forceinline__ float block_sum_256(float v, float* red)
{
    v = ppk_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// backward of the stack update for one step; one block per active row.  Both row images (the incoming
// stack gradient and the previous stack) are staged in LDS with 16-byte loads.
__global__ __launch_bounds__(256) void k_stack_bwd(const float* __restrict__ dS_in,   // [B, D, W] carry (zeros where none)
                                                   const float* __restrict__ dXtop, long long ldx,
                                                   const float* __restrict__ Sprev,   // null at t == 0
                                                   const float* __restrict__ CS, long long ldcs,
                                                   float* __restrict__ DCS, float* __restrict__ dS_out, int D, int W)
{
    extern __shared__ __align__(16) float sm[];      // dS_t row image [D*W] (padded to 4), previous stack row [D*W] (padded), 3 + W activations
    __shared__ float red[4];
    const long long b = blockIdx.x;
    const int DW = D * W, DW4 = (DW + 3) & ~3;
    const bool vec = (DW & 3) == 0;
    float* dst = sm;
    float* spl = sm + DW4;
    float* act = sm + 2 * DW4;
    const float* cs = CS + b * ldcs;
    for (int i = threadIdx.x; i < 3 + W; i += 256) act[i] = cs[i];
    const long long base = b * (long long)DW;
    row_to_lds(dst, dS_in + base, DW, vec);
    if (Sprev) row_to_lds(spl, Sprev + base, DW, vec);
    else for (int i = threadIdx.x; i < DW; i += 256) spl[i] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < W; i += 256) dst[i] += dXtop[b * ldx + i];
    __syncthreads();
    const float a_push = act[0], a_pop = act[1], a_noop = act[2];
    float s_noop = 0.f, s_push = 0.f, s_pop = 0.f;
    auto elem = [&](int i) {
        const int d = i / W, w = i - d * W;
        const float g = dst[i];
        const float up = d == 0 ? act[3 + w] : spl[i - W];
        const float down = d < D - 1 ? spl[i + W] : 0.f;
        s_noop = fmaf(g, spl[i], s_noop);
        s_push = fmaf(g, up, s_push);
        s_pop = fmaf(g, down, s_pop);
        // gradient wrt the previous stack
        float o = a_noop * g;
        if (d + 1 < D) o = fmaf(a_push, dst[i + W], o);
        if (d >= 1) o = fmaf(a_pop, dst[i - W], o);
        return o;
    };
    if (vec) {
        for (int j = threadIdx.x; j < (DW >> 2); j += 256) {
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = elem(4 * j + k);
            reinterpret_cast<f32x4*>(dS_out + base)[j] = o;
        }
    } else {
        for (int i = threadIdx.x; i < DW; i += 256) dS_out[base + i] = elem(i);
    }
    s_noop = block_sum_256(s_noop, red);
    s_push = block_sum_256(s_push, red);
    s_pop = block_sum_256(s_pop, red);
    float* dcs = DCS + b * ldcs;
    if (threadIdx.x == 0) {
        const float dot = a_push * s_push + a_pop * s_pop + a_noop * s_noop;
        dcs[0] = a_push * (s_push - dot);
        dcs[1] = a_pop * (s_pop - dot);
        dcs[2] = a_noop * (s_noop - dot);
    }
    for (int w = threadIdx.x; w < W; w += 256) {
        const float s = act[3 + w];
        dcs[3 + w] = a_push * dst[w] * (1.0f - s * s);
    }
    for (long long w = 3 + W + threadIdx.x; w < ldcs; w += 256) dcs[w] = 0.f;     // zero padding of the K operand
}

struct SeqDims { int64_t T, B, NT, H, L, W, D, V, ks, ldcs; };

int check_dims(const SeqDims& d, const int64_t* bs, std::vector<int64_t>& off)
{
    if (d.T < 1 || d.B < 1 || d.H < 1 || d.L < 1 || d.W < 1 || d.D < 1 || d.V < 1)
        PPK_FAIL(PPK_ERR_SHAPE, "stackgru: non-positive size");
    if (d.H % 32) PPK_FAIL(PPK_ERR_UNSUP, "stackgru: rnn_cell_size must be a multiple of 32 (the host side pads with zero units)");
    if (d.ks < d.W || d.ks % 32 || d.ldcs < 3 + d.W || d.ldcs % 32)
        PPK_FAIL(PPK_ERR_ALIGN, "stackgru: ks / ldcs must cover W / 3+W and be multiples of 32");
    if ((size_t)(2 * (d.D * d.W + 4) + 3 + d.W) * 4 > 60 * 1024) PPK_FAIL(PPK_ERR_UNSUP, "stackgru: stack_depth*stack_width too large for LDS");
    if (!bs) PPK_FAIL(PPK_ERR_NULL, "stackgru: batch_sizes is null");
    off.assign((size_t)d.T + 1, 0);
    for (int64_t t = 0; t < d.T; ++t) {
        if (bs[t] < 1 || (t > 0 && bs[t] > bs[t - 1])) PPK_FAIL(PPK_ERR_SHAPE, "stackgru: batch_sizes must be positive and non-increasing");
        off[(size_t)t + 1] = off[(size_t)t] + bs[t];
    }
    if (bs[0] != d.B || off[(size_t)d.T] != d.NT) PPK_FAIL(PPK_ERR_SHAPE, "stackgru: batch_sizes inconsistent with B / NT");
    return PPK_OK;
}

ppk::NtProblem nt_store(const float* A, int64_t lda, int64_t M, const float* Bm, int64_t ldb, int64_t N, int64_t K,
                        float* C, int64_t ldc, const float* bias, float beta, int64_t H)
{
    ppk::NtProblem p{};
    p.A = A; p.B = Bm; p.lda = lda; p.ldb = ldb;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.a_rows = (int)M;
    p.epi = ppk::NT_STORE; p.H = (int)H;
    p.C = C; p.ldc = ldc; p.bias = bias; p.beta = beta;
    return p;
}

}  // namespace

extern "C" {

size_t ppk_stackgru_seq_fwd_workspace_bytes(int64_t B, int64_t H, int64_t L, int64_t W)
{
    if (B < 1 || H < 1 || L < 1 || W < 1) return 0;
    const int64_t ldcs = (3 + W + 31) / 32 * 32;
    return (size_t)L * ppk_align_up(sizeof(float) * (size_t)B * 3 * (size_t)H, 256) + ppk_align_up(sizeof(float) * (size_t)B * (size_t)ldcs, 256)
           + std::max(std::max(ppk::seq_fwd_persist_ws_bytes(B, H, L, W), ppk::seq_fwd_ws_ws_bytes(B, H, L, W)), ppk::seq_fwd_wsl_ws_bytes(B, H, L, W));
}

// table [V, 3H]  = emb . W_ih0[:, :E]^T + b_ih0      (embedding half of the layer-0 input product)
// wx0s  [3H, ks] = W_ih0[:, E:E+W], zero padded       (stack half)
// w_ih[l], b_ih[l] are read for l >= 1 only.  S0 [B, D, W]: stack entering step 0, NULL = zeros (training always
// starts from the empty stack; single-step generation carries it).
int ppk_stackgru_seq_fwd(int64_t T, const int64_t* batch_sizes_host, int64_t B, int64_t NT,
                         int64_t V, int64_t H, int64_t L, int64_t W, int64_t D,
                         const int64_t* tokens, const float* table, const float* wx0s, int64_t ks,
                         const float* const* w_ih, const float* const* w_hh,
                         const float* const* b_ih, const float* const* b_hh,
                         const float* wcs, const float* bcs, const float* h0, const float* S0,
                         float* XS, float* Hs, float* HP, float* gates, float* S,
                         float* CS, int64_t ldcs, const float* drop, float* Hdrop,
                         void* workspace, size_t ws_bytes, hipStream_t stream)
{
    SeqDims d{T, B, NT, H, L, W, D, V, ks, ldcs};
    std::vector<int64_t> off;
    int rc = check_dims(d, batch_sizes_host, off);
    if (rc) return rc;
    if (!tokens || !table || !wx0s || !w_ih || !w_hh || !b_ih || !b_hh || !wcs || !bcs || !h0 || !XS || !Hs || !HP || !gates || !S || !CS || !workspace)
        PPK_FAIL(PPK_ERR_NULL, "stackgru_seq_fwd: null pointer");
    if (ws_bytes < ppk_stackgru_seq_fwd_workspace_bytes(B, H, L, W)) PPK_FAIL(PPK_ERR_WS, "stackgru_seq_fwd: workspace too small");
    if (drop && L > 1 && !Hdrop) PPK_FAIL(PPK_ERR_NULL, "stackgru_seq_fwd: dropout factors without Hdrop");
    if (L == 1) drop = nullptr;                       // nn.GRU applies dropout between layers only
    // <= 32 rows: the whole time loop as one launch with the weights resident in registers (ppk_seq_ws.hip)
    if (ppk::seq_fwd_ws_ok(T, B, NT, H, L, W, D, ks, ldcs))
        return ppk::seq_fwd_ws(T, batch_sizes_host, B, NT, V, H, L, W, D, tokens, table, wx0s, ks, w_ih, w_hh, b_ih, b_hh,
                               wcs, bcs, h0, S0, XS, Hs, HP, gates, S, CS, ldcs, drop, Hdrop, workspace, ws_bytes, stream);
    if (ppk::seq_fwd_wsl_ok(T, B, NT, H, L, W, D, ks, ldcs))
        return ppk::seq_fwd_wsl(T, batch_sizes_host, B, NT, V, H, L, W, D, tokens, table, wx0s, ks, w_ih, w_hh, b_ih, b_hh,
                                wcs, bcs, h0, S0, XS, Hs, HP, gates, S, CS, ldcs, drop, Hdrop, workspace, ws_bytes, stream);
    // the whole time loop as one persistent launch (ppk_seq_persist.hip) when the shape allows it
    if (ppk::seq_fwd_persist_ok(T, B, NT, H, L, W, D, ks, ldcs))
        return ppk::seq_fwd_persist(T, batch_sizes_host, B, NT, V, H, L, W, D, tokens, table, wx0s, ks, w_ih, w_hh, b_ih, b_hh,
                                    wcs, bcs, h0, S0, XS, Hs, HP, gates, S, CS, ldcs, drop, Hdrop, workspace, ws_bytes, stream);
    // per-step launches: the carve below follows the ldcs the caller passed (the query assumes the minimal padding)
    if (ws_bytes < (size_t)L * ppk_align_up(sizeof(float) * (size_t)B * 3 * (size_t)H, 256) + ppk_align_up(sizeof(float) * (size_t)B * (size_t)ldcs, 256))
        PPK_FAIL(PPK_ERR_WS, "stackgru_seq_fwd: workspace too small for ldcs = %lld", (long long)ldcs);
    char* cur = (char*)workspace;
    std::vector<float*> gh((size_t)L);
    for (int64_t l = 0; l < L; ++l) gh[(size_t)l] = (float*)ppk_carve(cur, sizeof(float) * (size_t)B * 3 * H);
    float* csraw = (float*)ppk_carve(cur, sizeof(float) * (size_t)B * ldcs);
    const size_t lds_stack = sizeof(float) * (size_t)(((D * W + 3) & ~(int64_t)3) + 3 + W);
    const int64_t H3 = 3 * H;

    // Products of the hidden state entering a step that are needed one step later:
    //   csraw = [W_c ; W_s] h_{L-1} + b   feeds k_stack_fwd of the next step: a launch of its own (tiny);
    //   gh[l] = W_hh[l] h_l + b_hh[l]     rides in a gate-layout launch that has matrix work to hide it in:
    //     l <  L-1: with cell l+1 of the same step (which has just consumed h_l as its input);
    //     l == L-1: with cell 0 of the NEXT step — that launch is a K = ks (64) product whose GRU epilogue
    //               is memory-bound, so the K = H product of the top layer shares its CUs (L >= 2 only:
    //               with one layer, cell 0 itself needs gh[0] and the product goes with the heads).
    auto heads = [&](const float* hlast, int64_t M) -> int {
        ppk::NtProblem ps[2] = {nt_store(hlast, H, M, wcs, H, 3 + W, H, csraw, ldcs, bcs, 0.f, H), ppk::NtProblem{}};
        int n = 1;
        if (L == 1) ps[n++] = nt_store(hlast, H, M, w_hh[0], H, H3, H, gh[0], H3, b_hh[0], 0.f, H);
        return ppk::gemm_nt64(ps, n, stream);
    };
    // prologue: the same products of the initial hidden state
    for (int64_t l = 0; l < L && L > 1; ++l) {
        const ppk::NtProblem p = nt_store(h0 + l * B * H, H, B, w_hh[l], H, H3, H, gh[(size_t)l], H3, b_hh[l], 0.f, H);
        rc = ppk::gemm_ntF(&p, 1, stream);
        if (rc) return rc;
    }
    rc = heads(h0 + (L - 1) * B * H, B);
    if (rc) return rc;

    for (int64_t t = 0; t < T; ++t) {
        const int64_t Bt = batch_sizes_host[t], o = off[(size_t)t], op = t > 0 ? off[(size_t)t - 1] : 0;
        const int64_t Bn = t + 1 < T ? batch_sizes_host[t + 1] : 0;
        auto hprev = [&](int64_t l) { return t == 0 ? h0 + l * B * H : Hs + (l * NT + op) * H; };
        { ppk::ProfScope prof(ppk::PK_STACK_FWD, 8.0 * (double)Bt * D * W, stream);
        k_stack_fwd<<<dim3((unsigned)Bt), dim3(256), lds_stack, stream>>>(
            csraw, ldcs, t == 0 ? S0 : S + op * D * W, S + o * D * W, CS + o * ldcs, XS + o * ks, ks, (int)D, (int)W); }
        PPK_CHECK_LAUNCH("k_stack_fwd");
        for (int64_t l = 0; l < L; ++l) {
            ppk::NtProblem ps[2];
            int n = 0;
            ppk::NtProblem& c = ps[n++];
            c = ppk::NtProblem{};
            if (l == 0) { c.A = XS + o * ks; c.lda = ks; c.K = (int)ks; c.B = wx0s; c.ldb = ks; c.tok = tokens + o; c.table = table; c.V = (int)V; c.bias = nullptr; }
            else        { c.A = (drop ? Hdrop : Hs) + ((l - 1) * NT + o) * H; c.lda = H; c.K = (int)H; c.B = w_ih[l]; c.ldb = H; c.bias = b_ih[l]; }
            if (drop && l < L - 1) { c.drop = drop + (l * NT + o) * H; c.hdrop = Hdrop + (l * NT + o) * H; }
            c.M = (int)Bt; c.N = (int)H3; c.a_rows = (int)Bt; c.H = (int)H; c.epi = ppk::NT_GRU_FWD;
            c.gh = gh[(size_t)l]; c.hprev = hprev(l);
            c.hnew = Hs + (l * NT + o) * H; c.hp_out = HP + (l * NT + o) * H; c.gates_out = gates + (l * NT + o) * 4 * H;
            // the layer below has just produced h_{l-1,t}: its recurrent product for step t+1 shares this launch
            if (l > 0 && Bn > 0)
                ps[n++] = nt_store(Hs + ((l - 1) * NT + o) * H, H, Bn, w_hh[l - 1], H, H3, H, gh[(size_t)l - 1], H3, b_hh[l - 1], 0.f, H);
            // the top layer's recurrent product of the PREVIOUS step (rows of this step) shares cell 0's launch
            if (l == 0 && L > 1 && t > 0)
                ps[n++] = nt_store(Hs + ((L - 1) * NT + op) * H, H, Bt, w_hh[L - 1], H, H3, H, gh[(size_t)L - 1], H3, b_hh[L - 1], 0.f, H);
            rc = ppk::gemm_ntF(ps, n, stream);
            if (rc) return rc;
        }
        if (Bn > 0) {
            rc = heads(Hs + ((L - 1) * NT + o) * H, Bn);
            if (rc) return rc;
        }
    }
    return PPK_OK;
}

size_t ppk_stackgru_seq_bwd_workspace_bytes(int64_t B, int64_t H, int64_t L, int64_t W, int64_t D)
{
    if (B < 1 || H < 1 || L < 1 || W < 1 || D < 1) return 0;
    return std::max(std::max(2 * ppk_align_up(sizeof(float) * (size_t)B * (size_t)D * (size_t)W, 256), ppk::seq_bwd_persist_ws_bytes(B, H, L, W, D)),
                    ppk::seq_bwd_ws_ws_bytes(B, H, L, W));
}

// dHtop [NT, H]: gradient arriving at the top layer's output of every step
// (decoder: from the logits; encoder: zero except each sequence's last step).
// Outputs: dXS [NT, ks] (gradient of the stack top fed to layer 0), DGI / DGH [L, NT, 3H],
// DCS [NT, ldcs], dh0 [L, B, H].  The weight gradients are single large GEMMs over these (host side).
// Weights arrive TRANSPOSED so that every per-step data-gradient GEMM reads k-contiguous operands:
//   w_ih_t[0] [W, 3H] (stack columns of W_ih0), w_ih_t[l >= 1] [H, 3H], w_hh_t[l] [H, 3H], wcs_t [H, ldcs].
// The gate-math backward of layer l-1 runs in the epilogue of the GEMM that produces its dh
// (dgi_l . W_ih[l] for l >= 1; the stack-head product for the top layer).
int ppk_stackgru_seq_bwd(int64_t T, const int64_t* batch_sizes_host, int64_t B, int64_t NT,
                         int64_t H, int64_t L, int64_t W, int64_t D,
                         const float* const* w_ih_t, const float* const* w_hh_t, const float* wcs_t,
                         const float* HP, const float* gates, const float* S, const float* CS, int64_t ldcs,
                         const float* dHtop, float* dXS, int64_t ks, float* DGI, float* DGH, float* DCS,
                         float* dh0, const float* drop, void* workspace, size_t ws_bytes, hipStream_t stream)
{
    SeqDims d{T, B, NT, H, L, W, D, 1, ks, ldcs};
    std::vector<int64_t> off;
    int rc = check_dims(d, batch_sizes_host, off);
    if (rc) return rc;
    if (!w_ih_t || !w_hh_t || !wcs_t || !HP || !gates || !S || !CS || !dHtop || !dXS || !DGI || !DGH || !DCS || !dh0 || !workspace)
        PPK_FAIL(PPK_ERR_NULL, "stackgru_seq_bwd: null pointer");
    if (ws_bytes < ppk_stackgru_seq_bwd_workspace_bytes(B, H, L, W, D)) PPK_FAIL(PPK_ERR_WS, "stackgru_seq_bwd: workspace too small");
    if (L == 1) drop = nullptr;
    // <= 32 rows: one launch with the transposed weights resident in registers (ppk_seq_ws.hip)
    if (ppk::seq_bwd_ws_ok(T, B, NT, H, L, W, D, ks, ldcs))
        return ppk::seq_bwd_ws(T, batch_sizes_host, B, NT, H, L, W, D, w_ih_t, w_hh_t, wcs_t, HP, gates, S, CS, ldcs,
                               dHtop, dXS, ks, DGI, DGH, DCS, dh0, drop, workspace, ws_bytes, stream);
    // the whole time loop as one persistent launch (ppk_seq_persist_bwd.hip) when the shape allows it
    if (ppk::seq_bwd_persist_ok(T, B, NT, H, L, W, D, ks, ldcs))
        return ppk::seq_bwd_persist(T, batch_sizes_host, B, NT, H, L, W, D, w_ih_t, w_hh_t, wcs_t, HP, gates, S, CS, ldcs,
                                    dHtop, dXS, ks, DGI, DGH, DCS, dh0, drop, workspace, ws_bytes, stream);
    char* cur = (char*)workspace;
    const size_t sbytes = sizeof(float) * (size_t)B * D * W;
    float* dSa = (float*)ppk_carve(cur, sbytes);
    float* dSb = (float*)ppk_carve(cur, sbytes);
    PPK_HIP(hipMemsetAsync(dSa, 0, sbytes, stream));
    PPK_HIP(hipMemsetAsync(dSb, 0, sbytes, stream));
    PPK_HIP(hipMemsetAsync(dh0, 0, sizeof(float) * (size_t)L * B * H, stream));
    const size_t lds_stack = sizeof(float) * (size_t)(2 * ((D * W + 3) & ~(int64_t)3) + 3 + W);
    const int64_t H3 = 3 * H;

    auto gru_bwd_problem = [&](const float* A, int64_t lda, int64_t a_rows, const float* Bm, int64_t ldb, int64_t K,
                               int64_t l, int64_t Bt, int64_t o, const float* ext) {
        ppk::NtProblem p{};
        p.A = A; p.lda = lda; p.B = Bm; p.ldb = ldb;
        p.M = (int)Bt; p.N = (int)H; p.K = (int)K; p.a_rows = (int)a_rows; p.H = (int)H;
        p.epi = ppk::NT_GRU_BWD;
        p.carry = dh0 + l * B * H; p.ext = ext;
        p.gates_in = gates + (l * NT + o) * 4 * H; p.hp_in = HP + (l * NT + o) * H;
        p.dgi = DGI + (l * NT + o) * H3; p.dgh = DGH + (l * NT + o) * H3;
        return p;
    };

    for (int64_t t = T - 1; t >= 0; --t) {
        const int64_t Bt = batch_sizes_host[t], o = off[(size_t)t], op = t > 0 ? off[(size_t)t - 1] : 0;
        const int64_t Bn = t + 1 < T ? batch_sizes_host[t + 1] : 0, on = t + 1 < T ? off[(size_t)t + 1] : 0;
        // top layer: dh = carry + dHtop_t + DCS_{t+1} . [W_c ; W_s]   (the stack of step t+1 read hidden[-1] of step t)
        {
            const ppk::NtProblem p = gru_bwd_problem(DCS + on * ldcs, ldcs, Bn, wcs_t, ldcs, ldcs, L - 1, Bt, o, dHtop + o * H);
            rc = ppk::gemm_nt64(&p, 1, stream);
            if (rc) return rc;
        }
        for (int64_t l = L - 1; l >= 0; --l) {
            const float* dgi = DGI + (l * NT + o) * H3;
            const float* dgh = DGH + (l * NT + o) * H3;
            ppk::NtProblem ps[2];
            ps[0] = nt_store(dgh, H3, Bt, w_hh_t[l], H3, H, H3, dh0 + l * B * H, H, nullptr, 1.f, H);      // carry_l += dgh . W_hh
            if (l > 0) {
                ps[1] = gru_bwd_problem(dgi, H3, Bt, w_ih_t[l], H3, H3, l - 1, Bt, o, nullptr);            // dh_{l-1} = carry + (dgi . W_ih) * drop_{l-1}
                if (drop) ps[1].drop = drop + ((l - 1) * NT + o) * H;
            }
            else       ps[1] = nt_store(dgi, H3, Bt, w_ih_t[0], H3, W, H3, dXS + o * ks, ks, nullptr, 0.f, H);
            rc = ppk::gemm_nt64(ps, 2, stream);
            if (rc) return rc;
        }
        { ppk::ProfScope prof(ppk::PK_STACK_BWD, 12.0 * (double)Bt * D * W, stream);
        k_stack_bwd<<<dim3((unsigned)Bt), dim3(256), lds_stack, stream>>>(
            dSa, dXS + o * ks, ks, t == 0 ? nullptr : S + op * D * W, CS + o * ldcs, ldcs, DCS + o * ldcs, dSb, (int)D, (int)W); }
        PPK_CHECK_LAUNCH("k_stack_bwd");
        float* tmp = dSa; dSa = dSb; dSb = tmp;
    }
    // the stack of step 0 read the initial hidden state of the top layer
    const ppk::NtProblem p = nt_store(DCS, ldcs, B, wcs_t, ldcs, H, ldcs, dh0 + (L - 1) * B * H, H, nullptr, 1.f, H);
    return ppk::gemm_nt64(&p, 1, stream);
}

// Launch schedule of the sequence entry points: -1 = auto (default: the persistent kernels where they are faster —
// ppk::seq_auto_persistent: one or two well-filled machine-fulls of workgroups, e.g. H 512 from batch 1024 up —,
// the per-step launches elsewhere), 0 = never persistent,
// 1 = persistent whenever the shape is supported, at any batch; any other value only queries.  Returns the value in force.
int ppk_seq_persist_mode(int mode) { return ppk::seq_persist_mode(mode); }
int ppk_mfma_mode(int mode) { return ppk::mfma_mode(mode); }

}  // extern "C"
