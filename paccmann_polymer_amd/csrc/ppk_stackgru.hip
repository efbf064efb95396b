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
//   X     [NT, ldx]   layer-0 input  [embedding | stack top]
//   Hs    [L, NT, H]  hidden after the step          HP [L, NT, H] hidden before it
//   gates [L, NT, 4H] r, z, n, (W_hn h + b_hn)
//   S     [NT, D, W]  stack after the step           CS [NT, ldcs] softmax(3) | tanh(W)
// The gate contractions are fp32-MFMA GEMMs (ppk_gemm.hip) launched per step:
// at batch 2048 each is ~3 GFLOP, far above the ~1.5 us kernel boundary, while a
// persistent LDS-resident-weights kernel would pay a 4-6 us grid barrier per
// layer-step (MI355X_MICROARCH.md price list) — see DESIGN.md.
#include "ppk_common.h"
#include "ppk_kernels.h"
#include <vector>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// X[row, 0:E] = emb[tok[row], :]
__global__ __launch_bounds__(256) void k_embed(const int64_t* __restrict__ tok, const float* __restrict__ emb,
                                               int V, int E, long long NT, float* __restrict__ X, long long ldx)
{
    const long long total = NT * E;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long row = e / E;
        const int c = (int)(e % E);
        const long long t = tok[row];
        X[row * ldx + c] = (t >= 0 && t < V) ? emb[t * E + c] : 0.0f;
    }
}

// one block per active row: stack update (Appendix B) + stack top into X
__global__ __launch_bounds__(256) void k_stack_fwd(const float* __restrict__ csraw, long long ldcs,
                                                   const float* __restrict__ Sprev,   // null at t == 0
                                                   float* __restrict__ Snew, float* __restrict__ CS,
                                                   float* __restrict__ Xtop, long long ldx, int D, int W)
{
    extern __shared__ float sm[];      // [0..2] controls, [3..3+W) s_in
    const long long b = blockIdx.x;
    const float* raw = csraw + b * ldcs;
    if (threadIdx.x == 0) {
        float p0 = raw[0], p1 = raw[1], p2 = raw[2];
        float m = fmaxf(p0, fmaxf(p1, p2));
        float e0 = expf(p0 - m), e1 = expf(p1 - m), e2 = expf(p2 - m);
        float inv = 1.0f / (e0 + e1 + e2);
        sm[0] = e0 * inv; sm[1] = e1 * inv; sm[2] = e2 * inv;
    }
    for (int w = threadIdx.x; w < W; w += 256) sm[3 + w] = tanhf(raw[3 + w]);
    __syncthreads();
    const float a_push = sm[0], a_pop = sm[1], a_noop = sm[2];
    float* cs = CS + b * ldcs;
    for (int i = threadIdx.x; i < 3 + W; i += 256) cs[i] = sm[i];
    const long long base = b * (long long)D * W;
    const float* sp = Sprev ? Sprev + base : nullptr;
    float* sn = Snew + base;
    for (int i = threadIdx.x; i < D * W; i += 256) {
        const int d = i / W, w = i - d * W;
        float cur = 0.f, up, down = 0.f;
        if (sp) {
            cur = sp[i];
            up = d == 0 ? sm[3 + w] : sp[i - W];
            if (d < D - 1) down = sp[i + W];
        } else {
            up = d == 0 ? sm[3 + w] : 0.f;
        }
        const float v = a_noop * cur + a_push * up + a_pop * down;
        sn[i] = v;
        if (d == 0) Xtop[b * ldx + w] = v;
    }
}

// GRU gate math for one layer-step: rows x H, float4 over H.
__global__ __launch_bounds__(256) void k_gate_fwd(const float* __restrict__ gi, const float* __restrict__ gh,
                                                  const float* __restrict__ hprev, int rows, int H,
                                                  float* __restrict__ hnew, float* __restrict__ hp_out,
                                                  float* __restrict__ gates)
{
    const int H4 = H >> 2;
    const long long total = (long long)rows * H4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long row = e / H4;
        const int j = (int)(e % H4) << 2;
        const float* gir = gi + row * 3 * H;
        const float* ghr = gh + row * 3 * H;
        const f32x4 ir = *(const f32x4*)(gir + j), iz = *(const f32x4*)(gir + H + j), in_ = *(const f32x4*)(gir + 2 * H + j);
        const f32x4 hr = *(const f32x4*)(ghr + j), hz = *(const f32x4*)(ghr + H + j), hn = *(const f32x4*)(ghr + 2 * H + j);
        const f32x4 hp = *(const f32x4*)(hprev + row * H + j);
        f32x4 r, z, n, h;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r[k] = sigmoidf_(ir[k] + hr[k]);
            z[k] = sigmoidf_(iz[k] + hz[k]);
            n[k] = tanhf(in_[k] + r[k] * hn[k]);
            h[k] = (1.0f - z[k]) * n[k] + z[k] * hp[k];
        }
        *(f32x4*)(hnew + row * H + j) = h;
        *(f32x4*)(hp_out + row * H + j) = hp;
        float* g = gates + row * 4 * H;
        *(f32x4*)(g + j) = r;
        *(f32x4*)(g + H + j) = z;
        *(f32x4*)(g + 2 * H + j) = n;
        *(f32x4*)(g + 3 * H + j) = hn;
    }
}

// backward of the gate math.  dh = carry + ext + dxup ; writes DGI, DGH and carry = dh * z
__global__ __launch_bounds__(256) void k_gate_bwd(float* __restrict__ carry, const float* __restrict__ ext,
                                                  const float* __restrict__ dxup, const float* __restrict__ gates,
                                                  const float* __restrict__ hp, int rows, int H,
                                                  float* __restrict__ dgi, float* __restrict__ dgh)
{
    const int H4 = H >> 2;
    const long long total = (long long)rows * H4;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long row = e / H4;
        const int j = (int)(e % H4) << 2;
        f32x4 dh = *(const f32x4*)(carry + row * H + j);
        if (ext) dh += *(const f32x4*)(ext + row * H + j);
        if (dxup) dh += *(const f32x4*)(dxup + row * H + j);
        const float* g = gates + row * 4 * H;
        const f32x4 r = *(const f32x4*)(g + j), z = *(const f32x4*)(g + H + j);
        const f32x4 n = *(const f32x4*)(g + 2 * H + j), hn = *(const f32x4*)(g + 3 * H + j);
        const f32x4 hprev = *(const f32x4*)(hp + row * H + j);
        f32x4 dr, dz, dn, dnr, c;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            dn[k] = dh[k] * (1.0f - z[k]) * (1.0f - n[k] * n[k]);
            dz[k] = dh[k] * (hprev[k] - n[k]) * z[k] * (1.0f - z[k]);
            dr[k] = dn[k] * hn[k] * r[k] * (1.0f - r[k]);
            dnr[k] = dn[k] * r[k];
            c[k] = dh[k] * z[k];
        }
        float* a = dgi + row * 3 * H;
        float* b = dgh + row * 3 * H;
        *(f32x4*)(a + j) = dr;  *(f32x4*)(a + H + j) = dz;  *(f32x4*)(a + 2 * H + j) = dn;
        *(f32x4*)(b + j) = dr;  *(f32x4*)(b + H + j) = dz;  *(f32x4*)(b + 2 * H + j) = dnr;
        *(f32x4*)(carry + row * H + j) = c;
    }
}

__device__ __forceinline__ float block_sum_256(float v, float* red)
{
    v = ppk_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// backward of the stack update for one step; one block per active row.
__global__ __launch_bounds__(256) void k_stack_bwd(const float* __restrict__ dS_in,   // [B, D, W] carry (zeros where none)
                                                   const float* __restrict__ dXtop, long long ldx,
                                                   const float* __restrict__ Sprev,   // null at t == 0
                                                   const float* __restrict__ CS, long long ldcs,
                                                   float* __restrict__ DCS, float* __restrict__ dS_out, int D, int W)
{
    extern __shared__ float sm[];      // dS_t row image [D*W] + 3 + W activations
    __shared__ float red[4];
    const long long b = blockIdx.x;
    const int DW = D * W;
    float* dst = sm;
    float* act = sm + DW;
    const float* cs = CS + b * ldcs;
    for (int i = threadIdx.x; i < 3 + W; i += 256) act[i] = cs[i];
    const long long base = b * (long long)DW;
    for (int i = threadIdx.x; i < DW; i += 256) {
        float v = dS_in[base + i];
        if (i < W) v += dXtop[b * ldx + i];
        dst[i] = v;
    }
    __syncthreads();
    const float a_push = act[0], a_pop = act[1], a_noop = act[2];
    const float* sp = Sprev ? Sprev + base : nullptr;
    float s_noop = 0.f, s_push = 0.f, s_pop = 0.f;
    for (int i = threadIdx.x; i < DW; i += 256) {
        const int d = i / W, w = i - d * W;
        const float g = dst[i];
        float cur = 0.f, up, down = 0.f;
        if (sp) {
            cur = sp[i];
            up = d == 0 ? act[3 + w] : sp[i - W];
            if (d < D - 1) down = sp[i + W];
        } else {
            up = d == 0 ? act[3 + w] : 0.f;
        }
        s_noop = fmaf(g, cur, s_noop);
        s_push = fmaf(g, up, s_push);
        s_pop = fmaf(g, down, s_pop);
        // gradient wrt the previous stack
        float o = a_noop * g;
        if (d + 1 < D) o = fmaf(a_push, dst[i + W], o);
        if (d >= 1) o = fmaf(a_pop, dst[i - W], o);
        dS_out[base + i] = o;
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
}

struct SeqDims { int64_t T, B, NT, E, H, L, W, D, V, ldx, ldcs; };

int check_dims(const SeqDims& d, const int64_t* bs, std::vector<int64_t>& off)
{
    if (d.T < 1 || d.B < 1 || d.E < 1 || d.H < 4 || d.L < 1 || d.W < 1 || d.D < 1 || d.V < 1)
        PPK_FAIL(PPK_ERR_SHAPE, "stackgru: non-positive size");
    if (d.H % 4) PPK_FAIL(PPK_ERR_UNSUP, "stackgru: rnn_cell_size must be a multiple of 4");
    if (d.ldx < d.E + d.W || d.ldx % 4 || d.ldcs < 3 + d.W || d.ldcs % 4)
        PPK_FAIL(PPK_ERR_ALIGN, "stackgru: ldx/ldcs must cover E+W / 3+W and be multiples of 4");
    if ((size_t)(d.D * d.W + 3 + d.W) * 4 > 60 * 1024) PPK_FAIL(PPK_ERR_UNSUP, "stackgru: stack_depth*stack_width too large for LDS");
    if (!bs) PPK_FAIL(PPK_ERR_NULL, "stackgru: batch_sizes is null");
    off.assign((size_t)d.T + 1, 0);
    for (int64_t t = 0; t < d.T; ++t) {
        if (bs[t] < 1 || (t > 0 && bs[t] > bs[t - 1])) PPK_FAIL(PPK_ERR_SHAPE, "stackgru: batch_sizes must be positive and non-increasing");
        off[(size_t)t + 1] = off[(size_t)t] + bs[t];
    }
    if (bs[0] != d.B || off[(size_t)d.T] != d.NT) PPK_FAIL(PPK_ERR_SHAPE, "stackgru: batch_sizes inconsistent with B / NT");
    return PPK_OK;
}

inline unsigned grid_for(long long work, int per_block = 256, unsigned cap = 4096)
{
    long long b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (unsigned)(b < cap ? b : cap);
}

}  // namespace

extern "C" {

size_t ppk_stackgru_seq_fwd_workspace_bytes(int64_t B, int64_t H, int64_t W)
{
    if (B < 1 || H < 1 || W < 1) return 0;
    const int64_t ldcs = (3 + W + 3) / 4 * 4;
    return 2 * ppk_align_up(sizeof(float) * (size_t)B * 3 * (size_t)H, 256) + ppk_align_up(sizeof(float) * (size_t)B * (size_t)ldcs, 256);
}

int ppk_stackgru_seq_fwd(int64_t T, const int64_t* batch_sizes_host, int64_t B, int64_t NT,
                         int64_t V, int64_t E, int64_t H, int64_t L, int64_t W, int64_t D,
                         const int64_t* tokens, const float* emb,
                         const float* const* w_ih, int64_t ld_wih0, const float* const* w_hh,
                         const float* const* b_ih, const float* const* b_hh,
                         const float* wcs, const float* bcs, const float* h0,
                         float* X, int64_t ldx, float* Hs, float* HP, float* gates, float* S,
                         float* CS, int64_t ldcs, void* workspace, size_t ws_bytes, hipStream_t stream)
{
    SeqDims d{T, B, NT, E, H, L, W, D, V, ldx, ldcs};
    std::vector<int64_t> off;
    int rc = check_dims(d, batch_sizes_host, off);
    if (rc) return rc;
    if (!tokens || !emb || !w_ih || !w_hh || !b_ih || !b_hh || !wcs || !bcs || !h0 || !X || !Hs || !HP || !gates || !S || !CS || !workspace)
        PPK_FAIL(PPK_ERR_NULL, "stackgru_seq_fwd: null pointer");
    if (ws_bytes < ppk_stackgru_seq_fwd_workspace_bytes(B, H, W)) PPK_FAIL(PPK_ERR_WS, "stackgru_seq_fwd: workspace too small");
    if (ld_wih0 < E + W) PPK_FAIL(PPK_ERR_SHAPE, "stackgru_seq_fwd: ld_wih0 < E + W");
    char* cur = (char*)workspace;
    float* gi = (float*)ppk_carve(cur, sizeof(float) * (size_t)B * 3 * H);
    float* gh = (float*)ppk_carve(cur, sizeof(float) * (size_t)B * 3 * H);
    float* csraw = (float*)ppk_carve(cur, sizeof(float) * (size_t)B * ldcs);
    const int64_t I0 = E + W;
    const size_t lds_stack = sizeof(float) * (size_t)(3 + W);

    k_embed<<<dim3(grid_for(NT * E)), dim3(256), 0, stream>>>(tokens, emb, (int)V, (int)E, NT, X, ldx);
    PPK_CHECK_LAUNCH("k_embed");
    for (int64_t t = 0; t < T; ++t) {
        const int64_t Bt = batch_sizes_host[t], o = off[(size_t)t], op = t > 0 ? off[(size_t)t - 1] : 0;
        auto hprev = [&](int64_t l) { return t == 0 ? h0 + l * B * H : Hs + (l * NT + op) * H; };
        rc = ppk::gemm(0, 1, Bt, 3 + W, H, 1.f, hprev(L - 1), H, wcs, H, 0.f, csraw, ldcs, bcs, nullptr, 0, stream);
        if (rc) return rc;
        { ppk::ProfScope prof(ppk::PK_STACK_FWD, 8.0 * (double)Bt * D * W, stream);
        k_stack_fwd<<<dim3((unsigned)Bt), dim3(256), lds_stack, stream>>>(
            csraw, ldcs, t == 0 ? nullptr : S + op * D * W, S + o * D * W, CS + o * ldcs, X + o * ldx + E, ldx, (int)D, (int)W); }
        PPK_CHECK_LAUNCH("k_stack_fwd");
        for (int64_t l = 0; l < L; ++l) {
            const float* in = l == 0 ? X + o * ldx : Hs + ((l - 1) * NT + o) * H;
            const int64_t Il = l == 0 ? I0 : H, ldin = l == 0 ? ldx : H;
            // both gate GEMMs of the layer-step in one launch
            const ppk::GemmProblem pi{Bt, 3 * H, Il, 1.f, in, ldin, w_ih[l], l == 0 ? ld_wih0 : H, 0.f, gi, 3 * H, b_ih[l]};
            const ppk::GemmProblem ph{Bt, 3 * H, H, 1.f, hprev(l), H, w_hh[l], H, 0.f, gh, 3 * H, b_hh[l]};
            rc = ppk::gemm_nt_pair(pi, ph, stream);
            if (rc) return rc;
            { ppk::ProfScope prof(ppk::PK_GATE_FWD, 4.0 * 13 * (double)Bt * H, stream);
            k_gate_fwd<<<dim3(grid_for(Bt * H / 4)), dim3(256), 0, stream>>>(
                gi, gh, hprev(l), (int)Bt, (int)H, Hs + (l * NT + o) * H, HP + (l * NT + o) * H, gates + (l * NT + o) * 4 * H); }
            PPK_CHECK_LAUNCH("k_gate_fwd");
        }
    }
    return PPK_OK;
}

size_t ppk_stackgru_seq_bwd_workspace_bytes(int64_t B, int64_t H, int64_t W, int64_t D)
{
    if (B < 1 || H < 1 || W < 1 || D < 1) return 0;
    return 2 * ppk_align_up(sizeof(float) * (size_t)B * (size_t)D * (size_t)W, 256) + ppk_align_up(sizeof(float) * (size_t)B * (size_t)H, 256);
}

// dHtop [NT, H]: gradient arriving at the top layer's output of every step
// (decoder: from the logits; encoder: zero except each sequence's last step).
// Outputs: dX [NT, ldx], DGI / DGH [L, NT, 3H], DCS [NT, ldcs], dh0 [L, B, H].
// The weight gradients are single large GEMMs over these (host side).
// Weights arrive TRANSPOSED (w_ih_t[l] [I_l, 3H], w_hh_t[l] [H, 3H], wcs_t [H, ldcs])
// so that every per-step data-gradient GEMM reads k-contiguous operands.
int ppk_stackgru_seq_bwd(int64_t T, const int64_t* batch_sizes_host, int64_t B, int64_t NT,
                         int64_t V, int64_t E, int64_t H, int64_t L, int64_t W, int64_t D,
                         const float* const* w_ih_t, const float* const* w_hh_t, const float* wcs_t,
                         const float* HP, const float* gates, const float* S, const float* CS, int64_t ldcs,
                         const float* dHtop, float* dX, int64_t ldx, float* DGI, float* DGH, float* DCS,
                         float* dh0, void* workspace, size_t ws_bytes, hipStream_t stream)
{
    SeqDims d{T, B, NT, E, H, L, W, D, V, ldx, ldcs};
    std::vector<int64_t> off;
    int rc = check_dims(d, batch_sizes_host, off);
    if (rc) return rc;
    if (!w_ih_t || !w_hh_t || !wcs_t || !HP || !gates || !S || !CS || !dHtop || !dX || !DGI || !DGH || !DCS || !dh0 || !workspace)
        PPK_FAIL(PPK_ERR_NULL, "stackgru_seq_bwd: null pointer");
    if (ws_bytes < ppk_stackgru_seq_bwd_workspace_bytes(B, H, W, D)) PPK_FAIL(PPK_ERR_WS, "stackgru_seq_bwd: workspace too small");
    char* cur = (char*)workspace;
    const size_t sbytes = sizeof(float) * (size_t)B * D * W;
    float* dSa = (float*)ppk_carve(cur, sbytes);
    float* dSb = (float*)ppk_carve(cur, sbytes);
    float* dxup = (float*)ppk_carve(cur, sizeof(float) * (size_t)B * H);
    PPK_HIP(hipMemsetAsync(dSa, 0, sbytes, stream));
    PPK_HIP(hipMemsetAsync(dSb, 0, sbytes, stream));
    PPK_HIP(hipMemsetAsync(dh0, 0, sizeof(float) * (size_t)L * B * H, stream));
    const int64_t I0 = E + W;
    const size_t lds_stack = sizeof(float) * (size_t)(D * W + 3 + W);

    for (int64_t t = T - 1; t >= 0; --t) {
        const int64_t Bt = batch_sizes_host[t], o = off[(size_t)t], op = t > 0 ? off[(size_t)t - 1] : 0;
        for (int64_t l = L - 1; l >= 0; --l) {
            float* carry = dh0 + l * B * H;
            float* dgi = DGI + (l * NT + o) * 3 * H;
            float* dgh = DGH + (l * NT + o) * 3 * H;
            { ppk::ProfScope prof(ppk::PK_GATE_BWD, 4.0 * 14 * (double)Bt * H, stream);
            k_gate_bwd<<<dim3(grid_for(Bt * H / 4)), dim3(256), 0, stream>>>(
                carry, l == L - 1 ? dHtop + o * H : nullptr, l < L - 1 ? dxup : nullptr,
                gates + (l * NT + o) * 4 * H, HP + (l * NT + o) * H, (int)Bt, (int)H, dgi, dgh); }
            PPK_CHECK_LAUNCH("k_gate_bwd");
            // carry += dgh . W_hh  and  dx = dgi . W_ih  (pre-transposed weights -> k-contiguous operands) in one launch
            const ppk::GemmProblem pc{Bt, H, 3 * H, 1.f, dgh, 3 * H, w_hh_t[l], 3 * H, 1.f, carry, H, nullptr};
            const ppk::GemmProblem px = l > 0
                ? ppk::GemmProblem{Bt, H, 3 * H, 1.f, dgi, 3 * H, w_ih_t[l], 3 * H, 0.f, dxup, H, nullptr}
                : ppk::GemmProblem{Bt, I0, 3 * H, 1.f, dgi, 3 * H, w_ih_t[0], 3 * H, 0.f, dX + o * ldx, ldx, nullptr};
            rc = ppk::gemm_nt_pair(pc, px, stream);
            if (rc) return rc;
        }
        { ppk::ProfScope prof(ppk::PK_STACK_BWD, 12.0 * (double)Bt * D * W, stream);
        k_stack_bwd<<<dim3((unsigned)Bt), dim3(256), lds_stack, stream>>>(
            dSa, dX + o * ldx + E, ldx, t == 0 ? nullptr : S + op * D * W, CS + o * ldcs, ldcs, DCS + o * ldcs, dSb, (int)D, (int)W); }
        PPK_CHECK_LAUNCH("k_stack_bwd");
        float* tmp = dSa; dSa = dSb; dSb = tmp;
        rc = ppk::gemm(0, 1, Bt, H, 3 + W, 1.f, DCS + o * ldcs, ldcs, wcs_t, ldcs, 1.f, dh0 + (L - 1) * B * H, H, nullptr, nullptr, 0, stream);
        if (rc) return rc;
    }
    return PPK_OK;
}

}  // extern "C"
