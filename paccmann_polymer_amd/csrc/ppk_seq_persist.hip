// K1 forward as ONE persistent launch: the whole packed StackGRU time loop
// (paccmann_chemistry StackGRU reached from paccmann_polymer/models/paccmann_encoder.py:166,168;
// semantics SURVEY.md Appendix B) without a kernel boundary per layer-step.
//
// The recurrence is sequential in TIME but independent across batch ROWS.  Rows are cut into
// "m-tiles" of 32 consecutive rows of the length-sorted batch; a GROUP owns two m-tiles — one from
// the long end and one from the short end of the batch (m-tiles g and 2G-1-g: every group carries
// the same number of row-steps) — and is served by H/32 workgroups, one per 32 hidden units
// ("n-tile": r, z, n of the same units, so the GRU cell runs in the epilogue).  A workgroup only
// ever waits for the other workgroups of ITS group (a counter per group and phase); groups drift
// freely, two workgroups of different groups share a CU, and one's epilogue / stack update / wait
// overlaps the other's MFMA loop.  There is no grid-wide barrier anywhere.
//
// Per step and layer l a workgroup computes, for its units u and the group's active rows,
//   acc_r, acc_z += W_hh[l] h_l(t-1)            "part A": operands complete one phase earlier,
//   acc_nh       += W_hn[l] h_l(t-1)              so this runs while the group's slower members finish
//   acc_r, acc_z, acc_ni += W_ih[l] x_l(t)      "part B": x_0 = stack top (K = ks), x_l = h_{l-1}(t)
// with fp32 MFMA 32x32x2.  K tiles of 32 are staged for the whole workgroup by LDS-DMA
// (global_load_lds, source-side XOR swizzle, one s_barrier per K tile; 3 buffers / two stages in flight with two
// active m-tiles, 4 buffers / three in flight with one); inside a K tile the
// four waves split K (wave w takes the 8-wide chunk w) and each wave accumulates ALL active m-tiles x
// gates: when the short m-tile of a group has ended, every wave still has work (half of it) instead
// of two waves idling.  The four partial tiles are summed in wave order in the epilogue (fixed
// order: bitwise reproducible).  After the top layer every workgroup multiplies ITS 32 units of the new
// hidden tile with its slice of the transposed head weights (kept in LDS; one 32 x 32 MFMA tile per
// wave) and hands that partial to the row's static owner (row rho belongs to n-tile rho mod ntiles); the
// owner adds the partials in tile order, applies softmax / tanh and the push / pop / no-op blend to its
// stack rows (requested before it waits for the partials) and publishes the new stack tops.
//
// Cross-workgroup hand-offs (hidden tiles, stack tops) follow MI355X_MICROARCH.md's valid forms:
// producer = write-through (sc1) 16-byte stores, every wave drains vmcnt, workgroup barrier, ONE
// agent-scope counter add; consumer = ONE lane polls (relaxed, agent), agent acquire (buffer_inv sc1),
// vmcnt(0), workgroup barrier, then plain loads / LDS-DMA.  Every hand-off line is written once
// before any reader touches it and by exactly one workgroup (128-byte row segments).  Polls are
// 32 x 64 cycles apart on counters that sit 4 KB apart (ppk_seq_common.h) and bounded: on a timeout (~3 s) the kernel raises a status
// word, every workgroup leaves at its next wait and a NaN is written into the result.
#include "ppk_common.h"
#include "ppk_kernels.h"
#include "ppk_seq_common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

using namespace ppk_seq;

constexpr int MAXT = ppk::SEQP_MAXT, MAXL = ppk::SEQP_MAXL;
constexpr int REGION = 4 * 4 * EPI_TILE;       // floats: 4 waves x 4 accumulators, transposed (also staging / stack scratch)
constexpr int WL_MAX = 32 * 64;                // floats: this workgroup's slice of the transposed head weights (ldcs <= 64)

struct SeqArgs {
    int T, B, H, L, W, D, V, ks, ldcs, ntiles, G, MT, g0, g1, tpx, xt, rb, pair, hgrid;
    long long NT;
    const int64_t* tokens;
    const float* table; const float* wx0s;
    const float* w_ih[MAXL]; const float* w_hh[MAXL]; const float* b_ih[MAXL]; const float* b_hh[MAXL];
    const float* wcs_t; const float* bcs; const float* h0; const float* S0;
    float* XS; float* Hs; float* HP; float* gates; float* S; float* CS;
    const float* drop; float* HD;      // inter-layer dropout (training mode of nn.GRU(dropout = p)): factors [L-1, NT, H] (0 or 1 / (1 - p)),
                                       // HD [L-1, NT, H] = Hs * drop = what the layer above reads; both null without dropout
    float* hpart;
    int* ctr; int* status;
    long long* dbg;
    int bs[MAXT];
};

// LDS of the kernel (file scope so that the non-inlined phases address it with DS instructions, not through flat pointers)
__shared__ __align__(1024) float g_lds[REGION];      // staging buffers | transposed accumulator partials | stack-phase scratch
__shared__ __align__(16) float g_wl[WL_MAX];         // rows 32 j .. 32 j + 31 of wcs_t [H, ldcs]: constant for the workgroup
__shared__ int g_off[MAXT + 1];
__shared__ int g_flag;
#ifdef PPK_SEQ_TIMING
__shared__ long long g_tacc[16];
__shared__ long long g_tlast;
#endif

// PPK_SEQ_TIMING build (tools/seq_timing.py): thread 0 adds the shader-clock cycles of every phase into LDS slots
#ifdef PPK_SEQ_TIMING
#define TSTAMP(id) do { if (threadIdx.x == 0) { const long long now_ = __builtin_readcyclecounter(); g_tacc[id] += now_ - g_tlast; g_tlast = now_; } } while (0)
#else
#define TSTAMP(id) do { } while (0)
#endif

// ---- K loop ---------------------------------------------------------------------------------
// acc[s][q] += A_s[32 x 32 nk] . B_q[32 x 32 nk]^T for the NA active m-tiles s and the gates q = r, z, n (n goes to
// accumulator NSEL: 2 = input side, 3 = hidden side); the wave takes chunk `wave` of every K tile.
// src[0..1]: this lane's source of its LDS-DMA piece of A slot 0 / 1, src[2..4]: of B gate r / z / n.
template <int NA, int NSEL, bool BF>
__device__ __forceinline__ void kloop(float* lds, const float* s0, const float* s1, const float* s2, const float* s3, const float* s4,
                                      int nk, f32x16 (&acc)[2][4], int wave, int lane)
{
    // buffer = the NA A blocks followed by the three B blocks; with one active m-tile the region holds FOUR buffers
    // (three stages in flight behind the one barrier per stage), with two m-tiles three (two in flight)
    constexpr int PPW = 3 + NA;
    constexpr int SB = (3 + NA) * 1024;               // floats per buffer
    constexpr int NB = NA == 1 ? 4 : 3;
    static_assert(NB * SB <= REGION, "staging fits the shared region");
    const int li = lane & 31, h = lane >> 5;
    const int xo = li * BK + 4 * ((2 * wave + h) ^ ((li >> 1) & 7));
    auto stage = [&](int buf) {
#ifdef PPK_EXP_NODMA
        return;
#endif
        float* base = lds + buf * SB + wave * 256;
        __builtin_amdgcn_global_load_lds(s0, PPK_LDS_PTR(base), 16, 0, DMA_AUX_HANDOFF); s0 += BK;
        if constexpr (NA == 2) { __builtin_amdgcn_global_load_lds(s1, PPK_LDS_PTR(base + 1024), 16, 0, DMA_AUX_HANDOFF); s1 += BK; }
        __builtin_amdgcn_global_load_lds(s2, PPK_LDS_PTR(base + NA * 1024), 16, 0, 0); s2 += BK;
        __builtin_amdgcn_global_load_lds(s3, PPK_LDS_PTR(base + (NA + 1) * 1024), 16, 0, 0); s3 += BK;
        __builtin_amdgcn_global_load_lds(s4, PPK_LDS_PTR(base + (NA + 2) * 1024), 16, 0, 0); s4 += BK;
    };
    __syncthreads();                                  // the shared region is free (previous epilogue / stack phase done)
    // (refilling a buffer one stage earlier behind a SECOND barrier pays in the backward's leaner stages,
    // ppk_seq_persist_bwd.hip, but measured slower here: 16.9 against 15.9 ms in a same-box A/B)
#pragma unroll
    for (int p = 0; p < NB - 1; ++p) if (nk > p) stage(p);
    int cur = 0, nxt = NB - 1;
    for (int it = 0; it < nk; ++it) {
        const int rem = nk - 1 - it;                  // stages issued after this one
        if constexpr (NB == 4) { if (rem >= 2) wait_vmcnt<2 * PPW>(); else if (rem == 1) wait_vmcnt<PPW>(); else wait_vmcnt<0>(); }
        else { if (rem >= 1) wait_vmcnt<PPW>(); else wait_vmcnt<0>(); }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + NB - 1 < nk) stage(nxt);
        nxt = nxt == NB - 1 ? 0 : nxt + 1;
        const float* base = lds + cur * SB + xo;
        Frag<BF> a[NA], b[3];
#pragma unroll
        for (int s = 0; s < NA; ++s) a[s].set(*(const f32x4*)(base + s * 1024));
#pragma unroll
        for (int q = 0; q < 3; ++q) b[q].set(*(const f32x4*)(base + (NA + q) * 1024));
#ifdef PPK_EXP_NOMMA
        asm volatile("" :: "v"(a[0].v), "v"(b[0].v), "v"(b[1].v), "v"(b[2].v));
        if constexpr (NA == 2) asm volatile("" :: "v"(a[NA - 1].v));
        if (nk < 0)
#endif
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int s = 0; s < NA; ++s) {
                mma_chunk(acc[s][0], a[s], b[0], st);
                mma_chunk(acc[s][1], a[s], b[1], st);
                mma_chunk(acc[s][NSEL], a[s], b[2], st);
            }
        cur = cur == NB - 1 ? 0 : cur + 1;
    }
}

// ---- stack heads, distributed: every n-tile workgroup multiplies ITS 32 hidden units of the top layer's new hidden
// tile with its slice of the transposed head weights, P[row][o] = sum_c h[row][c] wcs_t[32 j + c][o], and hands the
// partial to the owner of the row, which adds the ntiles partials in tile order (fixed order).  The K = H contraction
// of the heads is thereby split over the workgroups that already hold the operands; nobody streams all of W_c / W_s.
//   hpart [(g ntiles + j) 64 + 32 slot + row][ldcs]      (rewritten every step: write-through stores, sc1 loads behind
//                                                          the acquire on the reading side)
// (noinline, called while no accumulator is live; small by-value context: the 2.5 KB argument block stays in the kernarg segment)
struct HeadsCtx { float* dst; int ldcs, n0, n1; };
__device__ __noinline__ void heads_partial(HeadsCtx c, f32x4 ho0, f32x4 ho1)
{
    const int tid = threadIdx.x, er = tid >> 3, c4 = (tid & 7) * 4, og = tid & 7, ldcs = c.ldcs;
    float* hs = g_lds;                                // [2][32][36]
    constexpr int HS_LD = 36, HS_TILE = 32 * HS_LD;
    __syncthreads();                                  // the partial sums of the last m-tile have been read
    st4(hs + er * HS_LD + c4, er < c.n0 ? ho0 : f32x4{0.f, 0.f, 0.f, 0.f});
    st4(hs + HS_TILE + er * HS_LD + c4, er < c.n1 ? ho1 : f32x4{0.f, 0.f, 0.f, 0.f});
    __syncthreads();
    // one 32 x 32 output tile per wave (m-tile s x 32 head columns), K = 32 as 16 MFMAs: the fragments are 4 LDS reads of
    // 16 bytes (h) and 16 of 4 bytes (weights), against 72 16-byte reads per thread for the same product on the VALU
    float* out = g_lds + 4096;                        // [tiles][32][EPI_LD]
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, hh = lane >> 5;
    const int nhn = ldcs >> 5;                        // head-column halves (ldcs <= 64)
    if (wave < 2 * nhn) {
        const int s = wave / nhn, nh = wave - s * nhn;
        if ((s ? c.n1 : c.n0) > 0) {                  // wave-uniform
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const f32x4 hv = ld4(hs + s * HS_TILE + li * HS_LD + 8 * cb + 4 * hh);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hv[k], g_wl[(8 * cb + 4 * hh + k) * ldcs + 32 * nh + li], acc, 0, 0, 0);
            }
            tile_to_lds(out + wave * EPI_TILE, acc, li, hh);
        }
    }
    __syncthreads();
    for (int o8 = og * 8; o8 < ldcs; o8 += 64) {
        for (int s = 0; s < 2; ++s) {
            const int n = s ? c.n1 : c.n0;
            if (er < n) {
                const float* t = out + (s * nhn + (o8 >> 5)) * EPI_TILE + er * EPI_LD + (o8 & 31);
                const long long off = ((long long)(32 * s + er) * ldcs + o8) * 4;
                st4_sc1(c.dst, off, ld4(t));
                st4_sc1(c.dst, off + 16, ld4(t + 4));
            }
        }
    }
}

__device__ __forceinline__ HeadsCtx heads_ctx(const SeqArgs& a, int g, int j, int n0, int n1)
{
    HeadsCtx c;
    c.dst = a.hpart + (long long)(g * a.ntiles + j) * 64 * a.ldcs;
    c.ldcs = a.ldcs; c.n0 = n0; c.n1 = n1;
    return c;
}

struct Step { int n[2]; long long base[2], pbase[2]; };   // active rows, packed row of the slot's first row now / one step earlier

// ---- one GRU layer of one step ------------------------------------------------------------------
template <int NA, bool BF>
__device__ __forceinline__ bool cell(const SeqArgs& a, float* lds, volatile int* flag, const int* r0, const Step& st, int t, int l,
                                     int g, int j, int wave, int lane, long long* tacc, long long& tlast)
{
    const int H = a.H;
    const int i = 8 * wave + (lane >> 3);                         // row of this lane's DMA piece inside a 32-row block
    const long long H3 = 3LL * H;
    // token of this thread's epilogue row (layer 0 looks its input-side pre-activations up in the [V, 3H] table):
    // fetched now, long before the table address is needed
    int tokv[2] = {0, 0};
    if (l == 0) {
#pragma unroll
        for (int s = 0; s < NA; ++s)
            if ((int)(threadIdx.x >> 3) < st.n[s]) {
                long long tok = a.tokens[st.base[s] + (threadIdx.x >> 3)];
                tokv[s] = (int)(tok < 0 ? 0 : (tok < a.V ? tok : a.V - 1));
            }
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][q][r] = 0.f;

    // part A: W_hh[l] . h_l(t-1)
    {
        const float* hp = t > 0 ? a.Hs + (long long)l * a.NT * H : a.h0 + (long long)l * a.B * H;
        const float* sa[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ns = s < NA ? st.n[s] : 1;
            const int ic = i < ns ? i : ns - 1;
            const long long row = (t > 0 ? st.pbase[s] : (long long)r0[s]) + ic;
            sa[s] = dma_src(hp, s < NA ? row : 0, H, i, lane);
        }
        const float* w = a.w_hh[l];
        const int u = 32 * j + i;
        kloop<NA, 3, BF>(lds, sa[0], sa[1], dma_src(w, u, H, i, lane), dma_src(w, (long long)H + u, H, i, lane),
                     dma_src(w, 2LL * H + u, H, i, lane), H / BK, acc, wave, lane);
    }
    TSTAMP(0);
    // the input of this layer at this step: the stack top (layer 0) or the layer below
    int* ctr = a.ctr + (long long)g * (a.L + 1) * CTR_STRIDE;
    if (!group_wait(l == 0 ? ctr + a.L * CTR_STRIDE : ctr + (l - 1) * CTR_STRIDE, a.ntiles * (t + 1), a.status, flag)) return false;
    TSTAMP(1);
    // part B: W_ih[l] . x_l(t)
    {
        const int K = l == 0 ? a.ks : H;
        const float* xp = l == 0 ? a.XS : (a.drop ? a.HD : a.Hs) + (long long)(l - 1) * a.NT * H;
        const float* sa[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ns = s < NA ? st.n[s] : 1;
            const int ic = i < ns ? i : ns - 1;
            sa[s] = dma_src(xp, s < NA ? st.base[s] + ic : 0, K, i, lane);
        }
        const float* w = l == 0 ? a.wx0s : a.w_ih[l];
        const int u = 32 * j + i;
        kloop<NA, 2, BF>(lds, sa[0], sa[1], dma_src(w, u, K, i, lane), dma_src(w, (long long)H + u, K, i, lane),
                     dma_src(w, 2LL * H + u, K, i, lane), K / BK, acc, wave, lane);
    }
    TSTAMP(2);
    // epilogue: sum the four K-chunk partials in wave order, GRU cell, stores
    const int li = lane & 31, hh = lane >> 5;
    const int tid = threadIdx.x, er = tid >> 3, c4 = (tid & 7) * 4, u = 32 * j + c4;
    f32x4 hout[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < NA; ++s) {
        // epilogue operands of this m-tile: issued before the transposition so that their latency hides behind it
        // (input-side bias: folded into the per-token table for layer 0, table = emb . W_ih0[:, :E]^T + b_ih0)
        f32x4 hp = {0.f, 0.f, 0.f, 0.f}, pf[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) pf[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (er < st.n[s]) {
            hp = ld4((t > 0 ? a.Hs + ((long long)l * a.NT + st.pbase[s] + er) * H : a.h0 + ((long long)l * a.B + r0[s] + er) * H) + u);
            const float* tp = l == 0 ? a.table + (long long)tokv[s] * H3 + u : a.b_ih[l] + u;
            const float* bp = a.b_hh[l] + u;
            pf[0] = ld4(tp); pf[1] = ld4(tp + H); pf[2] = ld4(tp + 2 * H);
            pf[3] = ld4(bp); pf[4] = ld4(bp + H); pf[5] = ld4(bp + 2 * H);
        }
        __syncthreads();                              // staging reads / the previous m-tile's sums are done
#pragma unroll
        for (int q = 0; q < 4; ++q) tile_to_lds(lds + (wave * 4 + q) * EPI_TILE, acc[s][q], li, hh);
        __syncthreads();
        if (er < st.n[s]) {
            f32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* p = lds + q * EPI_TILE + er * EPI_LD + c4;
                v[q] = ld4(p);
#pragma unroll
                for (int w = 1; w < 4; ++w) v[q] += ld4(p + w * 4 * EPI_TILE);
            }
            const long long row = st.base[s] + er;
            v[0] += pf[0] + pf[3];
            v[1] += pf[1] + pf[4];
            v[2] += pf[2];
            v[3] += pf[5];
            f32x4 rr, zz, nn, ho;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                rr[k] = sigmoidf_(v[0][k]);
                zz[k] = sigmoidf_(v[1][k]);
                nn[k] = tanhf(v[2][k] + rr[k] * v[3][k]);
                ho[k] = (1.0f - zz[k]) * nn[k] + zz[k] * hp[k];
            }
            const f32x4 hn = v[3];
            st4_sc1(a.Hs + ((long long)l * a.NT + st.base[s]) * H, ((long long)er * H + u) * 4, ho);      // handed to the group
            if (a.drop && l < a.L - 1)                   // launch-uniform branch: the layer above reads the dropped-out copy
                st4_sc1(a.HD + ((long long)l * a.NT + st.base[s]) * H, ((long long)er * H + u) * 4,
                        ho * ld4(a.drop + ((long long)l * a.NT + row) * H + u));
            st4(a.HP + ((long long)l * a.NT + row) * H + u, hp);
            float* gt = a.gates + ((long long)l * a.NT + row) * 4 * H + u;
            st4(gt, rr); st4(gt + H, zz); st4(gt + 2 * H, nn); st4(gt + 3 * H, hn);
            hout[s] = ho;
        }
    }
    TSTAMP(13);
    if (l == a.L - 1) heads_partial(heads_ctx(a, g, j, st.n[0], NA == 2 ? st.n[1] : 0), hout[0], hout[1]);
    TSTAMP(3);
    group_signal(ctr + l * CTR_STRIDE);
    TSTAMP(4);
    return true;
}

// ---- stack update of the rows this workgroup owns, for the stack ENTERING step t ----------------
// (heads = sum of the n-tile partials + bias -> softmax(3) / tanh(W); push / pop / no-op blend of the previous
// stack row, staged in LDS with 16-byte loads issued together; stack top -> XS, handed to the group)
struct StackCtx {
    const float* hpart_g;        // partials of this group: [ntiles][64][ldcs]
    const float* bcs; const float* Sprev; float* Snew; float* CS; float* XS;     // Sprev / Snew / CS / XS: at this step's first packed row
    int W, DW, ldcs, ks, rb, ntiles, bt, j, r0a, r0b;
    int* wctr; int* status; int wtarget;      // the group wait for the top layer's heads happens INSIDE: the previous stack rows
                                              // (this workgroup's own stores of the step before) are requested first
};
__device__ __noinline__ bool stack_phase(StackCtx c)
{
    float* lds = g_lds;
    const int W = c.W, DW = c.DW, DW4 = (DW + 3) & ~3, ldcs = c.ldcs, ks = c.ks, tid = threadIdx.x, RB = c.rb;
    const int nout = 3 + W;
    const bool vec = (DW & 3) == 0;
    float* spl = lds;                                 // [RB][DW4]
    float* act = spl + RB * DW4;                      // [RB][ldcs]
    float* top = act + RB * ldcs;                     // [RB][ks]
    // owned group rows: rho = j + k ntiles < 64, in batches of RB; srow = row of the sorted batch
    auto rowinfo = [&](int rho, long long& srow) -> bool {
        if (rho >= 64) return false;
        const int r0 = rho >= 32 ? c.r0b : c.r0a;
        if (r0 < 0) return false;
        srow = (long long)r0 + (rho & 31);
        return srow < c.bt;
    };
    bool waited = false;
    for (int rho0 = c.j; rho0 < 64; rho0 += RB * c.ntiles) {
        bool any = false;
        for (int k = 0; k < RB; ++k) { long long sr; any = any || rowinfo(rho0 + k * c.ntiles, sr); }
        if (!any) continue;                           // workgroup-uniform
        __syncthreads();
        // previous stack rows: the loads of all rows are issued FIRST (up to 16 x 16 bytes in flight per thread), the
        // partial-head loads right behind them, so both sets share one memory round trip
        const float* sp[4];
        bool on[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            long long sr = 0;
            on[k] = k < RB && rowinfo(rho0 + k * c.ntiles, sr);
            sp[k] = c.Sprev ? c.Sprev + sr * DW : nullptr;
        }
        const int n4 = DW >> 2;
        const bool vload = c.Sprev && vec;
        f32x4 tmp[4][4];
        if (vload) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int e = x * 256 + tid;
                    if (on[k] && e < n4) tmp[k][x] = ld4(sp[k] + 4 * e);
                }
        }
        if (!waited) { if (!group_wait(c.wctr, c.wtarget, c.status, &g_flag)) return false; waited = true; }
        TSTAMP(5);
        // heads = partials of the ntiles workgroups added in tile order + bias; sixteen independent 16-byte loads per thread
        for (int e = tid; e < RB * (ldcs >> 2); e += 256) {
            const int k = e / (ldcs >> 2), o = (e - k * (ldcs >> 2)) * 4, rho = rho0 + k * c.ntiles;
            long long sr;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (rowinfo(rho, sr)) {
                const long long b0 = ((long long)rho * ldcs + o) * 4;
                for (int j0 = 0; j0 < c.ntiles; j0 += 16) {
                    f32x4 p[16];
#pragma unroll
                    for (int x = 0; x < 16; ++x) {
                        const int jj = j0 + x < c.ntiles ? j0 + x : c.ntiles - 1;
                        p[x] = ld4_sc1(c.hpart_g, b0 + (long long)jj * 64 * ldcs * 4);
                    }
#pragma unroll
                    for (int x = 0; x < 16; ++x) if (j0 + x < c.ntiles) v += p[x];
                }
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] = o + x < nout ? v[x] + c.bcs[o + x] : 0.f;
            }
            st4(act + k * ldcs + o, v);
        }
        if (!c.Sprev) {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (on[k]) for (int e = tid; e < DW; e += 256) spl[k * DW4 + e] = 0.f;
        } else if (vec) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int e = x * 256 + tid;
                    if (on[k] && e < n4) st4(spl + k * DW4 + 4 * e, tmp[k][x]);
                }
            for (int e0 = 4 * 256; e0 < n4; e0 += 4 * 256) {           // rows beyond 4096 floats
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const int e = e0 + x * 256 + tid;
                        if (on[k] && e < n4) tmp[k][x] = ld4(sp[k] + 4 * e);
                    }
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const int e = e0 + x * 256 + tid;
                        if (on[k] && e < n4) st4(spl + k * DW4 + 4 * e, tmp[k][x]);
                    }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (on[k]) for (int e = tid; e < DW; e += 256) spl[k * DW4 + e] = sp[k][e];
        }
        TSTAMP(8);
        __syncthreads();
        TSTAMP(9);
        if (tid < RB) {
            float* r = act + tid * ldcs;
            const float p0 = r[0], p1 = r[1], p2 = r[2];
            const float m = fmaxf(p0, fmaxf(p1, p2));
            const float e0 = expf(p0 - m), e1 = expf(p1 - m), e2 = expf(p2 - m);
            const float inv = 1.0f / (e0 + e1 + e2);
            r[0] = e0 * inv; r[1] = e1 * inv; r[2] = e2 * inv;
        }
        for (int e = tid; e < RB * ldcs; e += 256) {
            const int o = e % ldcs;
            if (o >= 3 && o < nout) act[e] = tanhf(act[e]);
        }
        __syncthreads();
        TSTAMP(10);
        for (int k = 0; k < RB; ++k) {
            long long sr;
            if (!rowinfo(rho0 + k * c.ntiles, sr)) continue;
            const float* ar = act + k * ldcs;
            const float* sp = spl + k * DW4;
            float* cs = c.CS + sr * ldcs;
            for (int o = tid; o < nout; o += 256) cs[o] = ar[o];
            const float a_push = ar[0], a_pop = ar[1], a_noop = ar[2];
            float* sn = c.Snew + sr * DW;
            auto upd = [&](int e) {
                const float up = e < W ? ar[3 + e] : sp[e - W];
                const float down = e + W < DW ? sp[e + W] : 0.f;
                return a_noop * sp[e] + a_push * up + a_pop * down;
            };
            if (vec) {
                for (int e4 = tid; e4 < (DW >> 2); e4 += 256) {
                    f32x4 v;
#pragma unroll
                    for (int x = 0; x < 4; ++x) v[x] = upd(4 * e4 + x);
                    st4(sn + 4 * e4, v);
#pragma unroll
                    for (int x = 0; x < 4; ++x) if (4 * e4 + x < W) top[k * ks + 4 * e4 + x] = v[x];
                }
            } else {
                for (int e = tid; e < DW; e += 256) {
                    const float v = upd(e);
                    sn[e] = v;
                    if (e < W) top[k * ks + e] = v;
                }
            }
            for (int o = W + tid; o < ks; o += 256) top[k * ks + o] = 0.f;
        }
        TSTAMP(11);
        __syncthreads();
        TSTAMP(12);
        for (int e = tid; e < RB * (ks / 4); e += 256) {
            const int k = e / (ks / 4), o = (e - k * (ks / 4)) * 4;
            long long sr;
            if (rowinfo(rho0 + k * c.ntiles, sr)) st4_sc1(c.XS, (sr * ks + o) * 4, ld4(top + k * ks + o));      // handed to the group
        }
    }
    if (!waited && !group_wait(c.wctr, c.wtarget, c.status, &g_flag)) return false;      // no owned row active: same step order
    return true;
}

__device__ __forceinline__ StackCtx stack_ctx(const SeqArgs& a, const int* r0, const int* off, int t, int g, int j)
{
    StackCtx c;
    c.wctr = a.ctr + ((long long)g * (a.L + 1) + (a.L - 1)) * CTR_STRIDE; c.status = a.status; c.wtarget = a.ntiles * (t + 1);      // + the "step -1" signal
    const long long DW = (long long)a.D * a.W, ot = off[t], op = t > 0 ? off[t - 1] : 0;
    c.hpart_g = a.hpart + (long long)g * a.ntiles * 64 * a.ldcs;
    c.bcs = a.bcs;
    c.Sprev = t > 0 ? a.S + op * DW : a.S0;
    c.Snew = a.S + ot * DW;
    c.CS = a.CS + ot * a.ldcs;
    c.XS = a.XS + ot * a.ks;
    c.W = a.W; c.DW = (int)DW; c.ldcs = a.ldcs; c.ks = a.ks; c.rb = a.rb; c.ntiles = a.ntiles; c.bt = a.bs[t]; c.j = j;
    c.r0a = r0[0]; c.r0b = r0[1];
    return c;
}
template <bool BF>
__global__ __launch_bounds__(256, 2) void k_seq_fwd(const SeqArgs a)
{
    float* lds = g_lds;
    int* off = g_off;
    long long* tacc = nullptr;
    long long tlast = 0;
#ifdef PPK_SEQ_TIMING
    if (threadIdx.x == 0) { for (int k = 0; k < 16; ++k) g_tacc[k] = 0; g_tlast = __builtin_readcyclecounter(); }
#endif
    // Workgroup ids go round-robin over the 8 XCDs (observed placement; speed only, never correctness).
    // With >= 8 n-tiles an XCD serves the SAME n-tiles of every group: its weight slice (ntiles / 8 x 96 rows of every
    // matrix, 1.2 MB at H = 512) stays in its 4 MiB L2, instead of every group streaming all 9.8 MB of weights
    // through it each step; the hand-off reads (hidden rows, written through to memory anyway) cross XCDs.
    // With fewer n-tiles a group's workgroups share an XCD.
    // pair mode: the launch's groups sorted by length, first half of the grid = the longer half, second half = the
    // shorter half mirrored, so that a CU that receives one workgroup of each half (breadth-first dispatch) carries
    // the same number of row-block steps as every other CU
    const int half = a.hgrid > 0 && (int)blockIdx.x >= a.hgrid ? 1 : 0;
    const int bid = (int)blockIdx.x - half * a.hgrid;
    const int xcd = bid & 7, slot = bid >> 3;
    int g, j;
    if (a.tpx > 0) {
        // XCD x serves tile block x % xt of the groups with g % (8 / xt) == x / xt   (xt = 8: every XCD a tile block of all groups)
        const int gdiv = 8 / a.xt;
        g = (slot / a.tpx) * gdiv + xcd / a.xt;
        j = (xcd % a.xt) * a.tpx + slot % a.tpx;
        if (j >= a.ntiles) return;
    } else { g = (slot / a.ntiles) * 8 + xcd; j = slot % a.ntiles; }
    if (a.hgrid > 0) {
        const int nh = (a.g1 - a.g0 + 1) / 2;
        if (g >= nh) return;
        g = half ? a.g1 - 1 - g : a.g0 + g;
        if (half && g < a.g0 + nh) return;
    } else {
        g += a.g0;
        if (g >= a.g1) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) {
        int s = 0;
        for (int t = 0; t < a.T; ++t) { off[t] = s; s += a.bs[t]; }
        off[a.T] = s;
    }
    for (int e = tid * 4; e < 32 * a.ldcs; e += 1024) st4(g_wl + e, ld4(a.wcs_t + (long long)32 * j * a.ldcs + e));
    __syncthreads();
    const int m0 = a.pair ? 2 * g : g, m1 = a.pair ? 2 * g + 1 : 2 * a.G - 1 - g;
    const int r0[2] = {32 * m0, m1 < a.MT && m1 != m0 ? 32 * m1 : -1};
    int* ctr = a.ctr + (long long)g * (a.L + 1) * CTR_STRIDE;

    // "step -1": the heads of the initial hidden state of the top layer
    {
        const int er = tid >> 3, c4 = (tid & 7) * 4;
        int n0[2];
        f32x4 hv[2];
        for (int s = 0; s < 2; ++s) {
            int n = r0[s] >= 0 ? a.B - r0[s] : 0;
            n0[s] = n < 0 ? 0 : (n > 32 ? 32 : n);
            hv[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (er < n0[s]) hv[s] = ld4(a.h0 + ((long long)(a.L - 1) * a.B + r0[s] + er) * a.H + 32 * j + c4);
        }
        heads_partial(heads_ctx(a, g, j, n0[0], n0[1]), hv[0], hv[1]);
    }
    group_signal(ctr + (a.L - 1) * CTR_STRIDE);
    if (!stack_phase(stack_ctx(a, r0, off, 0, g, j))) return;      // a hand-off never arrived (status word set; k_seq_poison makes it visible in every output)
    group_signal(ctr + a.L * CTR_STRIDE);
    for (int t = 0; t < a.T; ++t) {
        const int bt = a.bs[t];
        if (bt <= r0[0]) break;
        Step st;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            int n = r0[s] >= 0 ? bt - r0[s] : 0;
            st.n[s] = n < 0 ? 0 : (n > 32 ? 32 : n);
            st.base[s] = (long long)off[t] + (r0[s] >= 0 ? r0[s] : 0);
            st.pbase[s] = (long long)(t > 0 ? off[t - 1] : 0) + (r0[s] >= 0 ? r0[s] : 0);
        }
        for (int l = 0; l < a.L; ++l) {
            const bool ok = st.n[1] > 0 ? cell<2, BF>(a, lds, &g_flag, r0, st, t, l, g, j, wave, lane, tacc, tlast)
                                        : cell<1, BF>(a, lds, &g_flag, r0, st, t, l, g, j, wave, lane, tacc, tlast);
            if (!ok) return;      // a hand-off never arrived (status word set; k_seq_poison makes it visible in every output)
        }
        if (t + 1 < a.T && a.bs[t + 1] > r0[0]) {
            if (!stack_phase(stack_ctx(a, r0, off, t + 1, g, j))) return;      // a hand-off never arrived (status word set; k_seq_poison makes it visible in every output)
            TSTAMP(6);
            group_signal(ctr + a.L * CTR_STRIDE);
            TSTAMP(7);
        }
    }
#ifdef PPK_SEQ_TIMING
    if (threadIdx.x == 0 && a.dbg) for (int k = 0; k < 16; ++k) a.dbg[(long long)(g * a.ntiles + j) * 16 + k] = g_tacc[k];
#endif
}

// [3+W, H] -> [H, ldcs] (zero padded): the heads read their weights k-major, coalesced over the outputs
__global__ __launch_bounds__(256) void k_wcs_transpose(const float* __restrict__ wcs, int nout, int H, int ldcs, float* __restrict__ out)
{
    const long long total = (long long)H * ldcs;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int k = (int)(e / ldcs), o = (int)(e - (long long)k * ldcs);
        out[e] = o < nout ? wcs[(long long)o * H + k] : 0.f;
    }
}

// A hand-off that never arrived (status word set by a timed-out wait): every buffer a caller consumes becomes NaN, so a
// lost hand-off cannot pass as numbers whatever the caller reads.  Launched after the sequence kernel on its stream;
// costs one status load per workgroup when nothing went wrong.
struct PoisonArgs { float* p[6]; long long n[6]; };
__global__ __launch_bounds__(256) void k_seq_poison(const int* __restrict__ status, PoisonArgs a)
{
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    const float nan = __builtin_nanf("");
    for (int k = 0; k < 6; ++k)
        for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < a.n[k]; e += (long long)gridDim.x * 256) a.p[k][e] = nan;
}

long long* g_dbg = nullptr;      // PPK_SEQ_TIMING builds only
int g_force = -2;                 // -1 auto, 0 never, 1 always (when supported)
int force_mode()
{
    if (g_force == -2) { const char* e = getenv("PPK_SEQ_PERSIST"); g_force = e ? atoi(e) : -1; }
    return g_force;
}

}  // namespace

namespace ppk {

int mfma_mode(int mode)
{
    static int g_mfma = -1;
    if (g_mfma < 0) { const char* e = getenv("PPK_MFMA"); g_mfma = e && atoi(e) == 1 ? 1 : 0; }
    if (mode == 0 || mode == 1) g_mfma = mode;
    return g_mfma;
}

int seq_persist_mode(int mode)
{
    if (mode >= -1 && mode <= 3) g_force = mode;
    return force_mode();
}

int seq_poison(const int* status, float* const* ptrs, const long long* counts, int n, hipStream_t stream)
{
    PoisonArgs a{};
    for (int k = 0; k < n && k < 6; ++k) { a.p[k] = ptrs[k]; a.n[k] = ptrs[k] ? counts[k] : 0; }
    k_seq_poison<<<dim3(512), dim3(256), 0, stream>>>(status, a);
    PPK_CHECK_LAUNCH("k_seq_poison");
    return PPK_OK;
}

// Workgroups of a sequence kernel that are resident together on the CURRENT device: every workgroup of a group must be,
// so a launch never carries more groups than fit.  Cached per device and per kernel variant (fp32 / bf16 operands).
int seq_capacity(const void* kernel, int slot)
{
    static int cache[64][4];
    int dev = 0, cus = 0, per = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cache[dev][slot] == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kernel, 256, 0) != hipSuccess) return 0;
        cache[dev][slot] = (per < 1 ? 0 : per) * (cus < 1 ? 0 : cus);
        if (cache[dev][slot] == 0) cache[dev][slot] = -1;
    }
    return cache[dev][slot] < 0 ? 0 : cache[dev][slot];
}

// auto mode: persistent or per-step launches (ppk_kernels.h has the measurements behind the rule)
bool seq_auto_persistent(int64_t B, int64_t H, int groups_per_launch, int capacity)
{
    const int64_t ntiles = H / 32, G = ((B + 31) / 32 + 1) / 2, W = G * ntiles;
    if (groups_per_launch < 1 || capacity < 1) return false;
    const int64_t fulls = (G + groups_per_launch - 1) / groups_per_launch;
    if (fulls == 1) return (ntiles >= 12 && 8 * W >= 3 * (int64_t)capacity) || (ntiles >= 8 && 10 * W >= 9 * (int64_t)capacity);
    if (fulls == 2) return ntiles >= 16 && 10 * W >= 18 * (int64_t)capacity;
    return false;
}

// groups per launch for `capacity` resident workgroups under the XCD placement (xt, tpx); 0 = a group does not fit
int seq_groups_per_launch(int capacity, int ntiles, int xt, int tpx)
{
    if (tpx > 0) return capacity / (xt * tpx) / (8 / xt) * (8 / xt);      // grid = 8 tpx workgroups per group class (some idle when ntiles % 8 != 0)
    int g = capacity / ntiles / 8 * 8;
    if (g < 8) g = capacity / ntiles;                                    // fewer than 8 groups at a time: still whole groups
    return g;
}

void seq_placement(int ntiles, int& xt, int& tpx)
{
    static int xt_env = -1;
    if (xt_env < 0) { const char* e = getenv("PPK_SEQ_XT"); xt_env = e ? atoi(e) : 4; }
    xt = ntiles >= 8 && (xt_env == 8 || xt_env == 4 || xt_env == 2) ? xt_env : 0;
    tpx = xt > 0 ? (ntiles + xt - 1) / xt : 0;
}

// share of a launch's algorithmic work: packed rows of the m-tiles of groups g0 .. g1-1 over all packed rows
double seq_row_share(const int* bs, int T, int G, int MT, int pair, int g0, int g1)
{
    double rows = 0.0, all = 0.0;
    for (int t = 0; t < T; ++t) {
        all += bs[t];
        for (int g = g0; g < g1; ++g) {
            const int m0 = pair ? 2 * g : g, m1 = pair ? 2 * g + 1 : 2 * G - 1 - g;
            for (int k = 0; k < 2; ++k) {
                const int m = k ? m1 : m0;
                if (k && (m >= MT || m == m0)) continue;
                const int n = bs[t] - 32 * m;
                rows += n < 0 ? 0 : (n > 32 ? 32 : n);
            }
        }
    }
    return all > 0 ? rows / all : 0.0;
}

bool seq_fwd_persist_ok(int64_t T, int64_t B, int64_t NT, int64_t H, int64_t L, int64_t W, int64_t D, int64_t ks, int64_t ldcs)
{
    if (force_mode() == 0) return false;
    if (T < 1 || T > MAXT || L < 1 || L > MAXL || H % 32 || H > 2048 || ks % 32 || ks > 512 || NT >= (1LL << 31) || B > (1 << 24)) return false;
    // the workspace (seq_fwd_persist_ws_bytes) is sized for the minimal paddings: a caller with wider ones takes the per-step path
    if (ldcs != (3 + W + 31) / 32 * 32 || ks != (W + 31) / 32 * 32) return false;
    const int64_t DW4 = (D * W + 3) / 4 * 4;
    if (DW4 + ldcs + ks > REGION) return false;                      // stack-phase scratch: at least one row at a time
    if (32 * ldcs > WL_MAX) return false;                            // the head-weight slice kept in LDS
    // every workgroup of a group must be resident together: a group that does not fit this device takes the per-step path
    int xt, tpx;
    seq_placement((int)(H / 32), xt, tpx);
    const int cap = ppk::mfma_mode() == 1 ? seq_capacity((const void*)k_seq_fwd<true>, 1) : seq_capacity((const void*)k_seq_fwd<false>, 0);
    const int gpl = seq_groups_per_launch(cap, (int)(H / 32), xt, tpx);
    if (gpl < 1) return false;
    if (force_mode() != 1 && !seq_auto_persistent(B, H, gpl, cap)) return false;      // auto (and mode 2 beyond the small-batch kernel's shapes): the per-layer-step launches are faster here
    return true;
}

size_t seq_fwd_persist_ws_bytes(int64_t B, int64_t H, int64_t L, int64_t W)
{
    const int64_t ldcs = (3 + W + 31) / 32 * 32;
    const int64_t G = ((B + 31) / 32 + 1) / 2;
    return ppk_align_up(sizeof(float) * (size_t)H * (size_t)ldcs, 256) + ppk_align_up(sizeof(int) * (size_t)(G * (L + 1) * CTR_STRIDE + 64), 256)
           + ppk_align_up(sizeof(float) * (size_t)G * (size_t)(H / 32) * 64 * (size_t)ldcs, 256);
}

int seq_fwd_persist(int64_t T, const int64_t* bs, int64_t B, int64_t NT, int64_t V, int64_t H, int64_t L, int64_t W, int64_t D,
                    const int64_t* tokens, const float* table, const float* wx0s, int64_t ks,
                    const float* const* w_ih, const float* const* w_hh, const float* const* b_ih, const float* const* b_hh,
                    const float* wcs, const float* bcs, const float* h0, const float* S0,
                    float* XS, float* Hs, float* HP, float* gates, float* S, float* CS, int64_t ldcs,
                    const float* drop, float* Hdrop, void* ws, size_t ws_bytes, hipStream_t stream)
{
    if (ws_bytes < seq_fwd_persist_ws_bytes(B, H, L, W)) PPK_FAIL(PPK_ERR_WS, "seq_fwd_persist: workspace too small");
    SeqArgs a{};
    a.T = (int)T; a.B = (int)B; a.H = (int)H; a.L = (int)L; a.W = (int)W; a.D = (int)D; a.V = (int)V; a.ks = (int)ks; a.ldcs = (int)ldcs;
    a.ntiles = (int)(H / 32);
    // XCD placement (speed only): xt tile blocks x 8 / xt group classes; xt = 8: tile-major, 0: group-major (PPK_SEQ_XT)
    seq_placement(a.ntiles, a.xt, a.tpx);
    { static int pair_env = -1; if (pair_env < 0) { const char* e = getenv("PPK_SEQ_PAIR"); pair_env = e ? atoi(e) : 0; } a.pair = pair_env; }   // 1: adjacent row blocks per group + mirrored grid halves (measured slower: 18.0 vs 17.3 ms forward; kept for experiments)
    a.MT = (int)((B + 31) / 32);
    a.G = (a.MT + 1) / 2;
    a.NT = NT;
    a.tokens = tokens; a.table = table; a.wx0s = wx0s;
    for (int l = 0; l < (int)L; ++l) { a.w_ih[l] = w_ih[l]; a.w_hh[l] = w_hh[l]; a.b_ih[l] = b_ih[l]; a.b_hh[l] = b_hh[l]; }
    a.bcs = bcs; a.h0 = h0; a.S0 = S0;
    a.dbg = g_dbg;
    a.XS = XS; a.Hs = Hs; a.HP = HP; a.gates = gates; a.S = S; a.CS = CS;
    a.drop = L > 1 ? drop : nullptr; a.HD = Hdrop;
    for (int t = 0; t < (int)T; ++t) a.bs[t] = (int)bs[t];
    char* cur = (char*)ws;
    float* wcs_t = (float*)ppk_carve(cur, sizeof(float) * (size_t)H * (size_t)ldcs);
    const size_t nctr = (size_t)a.G * (size_t)(L + 1) * CTR_STRIDE + 64;
    int* ctr = (int*)ppk_carve(cur, sizeof(int) * nctr);
    a.wcs_t = wcs_t; a.ctr = ctr; a.status = ctr + (size_t)a.G * (size_t)(L + 1) * CTR_STRIDE;
    a.hpart = (float*)ppk_carve(cur, sizeof(float) * (size_t)a.G * (size_t)a.ntiles * 64 * (size_t)ldcs);
    {
        const int64_t DW4 = (D * W + 3) / 4 * 4;
        int64_t rb = (REGION) / (DW4 + ldcs + ks);
        a.rb = (int)(rb > 4 ? 4 : rb);
    }
    PPK_HIP(hipMemsetAsync(ctr, 0, sizeof(int) * nctr, stream));
    k_wcs_transpose<<<dim3((unsigned)std::min<int64_t>((H * ldcs + 255) / 256, 1024)), dim3(256), 0, stream>>>(wcs, (int)(3 + W), (int)H, (int)ldcs, wcs_t);
    PPK_CHECK_LAUNCH("k_wcs_transpose");

    // every workgroup of a group must be resident together: launch at most one machine-full of groups at a time
    const bool bf = ppk::mfma_mode() == 1;
    const int capacity = bf ? seq_capacity((const void*)k_seq_fwd<true>, 1) : seq_capacity((const void*)k_seq_fwd<false>, 0);
    const int groups_per_launch = seq_groups_per_launch(capacity, a.ntiles, a.xt, a.tpx);
    if (groups_per_launch < 1) PPK_FAIL(PPK_ERR_UNSUP, "seq_fwd_persist: a group of %d workgroups does not fit the device", a.ntiles);
    const double flops = 2.0 * (double)NT * 3.0 * (double)H * ((double)ks + (2.0 * (double)L - 1.0) * (double)H);
    for (int g0 = 0; g0 < a.G; g0 += groups_per_launch) {
        const int ng = std::min(groups_per_launch, a.G - g0);
        a.g0 = g0;
        a.g1 = g0 + ng;
        auto grid_for = [&](int n) { return a.tpx > 0 ? 8 * a.tpx * ((n + 8 / a.xt - 1) / (8 / a.xt)) : ((n + 7) / 8) * 8 * a.ntiles; };
        a.hgrid = a.pair && ng > 1 ? grid_for((ng + 1) / 2) : 0;
        if (2 * a.hgrid > capacity) a.hgrid = 0;                  // rounding of the halves: never more workgroups than are resident together
        const unsigned grid = (unsigned)(a.hgrid > 0 ? 2 * a.hgrid : grid_for(ng));
        ProfScope prof(PK_SEQ_FWD, flops * seq_row_share(a.bs, a.T, a.G, a.MT, a.pair, a.g0, a.g1), stream);
        if (bf) k_seq_fwd<true><<<dim3(grid), dim3(256), 0, stream>>>(a);
        else k_seq_fwd<false><<<dim3(grid), dim3(256), 0, stream>>>(a);
        PPK_CHECK_LAUNCH("k_seq_fwd");
    }
    {
        float* ptrs[5] = {Hs, CS, XS, S, a.drop ? Hdrop : nullptr};
        const long long counts[5] = {(long long)L * NT * H, NT * ldcs, NT * ks, NT * D * W, (long long)(L - 1) * NT * H};
        return seq_poison(a.status, ptrs, counts, 5, stream);
    }
}

}  // namespace ppk

#ifdef PPK_SEQ_TIMING
// diagnosis builds only (make EXTRA=-DPPK_SEQ_TIMING; tools/seq_timing.py): [workgroups][8] int64 cycle sums per phase
extern "C" void ppk_dbg_seq_timing_buffer(long long* p) { g_dbg = p; }
#endif
