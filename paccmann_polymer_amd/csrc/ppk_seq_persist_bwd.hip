// K1 backward as ONE persistent launch: the hand-written backward of the packed StackGRU time loop
// (the autograd graph the reference builds through paccmann_chemistry StackGRU + nn.GRU(seq_len = 1) and walks at
// experiments/chem_reactions/reactions_vae.py:94-95, loss.backward()) without a kernel boundary per layer-step.
//
// Same decomposition as the forward (ppk_seq_persist.hip): rows of the length-sorted batch in m-tiles of 32, a GROUP =
// one long + one short m-tile, served by H/32 workgroups (one per 32 hidden units of the OUTPUT of every product),
// synchronised among themselves only.  Per step t (descending) and layer l (descending), for its 32 units u:
//   TOP      dh_{L-1} = carry_{L-1} + dHtop_t + DCS_{t+1} . [W_c ; W_s]                 (K = ldcs; the stack of step t+1
//                                                                                        read hidden[-1] of step t)
//   GATES_l  gate backward of layer l in the epilogue: dgi = [dr | dz | dn], dgh = [dr | dz | dn r], carry_l = dh z
//   P_l      accD = dgi_l . W_ih[l]  -> dh_{l-1} = accD + carry_{l-1} (next GATES)      (K = 3H, both from the SAME
//            accC = dgh_l . W_hh[l]  -> carry_l += accC                                   staged rows where dgi == dgh:
//                                                                                        the dr, dz two thirds of K)
//   layer 0: dXS = dgi_0 . W_ih0[:, E:E+W] as per-n-tile partials (K split over the workgroups that hold the operands),
//            added by the owner of the stack row, then the stack-update backward of the rows a workgroup owns
//            (two rows at a time in LDS, all threads on them, the three dot products per row as block reductions; the first
//            pair of rows is fetched by LDS-DMA before the wait for the partials), DCS_t handed to the group;
//            accC of layer 0 (carry_0 += dgh_0 . W_hh[0]) runs AFTER that hand-off: nothing waits for it but this
//            workgroup's own next step, so it fills the group's synchronisation latency.
// MFMA fp32 32x32x2 (bf16 operands under ppk_mfma_mode(1)), two products or two K tiles per stage, staged for the workgroup by
// LDS-DMA: 3 buffers, three stages in flight, a buffer refilled behind a second s_barrier once every wave holds its
// fragments; the four waves split each K tile (wave w takes chunk w) and accumulate all active m-tiles; partial tiles summed
// in wave order.
// Hand-offs (dgi / dgh rows, DCS rows, dXS partials) as in the forward: write-through stores + counter, poll + acquire.
#include "ppk_common.h"
#include "ppk_kernels.h"
#include "ppk_seq_common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

using namespace ppk_seq;

constexpr int MAXT = ppk::SEQP_MAXT, MAXL = ppk::SEQP_MAXL;
constexpr int STG_BUF = 6 * 1024;              // floats per staging buffer: A_x slot 0 / 1, A_y slot 0 / 1, B_x, B_y (32 x 32 each)
constexpr int NBUF = 3;
constexpr int REGION = NBUF * STG_BUF;         // >= 4 waves x 4 accumulators transposed; also the stack-phase scratch
static_assert(REGION >= 4 * 4 * EPI_TILE, "the transposed partials fit the shared region");

struct BwdArgs {
    int T, B, H, L, W, D, ks, ldcs, ntiles, G, MT, g0, g1, tpx, xt, rb, pair, hgrid;
    long long NT;
    const float* w_ih_t[MAXL]; const float* w_hh_t[MAXL];      // [H, 3H] each ([0]: unused, see wx)
    const float* wx;                                           // [3H, ks] = stack columns of W_ih0, k-major (zero padded)
    const float* wcs_t;                                        // [H, ldcs]
    const float* HP; const float* gates; const float* S; const float* CS; const float* dHtop;
    const float* drop;                                         // [L-1, NT, H] dropout factors of the forward (null: none)
    float* dXS; float* DGI; float* DGH; float* DCS; float* carry;     // carry = dh0 [L, B, H]
    float* dS;                                                 // [B, D W] stack gradient carried from step to step (rows by sorted index)
    float* xpart;                                              // [(g ntiles + j) 64 + row][ks] partial dXS
    int* ctr; int* status;
    long long* dbg;
    int bs[MAXT];
};

__shared__ __align__(1024) float g_lds[REGION];
__shared__ int g_off[MAXT + 1];
__shared__ int g_flag;
#ifdef PPK_SEQ_TIMING
__shared__ long long g_tacc[16];
__shared__ long long g_tlast;
#define TSTAMP(id) do { if (threadIdx.x == 0) { const long long now_ = __builtin_readcyclecounter(); g_tacc[id] += now_ - g_tlast; g_tlast = now_; } } while (0)
#else
#define TSTAMP(id) do { } while (0)
#endif

// ---- K loop: NBUF staging buffers, NBUF - 1 stages in flight, one s_barrier per stage; the wave takes chunk `wave` of
// every staged K tile.  A stage carries TWO products so that a wave issues 8 NA MFMAs per barrier:
//   accX[s] += Ax_s . Bx^T     accY[s] += Ay_s . By^T
// SHARE: Ay == Ax (staged once: the dr, dz columns, where dgi == dgh).  accX and accY may be the SAME accumulators
// (two consecutive K tiles of one product: the caller passes sources 32 columns apart and advances by 64).
template <int N> struct IntC { static constexpr int v = N; };
// waits until at most `stages` later stages (PPW loads each) are still in flight; stages in 0 .. 5
template <int PPW> __device__ __forceinline__ void wait_stages(int stages)
{
    switch (stages) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<PPW>(); break;
        case 2: wait_vmcnt<2 * PPW>(); break;
        case 3: wait_vmcnt<3 * PPW>(); break;
        case 4: wait_vmcnt<4 * PPW>(); break;
        default: wait_vmcnt<5 * PPW>(); break;
    }
}
template <int NA, bool SHARE, bool BF>
__device__ __forceinline__ void kloop2(float* lds, const float* x0, const float* x1, const float* y0, const float* y1,
                                       const float* bx, const float* by, int adv, int nst,
                                       f32x16 (&accX)[2], f32x16 (&accY)[2], int wave, int lane)
{
    constexpr int PPW = 2 + (SHARE ? NA : 2 * NA);
    // PPK_SEQ_BWD_DEEP: a buffer holds only the blocks this instantiation stages, so the shared region takes 6 / 4 / 3
    // buffers at 3 / 4 / 6 blocks per stage (twice the bytes in flight while ONE m-tile is active).  Measured equal to
    // the fixed three-buffer layout in a same-box ABBA run (k_seq_bwd 19.0 vs 18.8 ms per step): the K loops are not
    // bound by the latency of their staging loads.  Kept as a switch, off.
#ifndef PPK_SEQ_BWD_DEEP
    constexpr int SB = STG_BUF, NB = NBUF, OY = 2048, OB = 4096;
#else
    constexpr int SB = PPW * 1024, NB = REGION / SB > 6 ? 6 : REGION / SB, OY = NA * 1024, OB = (SHARE ? NA : 2 * NA) * 1024;
#endif
    static_assert(NB >= 3 && NB * SB <= REGION && 5 * PPW <= 63, "staging fits the shared region and the vmcnt field");
    const int li = lane & 31, h = lane >> 5;
    const int xo = li * BK + 4 * ((2 * wave + h) ^ ((li >> 1) & 7));
    auto stage = [&](int buf) {
#ifdef PPK_EXP_NODMA
        return;
#endif
        float* base = lds + buf * SB + wave * 256;
        __builtin_amdgcn_global_load_lds(x0, PPK_LDS_PTR(base), 16, 0, DMA_AUX_HANDOFF); x0 += adv;
        if constexpr (NA == 2) { __builtin_amdgcn_global_load_lds(x1, PPK_LDS_PTR(base + 1024), 16, 0, DMA_AUX_HANDOFF); x1 += adv; }
        if constexpr (!SHARE) {
            __builtin_amdgcn_global_load_lds(y0, PPK_LDS_PTR(base + OY), 16, 0, DMA_AUX_HANDOFF); y0 += adv;
            if constexpr (NA == 2) { __builtin_amdgcn_global_load_lds(y1, PPK_LDS_PTR(base + OY + 1024), 16, 0, DMA_AUX_HANDOFF); y1 += adv; }
        }
        __builtin_amdgcn_global_load_lds(bx, PPK_LDS_PTR(base + OB), 16, 0, 0); bx += adv;
        __builtin_amdgcn_global_load_lds(by, PPK_LDS_PTR(base + OB + 1024), 16, 0, 0); by += adv;
    };
    __syncthreads();                                  // the shared region is free
    // NB stages in flight: a buffer is refilled as soon as every wave holds its fragments of it in registers
#pragma unroll
    for (int p = 0; p < NB; ++p) if (nst > p) stage(p);
    int cur = 0;
    for (int it = 0; it < nst; ++it) {
        const int r = nst - 1 - it;
        wait_stages<PPW>(r < NB - 1 ? r : NB - 1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const float* base = lds + cur * SB + xo;
        Frag<BF> ax[NA], ay[NA], fx, fy;
#pragma unroll
        for (int s = 0; s < NA; ++s) {
            ax[s].set(*(const f32x4*)(base + s * 1024));
            if constexpr (!SHARE) ay[s].set(*(const f32x4*)(base + OY + s * 1024)); else ay[s] = ax[s];
        }
        fx.set(*(const f32x4*)(base + OB));
        fy.set(*(const f32x4*)(base + OB + 1024));
        if (it + NB < nst) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();             // every wave has read this buffer
            asm volatile("" ::: "memory");
            stage(cur);
        }
#ifdef PPK_EXP_NOMMA
        asm volatile("" :: "v"(ax[0].v), "v"(ay[0].v), "v"(fx.v), "v"(fy.v));
        if constexpr (NA == 2) asm volatile("" :: "v"(ax[NA - 1].v), "v"(ay[NA - 1].v));
        if (nst < 0)
#endif
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int s = 0; s < NA; ++s) {
                mma_chunk(accX[s], ax[s], fx, st);
                mma_chunk(accY[s], ay[s], fy, st);
            }
        cur = cur == NB - 1 ? 0 : cur + 1;
    }
}

// one product, one K tile per stage (odd remainders and K = 32 products)
template <int NA, bool BF>
__device__ __forceinline__ void kloop1(float* lds, const float* x0, const float* x1, const float* bx, int nk, f32x16 (&acc)[2], int wave, int lane)
{
    constexpr int PPW = 1 + NA;
    const int li = lane & 31, h = lane >> 5;
    const int xo = li * BK + 4 * ((2 * wave + h) ^ ((li >> 1) & 7));
    auto stage = [&](int buf) {
#ifdef PPK_EXP_NODMA
        return;
#endif
        float* base = lds + buf * STG_BUF + wave * 256;
        __builtin_amdgcn_global_load_lds(x0, PPK_LDS_PTR(base), 16, 0, DMA_AUX_HANDOFF); x0 += BK;
        if constexpr (NA == 2) { __builtin_amdgcn_global_load_lds(x1, PPK_LDS_PTR(base + 1024), 16, 0, DMA_AUX_HANDOFF); x1 += BK; }
        __builtin_amdgcn_global_load_lds(bx, PPK_LDS_PTR(base + 4096), 16, 0, 0); bx += BK;
    };
    __syncthreads();
    if (nk > 0) stage(0);
    if (nk > 1) stage(1);
    if (nk > 2) stage(2);
    int cur = 0;
    for (int it = 0; it < nk; ++it) {
        const int r = nk - 1 - it;
        if (r >= 2) wait_vmcnt<2 * PPW>(); else if (r == 1) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const float* base = lds + cur * STG_BUF + xo;
        Frag<BF> ax[NA], fx;
#pragma unroll
        for (int s = 0; s < NA; ++s) ax[s].set(*(const f32x4*)(base + s * 1024));
        fx.set(*(const f32x4*)(base + 4096));
        if (it + NBUF < nk) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stage(cur);
        }
#ifdef PPK_EXP_NOMMA
        asm volatile("" :: "v"(ax[0].v), "v"(fx.v));
        if constexpr (NA == 2) asm volatile("" :: "v"(ax[NA - 1].v));
        if (nk < 0)
#endif
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int s = 0; s < NA; ++s) mma_chunk(acc[s], ax[s], fx, st);
        cur = cur == NBUF - 1 ? 0 : cur + 1;
    }
}

// acc[s] += A_s[.. x 32 nk] . B[.. x 32 nk]^T, two K tiles per stage (+ one single stage when nk is odd)
template <int NA, bool BF>
__device__ __forceinline__ void kloop_one_product(float* lds, const float* x0, const float* x1, const float* bx, int nk,
                                                  f32x16 (&acc)[2], int wave, int lane)
{
    const int n2 = nk >> 1;
    if (n2 > 0) kloop2<NA, false, BF>(lds, x0, x1, x0 + BK, x1 + BK, bx, bx + BK, 2 * BK, n2, acc, acc, wave, lane);
    if (nk & 1) kloop1<NA, BF>(lds, x0 + 2 * BK * n2, x1 + 2 * BK * n2, bx + 2 * BK * n2, 1, acc, wave, lane);
}

struct Step {
    int n[2];            // active rows of the two m-tiles at step t
    int nn[2];           // ... at step t + 1 (rows that fed a stack update from hidden[-1] of step t)
    long long base[2];   // packed row of the m-tile's first row at step t
    long long nbase[2];  // ... at step t + 1
};

__device__ __forceinline__ void zero_acc(f32x16 (&a)[2])
{
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[s][r] = 0.f;
}

// sum of the four K-chunk partial tiles (wave order) of accumulator slot q of the dump layout, this thread's 4 columns
__device__ __forceinline__ f32x4 dump_sum(const float* lds, int q, int er, int c4)
{
    const float* p = lds + q * EPI_TILE + er * EPI_LD + c4;
    f32x4 v = ld4(p);
#pragma unroll
    for (int w = 1; w < 4; ++w) v += ld4(p + w * 4 * EPI_TILE);
    return v;
}

// ---- gate backward of layer l for the rows of m-tile s: dh = dsum + carry (+ ext) ------------------------------
// writes dgi / dgh (handed to the group) and carry = dh z (private to this workgroup); returns dr, dz, dn (layer 0: dXS).
// The operands are fetched BEFORE the accumulators are transposed through LDS, so their latency hides behind that.
struct GatePf { f32x4 cr, ex, rr, zz, nn, hn, hp; };
__device__ __forceinline__ void gates_prefetch(const BwdArgs& a, int l, long long row0, long long srow0, int er, int u, bool top, GatePf& f)
{
    const int H = a.H;
    const long long row = row0 + er;
    f.cr = ld4(a.carry + ((long long)l * a.B + srow0 + er) * H + u);
    f.ex = top ? ld4(a.dHtop + row * H + u) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float* gt = a.gates + ((long long)l * a.NT + row) * 4 * H + u;
    f.rr = ld4(gt); f.zz = ld4(gt + H); f.nn = ld4(gt + 2 * H); f.hn = ld4(gt + 3 * H);
    f.hp = ld4(a.HP + ((long long)l * a.NT + row) * H + u);
}
__device__ __forceinline__ void gates_bwd(const BwdArgs& a, int l, long long row0, long long srow0, int er, int u, const GatePf& f,
                                          const f32x4& dsum, f32x4& dr, f32x4& dz, f32x4& dn)
{
    const int H = a.H;
    const f32x4 dh = dsum + f.cr + f.ex;
    f32x4 dnr, c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        dn[k] = dh[k] * (1.0f - f.zz[k]) * (1.0f - f.nn[k] * f.nn[k]);
        dz[k] = dh[k] * (f.hp[k] - f.nn[k]) * f.zz[k] * (1.0f - f.zz[k]);
        dr[k] = dn[k] * f.hn[k] * f.rr[k] * (1.0f - f.rr[k]);
        dnr[k] = dn[k] * f.rr[k];
        c[k] = dh[k] * f.zz[k];
    }
    const float* gi = a.DGI + ((long long)l * a.NT + row0) * 3 * H;
    const float* gh = a.DGH + ((long long)l * a.NT + row0) * 3 * H;
    const long long o = ((long long)er * 3 * H + u) * 4;
    st4_sc1(gi, o, dr); st4_sc1(gi, o + 4LL * H, dz); st4_sc1(gi, o + 8LL * H, dn);
    st4_sc1(gh, o, dr); st4_sc1(gh, o + 4LL * H, dz); st4_sc1(gh, o + 8LL * H, dnr);
    st4(a.carry + ((long long)l * a.B + srow0 + er) * H + u, c);
}

// ---- dXS partial of this workgroup: P[row][w] = sum_{q, c} dgi0[row][q H + 32 j + c] wx[q H + 32 j + c][w] -----------
// gs (LDS) [2 m-tiles][3 gates][32 rows][36] holds the workgroup's dgi_0 tile; the wx rows come from L2.
struct XpartCtx { const float* wx; float* dst; int H, ks, j, n0, n1; };
__device__ __forceinline__ void xs_partial(XpartCtx c)
{
    constexpr int GLD = 36, GT = 32 * GLD;
    const int tid = threadIdx.x, er = tid >> 3, og = tid & 7, ks = c.ks;
    float* gs = g_lds;                                // [2][3][32][36]   (written by the caller, barrier below)
    float* wl = g_lds + 6 * GT;                       // [96][ks]
    for (int e = tid * 4; e < 96 * ks; e += 1024) {
        const int k = e / ks, w = e - k * ks;         // k = 32 q + c
        st4(wl + e, ld4(c.wx + ((long long)(k >> 5) * c.H + 32 * c.j + (k & 31)) * ks + w));
    }
    __syncthreads();
    for (int o8 = og * 8; o8 < ks; o8 += 64) {
        for (int s = 0; s < 2; ++s) {
            const int n = s ? c.n1 : c.n0;
            if (n == 0) continue;                     // workgroup-uniform
            f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int cb = 0; cb < 8; ++cb) {
                    const f32x4 gv = ld4(gs + (s * 3 + q) * GT + er * GLD + 4 * cb);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float* wp = wl + (q * 32 + 4 * cb + k) * ks + o8;
                        const f32x4 w0 = ld4(wp), w1 = ld4(wp + 4);
#pragma unroll
                        for (int x = 0; x < 4; ++x) { p0[x] = fmaf(gv[k], w0[x], p0[x]); p1[x] = fmaf(gv[k], w1[x], p1[x]); }
                    }
                }
            }
            if (er < n) {
                const long long off = ((long long)(32 * s + er) * ks + o8) * 4;
                st4_sc1(c.dst, off, p0);
                st4_sc1(c.dst, off + 16, p1);
            }
        }
    }
}

// ---- stack-update backward of the rows this workgroup owns at step t (one wave per row) ------------------------
struct StackCtx {
    const float* xpart_g;        // partial dXS of this group: [ntiles][64][ks]
    const float* Sprev;          // S at step t - 1, first packed row (null at t == 0: the empty stack)
    const float* CS; float* DCS; float* dXS;      // at step t's first packed row
    float* dS;                   // [B, D W] by sorted row
    int W, DW, ldcs, ks, rb, ntiles, bt, j, r0a, r0b;
};
// The first pair's rows of the incoming stack gradient and of the previous stack do not depend on the other
// workgroups' dXS partials: they are requested by LDS-DMA BEFORE the group wait (wctr / wtarget), which therefore
// happens in here.
__device__ __forceinline__ bool stack_bwd(StackCtx c, int* wctr, int wtarget, int* status)
{
    const int W = c.W, DW = c.DW, DW4 = (DW + 3) & ~3, ldcs = c.ldcs, ks = c.ks, tid = threadIdx.x, RB = c.rb;
    const int lane = tid & 63, wave = tid >> 6;
    const int nout = 3 + W;
    const bool vec = (DW & 3) == 0;
    // LDS: two rows at a time of the incoming gradient and of the previous stack (all 256 threads work on them; the
    // three dot products per row are block reductions), per-row heads for up to four rows
    float* dst = g_lds;                               // [2][DW4]
    float* spl = dst + 2 * DW4;                       // [2][DW4]
    float* act = spl + 2 * DW4;                       // [4][ldcs]  softmax(3) | tanh(W) saved by the forward
    float* xr = act + 4 * ldcs;                       // [4][ks]    dXS rows
    float* dc = xr + 4 * ks;                          // [4][ldcs]  DCS rows being assembled
    float* red = dc + 4 * ldcs;                       // [4 waves][2 rows][4]
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
        for (int k = 0; k < RB; ++k) { long long s_; any = any || rowinfo(rho0 + k * c.ntiles, s_); }
        if (!any) continue;                           // workgroup-uniform
        __syncthreads();
        bool pre = false;                             // pair 0 already in dst / spl (without its dXS share)
        if (!waited) {
            if (vec && (DW >> 2) <= 4 * 256) {
                // 1 KB pieces (one wave instruction each), dealt to the four waves; lanes beyond the row stay off
                const int n4 = DW >> 2;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    long long srk = 0;
                    if (!(k < RB && rowinfo(rho0 + k * c.ntiles, srk))) continue;       // workgroup-uniform
                    for (int pi = wave; pi * 64 < n4; pi += 4) {
                        const int e = pi * 64 + lane;
                        if (e < n4) {
                            __builtin_amdgcn_global_load_lds(c.dS + srk * DW + 4 * e, PPK_LDS_PTR(dst + k * DW4 + pi * 256), 16, 0, 0);
                            if (c.Sprev) __builtin_amdgcn_global_load_lds(c.Sprev + srk * DW + 4 * e, PPK_LDS_PTR(spl + k * DW4 + pi * 256), 16, 0, 0);
                            else st4(spl + k * DW4 + 4 * e, f32x4{0.f, 0.f, 0.f, 0.f});
                        }
                    }
                }
                pre = true;
            }
            if (!group_wait(wctr, wtarget, status, &g_flag)) return false;
            waited = true;
            wait_vmcnt<0>();                          // this wave's pieces have landed
            __syncthreads();
        }
        // dXS rows = partials of the ntiles workgroups added in tile order (sixteen loads in flight); heads of the rows
        for (int e = tid; e < RB * (ks >> 2); e += 256) {
            const int k = e / (ks >> 2), o = (e - k * (ks >> 2)) * 4, rho = rho0 + k * c.ntiles;
            long long s_;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (rowinfo(rho, s_)) {
                const long long b0 = ((long long)rho * ks + o) * 4;
                for (int j0 = 0; j0 < c.ntiles; j0 += 16) {
                    f32x4 p[16];
#pragma unroll
                    for (int x = 0; x < 16; ++x) {
                        const int jj = j0 + x < c.ntiles ? j0 + x : c.ntiles - 1;
                        p[x] = ld4_sc1(c.xpart_g, b0 + (long long)jj * 64 * ks * 4);
                    }
#pragma unroll
                    for (int x = 0; x < 16; ++x) if (j0 + x < c.ntiles) v += p[x];
                }
                st4(c.dXS + s_ * ks + o, v);
            }
            st4(xr + k * ks + o, v);
        }
        for (int e = tid; e < RB * ldcs; e += 256) {
            const int k = e / ldcs, o = e - k * ldcs;
            long long s_;
            act[e] = o < nout && rowinfo(rho0 + k * c.ntiles, s_) ? c.CS[s_ * ldcs + o] : 0.f;
        }
        for (int kb = 0; kb < RB; kb += 2) {
            bool on[2];
            long long sr[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) { sr[k] = 0; on[k] = kb + k < RB && rowinfo(rho0 + (kb + k) * c.ntiles, sr[k]); }
            if (!on[0] && !on[1]) continue;           // workgroup-uniform
            __syncthreads();                          // xr / act written; the previous pair of rows is done with dst / spl
            // the two rows -> LDS, all loads issued together
            if (kb == 0 && pre) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (on[k]) for (int e = tid; e < W; e += 256) dst[k * DW4 + e] += xr[k * ks + e];      // the stack top fed layer 0: + dXS
            } else if (vec) {
                const int n4 = DW >> 2;
                for (int e0 = 0; e0 < n4; e0 += 4 * 256) {
                    f32x4 tg[2][4], ts[2][4];
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const int e = e0 + x * 256 + tid;
                            if (on[k] && e < n4) {
                                tg[k][x] = ld4(c.dS + sr[k] * DW + 4 * e);
                                ts[k][x] = c.Sprev ? ld4(c.Sprev + sr[k] * DW + 4 * e) : f32x4{0.f, 0.f, 0.f, 0.f};
                            }
                        }
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const int e = e0 + x * 256 + tid;
                            if (on[k] && e < n4) {
                                if (4 * e < W) {                                   // the stack top fed layer 0: + dXS
#pragma unroll
                                    for (int y = 0; y < 4; ++y) if (4 * e + y < W) tg[k][x][y] += xr[(kb + k) * ks + 4 * e + y];
                                }
                                st4(dst + k * DW4 + 4 * e, tg[k][x]);
                                st4(spl + k * DW4 + 4 * e, ts[k][x]);
                            }
                        }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (on[k]) for (int e = tid; e < DW; e += 256) {
                        dst[k * DW4 + e] = c.dS[sr[k] * DW + e] + (e < W ? xr[(kb + k) * ks + e] : 0.f);
                        spl[k * DW4 + e] = c.Sprev ? c.Sprev[sr[k] * DW + e] : 0.f;
                    }
            }
            __syncthreads();
            float sums[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!on[k]) continue;
                const float* g = dst + k * DW4;
                const float* sp = spl + k * DW4;
                const float* ar = act + (kb + k) * ldcs;
                const float a_push = ar[0], a_pop = ar[1], a_noop = ar[2];
                float* out = c.dS + sr[k] * DW;
                auto elem = [&](int i) {
                    const float gv = g[i];
                    const float up = i < W ? ar[3 + i] : sp[i - W];
                    const float down = i + W < DW ? sp[i + W] : 0.f;
                    sums[k][0] = fmaf(gv, sp[i], sums[k][0]);
                    sums[k][1] = fmaf(gv, up, sums[k][1]);
                    sums[k][2] = fmaf(gv, down, sums[k][2]);
                    float o = a_noop * gv;                                        // gradient wrt the previous stack
                    if (i + W < DW) o = fmaf(a_push, g[i + W], o);
                    if (i >= W) o = fmaf(a_pop, g[i - W], o);
                    return o;
                };
                if (vec) {
                    for (int e = tid; e < (DW >> 2); e += 256) {
                        f32x4 o;
#pragma unroll
                        for (int y = 0; y < 4; ++y) o[y] = elem(4 * e + y);
                        st4(out + 4 * e, o);
                    }
                } else {
                    for (int i = tid; i < DW; i += 256) out[i] = elem(i);
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float v = ppk_wave_sum(sums[k][q]);
                    if (lane == 0) red[(wave * 2 + k) * 4 + q] = v;
                }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!on[k]) continue;
                const float* g = dst + k * DW4;
                const float* ar = act + (kb + k) * ldcs;
                const float a_push = ar[0], a_pop = ar[1], a_noop = ar[2];
                float t3[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) t3[q] = (red[(0 * 2 + k) * 4 + q] + red[(1 * 2 + k) * 4 + q]) + (red[(2 * 2 + k) * 4 + q] + red[(3 * 2 + k) * 4 + q]);
                const float s_noop = t3[0], s_push = t3[1], s_pop = t3[2];
                const float dot = a_push * s_push + a_pop * s_pop + a_noop * s_noop;
                for (int o = tid; o < ldcs; o += 256) {
                    float v = 0.f;
                    if (o == 0) v = a_push * (s_push - dot);
                    else if (o == 1) v = a_pop * (s_pop - dot);
                    else if (o == 2) v = a_noop * (s_noop - dot);
                    else if (o < nout) { const float sv = ar[o]; v = a_push * g[o - 3] * (1.0f - sv * sv); }
                    dc[(kb + k) * ldcs + o] = v;
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < RB * (ldcs >> 2); e += 256) {
            const int k = e / (ldcs >> 2), o = (e - k * (ldcs >> 2)) * 4;
            long long s_;
            if (rowinfo(rho0 + k * c.ntiles, s_)) st4_sc1(c.DCS, (s_ * ldcs + o) * 4, ld4(dc + k * ldcs + o));     // handed to the group
        }
    }
    if (!waited && !group_wait(wctr, wtarget, status, &g_flag)) return false;      // no owned row active: same step order
    return true;
}

__device__ __forceinline__ StackCtx stack_ctx(const BwdArgs& a, const int* r0, const int* off, int t, int g, int j)
{
    StackCtx c;
    const long long DW = (long long)a.D * a.W, ot = off[t], op = t > 0 ? off[t - 1] : 0;
    c.xpart_g = a.xpart + (long long)g * a.ntiles * 64 * a.ks;
    c.Sprev = t > 0 ? a.S + op * DW : nullptr;
    c.CS = a.CS + ot * a.ldcs;
    c.DCS = a.DCS + ot * a.ldcs;
    c.dXS = a.dXS + ot * a.ks;
    c.dS = a.dS;
    c.W = a.W; c.DW = (int)DW; c.ldcs = a.ldcs; c.ks = a.ks; c.rb = a.rb; c.ntiles = a.ntiles; c.bt = a.bs[t]; c.j = j;
    c.r0a = r0[0]; c.r0b = r0[1];
    return c;
}

// source of this lane's LDS-DMA piece for the rows of an m-tile (rows >= n are clamped; their products are zeroed
// or never stored by the epilogues)
__device__ __forceinline__ const float* a_src(const float* base, long long row0, int n, long long ld, int i, int lane)
{
    const int ic = i < n ? i : (n > 0 ? n - 1 : 0);
    return dma_src(base, row0 + ic, ld, i, lane);
}

template <int NA, bool BF>
__device__ __forceinline__ bool step_bwd(const BwdArgs& a, float* lds, const int* r0, const int* off, const Step& st, int t, int k,
                                         int g, int j, int wave, int lane)
{
    const int H = a.H, L = a.L, H3 = 3 * H;
    const int i = 8 * wave + (lane >> 3);
    const int li = lane & 31, hh = lane >> 5;
    const int tid = threadIdx.x, er = tid >> 3, c4 = (tid & 7) * 4, u = 32 * j + c4;
    const int ub = 32 * j + i;                        // weight row of this lane's DMA piece
    int* ctr = a.ctr + (long long)g * (L + 1) * CTR_STRIDE;
    f32x16 accD[2], accC[2];
    f32x4 d0[2][3];                                   // dgi of layer 0 of this thread's cells (for the dXS partial)

    // ---- TOP: DCS_{t+1} . [W_c ; W_s] for the rows that were still active at t + 1
    zero_acc(accD);
    if (st.nn[0] > 0) {
        if (!group_wait(ctr + L * CTR_STRIDE, a.ntiles * k, a.status, &g_flag)) return false;         // stack backward of step t + 1
        const float* s0 = a_src(a.DCS, st.nbase[0], st.nn[0], a.ldcs, i, lane);
        const float* s1 = a_src(a.DCS, st.nbase[1], NA == 2 ? st.nn[1] : 0, a.ldcs, i, lane);
        const float* sx = dma_src(a.wcs_t, ub, a.ldcs, i, lane);
        if (NA == 2 && st.nn[1] > 0) kloop_one_product<2, BF>(lds, s0, s1, sx, a.ldcs / BK, accD, wave, lane);
        else kloop_one_product<1, BF>(lds, s0, s0, sx, a.ldcs / BK, accD, wave, lane);
    }
    TSTAMP(0);
    for (int l = L - 1; l >= 0; --l) {
        // ---- gate backward of layer l: dh = accD (+ carry, + dHtop for the top layer)
#pragma unroll
        for (int s = 0; s < NA; ++s) {
            GatePf pf;
            if (er < st.n[s]) gates_prefetch(a, l, st.base[s], r0[s], er, u, l == L - 1, pf);
            __syncthreads();
            tile_to_lds(lds + (wave * 4 + 0) * EPI_TILE, accD[s], li, hh);
            if (l < L - 1) tile_to_lds(lds + (wave * 4 + 1) * EPI_TILE, accC[s], li, hh);
            __syncthreads();
            if (er < st.n[s]) {
                f32x4 dsum = dump_sum(lds, 0, er, c4);
                if (l == L - 1) { if (er >= st.nn[s]) dsum = f32x4{0.f, 0.f, 0.f, 0.f}; }           // no stack read this row's hidden at t + 1
                else {
                    // carry_{l+1} += dgh_{l+1} . W_hh[l+1]  (this workgroup's own tile: plain read-modify-write)
                    float* cp = a.carry + ((long long)(l + 1) * a.B + r0[s] + er) * H + u;
                    st4(cp, ld4(cp) + dump_sum(lds, 1, er, c4));
                    // layer l + 1 read h_l * drop_l: its input gradient comes back through the same factors (launch-uniform branch)
                    if (a.drop) dsum *= ld4(a.drop + ((long long)l * a.NT + st.base[s] + er) * H + u);
                }
                f32x4 dr, dz, dn;
                gates_bwd(a, l, st.base[s], r0[s], er, u, pf, dsum, dr, dz, dn);
                if (l == 0) { d0[s][0] = dr; d0[s][1] = dz; d0[s][2] = dn; }
            }
        }
        TSTAMP(1);
        if (l == 0) break;
        group_signal(ctr + l * CTR_STRIDE);
        // ---- P_l: accD = dgi_l . W_ih[l] (-> dh_{l-1}), accC = dgh_l . W_hh[l] (-> carry_l)
        zero_acc(accD);
        zero_acc(accC);
        if (!group_wait(ctr + l * CTR_STRIDE, a.ntiles * (k + 1), a.status, &g_flag)) return false;
        TSTAMP(2);
        {
            const float* gi = a.DGI + (long long)l * a.NT * H3;
            const float* gh = a.DGH + (long long)l * a.NT * H3;
            const float* wi = a.w_ih_t[l];
            const float* wh = a.w_hh_t[l];
            const float* s0 = a_src(gi, st.base[0], st.n[0], H3, i, lane);
            const float* s1 = a_src(gi, st.base[1], NA == 2 ? st.n[1] : 0, H3, i, lane);
            const float* sx = dma_src(wi, ub, H3, i, lane);
            const float* sy = dma_src(wh, ub, H3, i, lane);
            kloop2<NA, true, BF>(lds, s0, s1, s0, s1, sx, sy, BK, 2 * H / BK, accD, accC, wave, lane);        // dr, dz: dgi == dgh, staged once
            const float* t0 = a_src(gh, st.base[0], st.n[0], H3, i, lane);
            const float* t1 = a_src(gh, st.base[1], NA == 2 ? st.n[1] : 0, H3, i, lane);
            kloop2<NA, false, BF>(lds, s0 + 2 * H, s1 + 2 * H, t0 + 2 * H, t1 + 2 * H, sx + 2 * H, sy + 2 * H, BK, H / BK,
                              accD, accC, wave, lane);                                                    // dn with W_ih, dn r with W_hh
        }
        TSTAMP(3);
    }
    // ---- layer 0 is done: dXS partial, hand-off, stack backward of the owned rows, hand-off
    {
        constexpr int GLD = 36, GT = 32 * GLD;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NA; ++s)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                st4(lds + (s * 3 + q) * GT + er * GLD + c4, er < st.n[s] ? d0[s][q] : f32x4{0.f, 0.f, 0.f, 0.f});
        XpartCtx xc;
        xc.wx = a.wx; xc.dst = a.xpart + (long long)(g * a.ntiles + j) * 64 * a.ks; xc.H = H; xc.ks = a.ks; xc.j = j;
        xc.n0 = st.n[0]; xc.n1 = NA == 2 ? st.n[1] : 0;
        xs_partial(xc);
    }
    TSTAMP(4);
    group_signal(ctr + 0);
    TSTAMP(5);
    if (!stack_bwd(stack_ctx(a, r0, off, t, g, j), ctr + 0, a.ntiles * (k + 1), a.status)) return false;     // waits for E_0 inside
    TSTAMP(6);
    group_signal(ctr + L * CTR_STRIDE);
    // ---- carry_0 += dgh_0 . W_hh[0]: only this workgroup's next step needs it — it fills the hand-off latency
    {
        zero_acc(accC);
        const float* gh = a.DGH;
        const float* s0 = a_src(gh, st.base[0], st.n[0], H3, i, lane);
        const float* s1 = a_src(gh, st.base[1], NA == 2 ? st.n[1] : 0, H3, i, lane);
        const float* sy = dma_src(a.w_hh_t[0], ub, H3, i, lane);
        kloop_one_product<NA, BF>(lds, s0, s1, sy, H3 / BK, accC, wave, lane);
#pragma unroll
        for (int s = 0; s < NA; ++s) {
            __syncthreads();
            tile_to_lds(lds + (wave * 4 + 0) * EPI_TILE, accC[s], li, hh);
            __syncthreads();
            if (er < st.n[s]) {
                float* cp = a.carry + ((long long)r0[s] + er) * H + u;
                st4(cp, ld4(cp) + dump_sum(lds, 0, er, c4));
            }
        }
    }
    TSTAMP(7);
    return true;
}

template <bool BF>
__global__ __launch_bounds__(256, 2) void k_seq_bwd(const BwdArgs a)
{
    float* lds = g_lds;
    int* off = g_off;
#ifdef PPK_SEQ_TIMING
    if (threadIdx.x == 0) { for (int k = 0; k < 16; ++k) g_tacc[k] = 0; g_tlast = __builtin_readcyclecounter(); }
#endif
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
    __syncthreads();
    const int m0 = a.pair ? 2 * g : g, m1 = a.pair ? 2 * g + 1 : 2 * a.G - 1 - g;
    const int r0[2] = {32 * m0, m1 < a.MT && m1 != m0 ? 32 * m1 : -1};
    int Tg = 0;
    while (Tg < a.T && a.bs[Tg] > r0[0]) ++Tg;        // steps of the group's longest row
    auto rows = [&](int t, int s) { int n = (t < a.T && r0[s] >= 0) ? a.bs[t] - r0[s] : 0; return n < 0 ? 0 : (n > 32 ? 32 : n); };
    for (int t = Tg - 1, k = 0; t >= 0; --t, ++k) {
        Step st;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            st.n[s] = rows(t, s);
            st.nn[s] = rows(t + 1, s);
            st.base[s] = (long long)off[t] + (r0[s] >= 0 ? r0[s] : 0);
            st.nbase[s] = (long long)(t + 1 < a.T ? off[t + 1] : 0) + (r0[s] >= 0 ? r0[s] : 0);
        }
        const bool ok = st.n[1] > 0 ? step_bwd<2, BF>(a, lds, r0, off, st, t, k, g, j, wave, lane)
                                    : step_bwd<1, BF>(a, lds, r0, off, st, t, k, g, j, wave, lane);
        if (!ok) return;      // a hand-off never arrived (status word set; k_seq_poison makes it visible in every output)
    }
    // the stack of step 0 read the INITIAL hidden state of the top layer: dh0[L-1] += DCS_0 . [W_c ; W_s]
    if (Tg > 0) {
        int* ctr = a.ctr + (long long)g * (a.L + 1) * CTR_STRIDE;
        if (!group_wait(ctr + a.L * CTR_STRIDE, a.ntiles * Tg, a.status, &g_flag)) return;      // a hand-off never arrived (status word set; k_seq_poison makes it visible in every output)
        const int i = 8 * wave + (lane >> 3), li = lane & 31, hh = lane >> 5, er = tid >> 3, c4 = (tid & 7) * 4, u = 32 * j + c4;
        f32x16 accD[2];
        zero_acc(accD);
        const int n0 = rows(0, 0), n1 = rows(0, 1);
        const float* s0 = a_src(a.DCS, r0[0], n0, a.ldcs, i, lane);
        const float* s1 = a_src(a.DCS, r0[1] >= 0 ? r0[1] : 0, n1, a.ldcs, i, lane);
        const float* sx = dma_src(a.wcs_t, 32 * j + i, a.ldcs, i, lane);
        if (n1 > 0) kloop_one_product<2, BF>(lds, s0, s1, sx, a.ldcs / BK, accD, wave, lane);
        else kloop_one_product<1, BF>(lds, s0, s0, sx, a.ldcs / BK, accD, wave, lane);
        for (int s = 0; s < 2; ++s) {
            const int n = s ? n1 : n0;
            if (n == 0) continue;
            __syncthreads();
            tile_to_lds(lds + (wave * 4 + 0) * EPI_TILE, accD[s], li, hh);
            __syncthreads();
            if (er < n) {
                float* cp = a.carry + ((long long)(a.L - 1) * a.B + r0[s] + er) * a.H + u;
                st4(cp, ld4(cp) + dump_sum(lds, 0, er, c4));
            }
        }
    }
#ifdef PPK_SEQ_TIMING
    if (threadIdx.x == 0 && a.dbg) for (int k = 0; k < 16; ++k) a.dbg[(long long)(g * a.ntiles + j) * 16 + k] = g_tacc[k];
#endif
}

// [W, 3H] -> [3H, ks] (zero padded): the stack columns of W_ih0 k-major, as the forward keeps them
__global__ __launch_bounds__(256) void k_wx_transpose(const float* __restrict__ wt, int W, int H3, int ks, float* __restrict__ out)
{
    const long long total = (long long)H3 * ks;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int k = (int)(e / ks), w = (int)(e - (long long)k * ks);
        out[e] = w < W ? wt[(long long)w * H3 + k] : 0.f;
    }
}

long long* g_dbg = nullptr;

}  // namespace

namespace ppk {

bool seq_bwd_persist_ok(int64_t T, int64_t B, int64_t NT, int64_t H, int64_t L, int64_t W, int64_t D, int64_t ks, int64_t ldcs)
{
    if (seq_persist_mode(-2) == 0) return false;
    if (T < 1 || T > MAXT || L < 1 || L > MAXL || H % 32 || H > 2048 || ks % 32 || ks > 64 || NT >= (1LL << 31) || B > (1 << 24)) return false;
    // the workspace (seq_bwd_persist_ws_bytes) is sized for the minimal paddings: a caller with wider ones takes the per-step path
    if (ldcs != (3 + W + 31) / 32 * 32 || ks != (W + 31) / 32 * 32 || ldcs > 512) return false;
    if (NT * 3 * H * 4 >= (1LL << 31)) return false;                // byte offsets of the buffer stores inside one layer's dgi
    const int64_t DW4 = (D * W + 3) / 4 * 4;
    if (4 * DW4 + 8 * ldcs + 4 * ks + 32 > REGION) return false;     // stack-phase scratch: two rows at a time + four rows' heads
    if (6 * 32 * 36 + 96 * ks > REGION) return false;                // dXS-partial scratch
    int xt, tpx;
    seq_placement((int)(H / 32), xt, tpx);
    const int cap = ppk::mfma_mode() == 1 ? seq_capacity((const void*)k_seq_bwd<true>, 3) : seq_capacity((const void*)k_seq_bwd<false>, 2);
    const int gpl = seq_groups_per_launch(cap, (int)(H / 32), xt, tpx);
    if (gpl < 1) return false;                                                      // a group does not fit this device: per-step path
    if (seq_persist_mode(-2) != 1 && !seq_auto_persistent(B, H, gpl, cap)) return false;
    return true;
}

size_t seq_bwd_persist_ws_bytes(int64_t B, int64_t H, int64_t L, int64_t W, int64_t D)
{
    const int64_t ks = (W + 31) / 32 * 32;
    const int64_t G = ((B + 31) / 32 + 1) / 2;
    return ppk_align_up(sizeof(float) * (size_t)3 * H * ks, 256) + ppk_align_up(sizeof(int) * (size_t)(G * (L + 1) * CTR_STRIDE + 64), 256)
           + ppk_align_up(sizeof(float) * (size_t)G * (size_t)(H / 32) * 64 * (size_t)ks, 256)
           + ppk_align_up(sizeof(float) * (size_t)B * (size_t)D * (size_t)W, 256);
}

int seq_bwd_persist(int64_t T, const int64_t* bs, int64_t B, int64_t NT, int64_t H, int64_t L, int64_t W, int64_t D,
                    const float* const* w_ih_t, const float* const* w_hh_t, const float* wcs_t,
                    const float* HP, const float* gates, const float* S, const float* CS, int64_t ldcs,
                    const float* dHtop, float* dXS, int64_t ks, float* DGI, float* DGH, float* DCS, float* dh0,
                    const float* drop, void* ws, size_t ws_bytes, hipStream_t stream)
{
    if (ws_bytes < seq_bwd_persist_ws_bytes(B, H, L, W, D)) PPK_FAIL(PPK_ERR_WS, "seq_bwd_persist: workspace too small");
    BwdArgs a{};
    a.T = (int)T; a.B = (int)B; a.H = (int)H; a.L = (int)L; a.W = (int)W; a.D = (int)D; a.ks = (int)ks; a.ldcs = (int)ldcs;
    a.ntiles = (int)(H / 32);
    // XCD placement (speed only): xt tile blocks x 8 / xt group classes; xt = 8: tile-major, 0: group-major (PPK_SEQ_XT)
    seq_placement(a.ntiles, a.xt, a.tpx);
    { static int pair_env = -1; if (pair_env < 0) { const char* e = getenv("PPK_SEQ_PAIR"); pair_env = e ? atoi(e) : 0; } a.pair = pair_env; }   // 1: adjacent row blocks per group + mirrored grid halves (measured slower: 18.0 vs 17.3 ms forward; kept for experiments)
    a.MT = (int)((B + 31) / 32);
    a.G = (a.MT + 1) / 2;
    a.NT = NT;
    for (int l = 0; l < (int)L; ++l) { a.w_ih_t[l] = w_ih_t[l]; a.w_hh_t[l] = w_hh_t[l]; }
    a.wcs_t = wcs_t;
    a.HP = HP; a.gates = gates; a.S = S; a.CS = CS; a.dHtop = dHtop;
    a.drop = L > 1 ? drop : nullptr;
    a.dXS = dXS; a.DGI = DGI; a.DGH = DGH; a.DCS = DCS; a.carry = dh0;
    a.dbg = g_dbg;
    for (int t = 0; t < (int)T; ++t) a.bs[t] = (int)bs[t];
    char* cur = (char*)ws;
    float* wx = (float*)ppk_carve(cur, sizeof(float) * (size_t)3 * H * ks);
    const size_t nctr = (size_t)a.G * (size_t)(L + 1) * CTR_STRIDE + 64;
    int* ctr = (int*)ppk_carve(cur, sizeof(int) * nctr);
    a.wx = wx; a.ctr = ctr; a.status = ctr + (size_t)a.G * (size_t)(L + 1) * CTR_STRIDE;
    a.xpart = (float*)ppk_carve(cur, sizeof(float) * (size_t)a.G * (size_t)a.ntiles * 64 * (size_t)ks);
    const size_t sbytes = sizeof(float) * (size_t)B * (size_t)D * (size_t)W;
    a.dS = (float*)ppk_carve(cur, sbytes);
    a.rb = 4;                                         // owned rows per pass of the stack phase (two at a time in LDS)
    PPK_HIP(hipMemsetAsync(ctr, 0, sizeof(int) * nctr, stream));
    PPK_HIP(hipMemsetAsync(a.dS, 0, sbytes, stream));
    PPK_HIP(hipMemsetAsync(dh0, 0, sizeof(float) * (size_t)L * B * H, stream));
    k_wx_transpose<<<dim3((unsigned)std::min<int64_t>((3 * H * ks + 255) / 256, 1024)), dim3(256), 0, stream>>>(w_ih_t[0], (int)W, (int)(3 * H), (int)ks, wx);
    PPK_CHECK_LAUNCH("k_wx_transpose");

    const bool bf = ppk::mfma_mode() == 1;
    const int capacity = bf ? seq_capacity((const void*)k_seq_bwd<true>, 3) : seq_capacity((const void*)k_seq_bwd<false>, 2);
    const int groups_per_launch = seq_groups_per_launch(capacity, a.ntiles, a.xt, a.tpx);
    if (groups_per_launch < 1) PPK_FAIL(PPK_ERR_UNSUP, "seq_bwd_persist: a group of %d workgroups does not fit the device", a.ntiles);
    const double flops = 2.0 * (double)NT * (double)H * (3.0 * (double)H * (2.0 * (double)L - 1.0) + (double)ldcs) + 2.0 * (double)NT * 3.0 * (double)H * (double)ks;
    for (int g0 = 0; g0 < a.G; g0 += groups_per_launch) {
        const int ng = std::min(groups_per_launch, a.G - g0);
        a.g0 = g0;
        a.g1 = g0 + ng;
        auto grid_for = [&](int n) { return a.tpx > 0 ? 8 * a.tpx * ((n + 8 / a.xt - 1) / (8 / a.xt)) : ((n + 7) / 8) * 8 * a.ntiles; };
        a.hgrid = a.pair && ng > 1 ? grid_for((ng + 1) / 2) : 0;
        if (2 * a.hgrid > capacity) a.hgrid = 0;                  // rounding of the halves: never more workgroups than are resident together
        const unsigned grid = (unsigned)(a.hgrid > 0 ? 2 * a.hgrid : grid_for(ng));
        ProfScope prof(PK_SEQ_BWD, flops * seq_row_share(a.bs, a.T, a.G, a.MT, a.pair, a.g0, a.g1), stream);
        if (bf) k_seq_bwd<true><<<dim3(grid), dim3(256), 0, stream>>>(a);
        else k_seq_bwd<false><<<dim3(grid), dim3(256), 0, stream>>>(a);
        PPK_CHECK_LAUNCH("k_seq_bwd");
    }
    {
        float* ptrs[5] = {dXS, DGI, DGH, DCS, dh0};
        const long long counts[5] = {NT * ks, (long long)L * NT * 3 * H, (long long)L * NT * 3 * H, NT * ldcs, (long long)L * B * H};
        return seq_poison(a.status, ptrs, counts, 5, stream);
    }
}

}  // namespace ppk

#ifdef PPK_SEQ_TIMING
extern "C" void ppk_dbg_seq_bwd_timing_buffer(long long* p) { g_dbg = p; }
#endif
