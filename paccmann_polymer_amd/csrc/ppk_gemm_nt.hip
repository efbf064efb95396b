// NT fp32-MFMA GEMMs with LDS-DMA staging and fused GRU epilogues (gfx950).
//
//   acc[M,N] = A[M,K] . B[N,K]^T        (both operands k-contiguous, K % 32 == 0)
//
// These are the per-timestep contractions of the StackGRU sequence — the work
// that reaches cuDNN/MIOpen/ATen `gru_cell` through `nn.GRU(seq_len=1)` in the
// reference (paccmann_encoder.py:166-168 -> paccmann_chemistry StackGRU).
//
// Staging: `global_load_lds_dwordx4` writes 8-row x 128-B pieces straight into
// LDS (no VGPR round trip, no ds_write); the LDS image is lane-linear, so the
// bank swizzle is applied to the per-lane SOURCE address and undone on the
// fragment read (16-B chunk p of row r holds logical chunk p ^ ((r >> 1) & 7),
// which makes every ds_read_b128 lane group hit 16 distinct bank quads).
// One raw s_barrier per K tile, counted vmcnt so the next tiles stay in flight.
//
// Four kernels (measured on MI355X: tools/gemm_lab3.hip, tools/bench_nt*.py):
//   k_nt64  64 x 64, 4 waves 2x2, 1 accumulator per wave, 4 LDS buffers    — generic, two co-resident workgroups per CU
//   k_ntF   128 x (3 x 32) "gate layout", 4 waves along M, 3 accumulators   — a lane holds r, z, n of one
//           (row, hidden unit), so the GRU cell is evaluated in registers
//   k_nt32 / k_ntFs   the same two shapes on 32-row tiles with K split over the waves (no barrier in the main
//           loop) for the short tail steps; the host picks the family per launch from a fitted cost model.
// A launch is a 1-D grid over the tiles of up to four problems, remapped so that every XCD gets a contiguous
// run of every problem's tiles (nt_tile).  Epilogue operands are prefetched into registers before the K loop;
// the bodies are instantiated per epilogue mode so that the outstanding-load counts stay compile-time constants.
// Every inner-loop variant tried lands at 70-81 % of the MFMA rate and every launch carries ~8 us that do not
// scale with K; the dead ends (deeper prefetch, 128 x 64 tiles, K-split 64 x 64, two accumulator chains, wave
// priorities) are listed in DESIGN.md.
#include "ppk_common.h"
#include "ppk_kernels.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
#define PPK_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// A launch is a 1-D grid over the tiles of up to four problems.  Workgroup ids are dealt round-robin over
// the 8 XCDs (id % 8), each with its own 4 MiB L2, so the grid index is remapped to give every XCD a
// CONTIGUOUS run of each problem's tiles in n-fastest order: the tiles that share an A row block (and walk
// K in step) sit on one XCD and fetch it from beyond L2 once, not once per XCD.  Every XCD gets the same
// share of EVERY problem, so a launch that pairs a short-K problem with a long-K one stays balanced.
struct NtBatch {
    ppk::NtProblem p[ppk::NT_MAX_GROUP];
    int tn[ppk::NT_MAX_GROUP];              // n-tiles of each problem
    int tiles[ppk::NT_MAX_GROUP];           // tiles of each problem
    int cp[ppk::NT_MAX_GROUP];              // tiles of each problem per XCD = ceil(tiles / 8)
    int coff[ppk::NT_MAX_GROUP + 1];        // prefix sums of cp: slots of one XCD
    int count;
};

// tile of this workgroup: problem z, tile coordinates (tm, tn); false = padding workgroup
__device__ __forceinline__ bool nt_tile(const NtBatch& nb, int& z, int& tm, int& tn)
{
    const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
    z = 0;
    while (z + 1 < nb.count && slot >= nb.coff[z + 1]) ++z;
    const int t = xcd * nb.cp[z] + (slot - nb.coff[z]);
    if (t >= nb.tiles[z]) return false;
    tn = t % nb.tn[z];
    tm = t / nb.tn[z];
    return true;
}

// ---------------------------------------------------------------------------
// epilogues.  A 32x32 accumulator has its column on the lane and 16 rows in registers
// (row(r) = (r & 3) + 8 (r >> 2) + 4 h): each wave first transposes its tile through its
// own LDS region (row stride 36 floats) so that a lane owns FOUR CONSECUTIVE COLUMNS of one
// row — every global access of the epilogue is then a 16-byte access and a wave covers
// 8 rows x 128 B per instruction.  Pass p handles rows 8 p + (lane >> 3), columns 4 (lane & 7).
// ---------------------------------------------------------------------------
constexpr int EPI_LD = 36;                 // floats per transposed row (16-B aligned, bank-friendly)
constexpr int EPI_TILE = 32 * EPI_LD;      // floats per 32x32 tile

__device__ __forceinline__ void tile_to_lds(float* t, const f32x16& acc, int li, int h)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) t[((r & 3) + 8 * (r >> 2) + 4 * h) * EPI_LD + li] = acc[r];
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }

// sum of NPART partial tiles (fixed order) at element offset o
template <int NPART>
__device__ __forceinline__ f32x4 ldsum(const float* t, int pstride, int o)
{
    f32x4 v = ld4(t + o);
#pragma unroll
    for (int k = 1; k < NPART; ++k) v += ld4(t + k * pstride + o);
    return v;
}

// Every epilogue handles passes p0 .. p0 + NP - 1 of a transposed 32 x 32 tile that is the sum of
// NPART partial tiles (K split over waves), `pstride` floats apart.

// C = acc + bias + beta C ; col0 = first of the tile's 32 columns, coff = column offset of the tile group in C / bias
template <int NP, int NPART>
__device__ __forceinline__ void epi_store(const ppk::NtProblem& g, const float* t, int pstride, int p0, int row0, int col0,
                                          long long coff, int ncols_total, int lane)
{
    const int c4 = (lane & 7) * 4, col = col0 + c4;
    const bool vec = !(g.ldc & 3) && !((uintptr_t)g.C & 15) && !(coff & 3) && col + 3 < ncols_total;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (g.bias)
#pragma unroll
        for (int k = 0; k < 4; ++k) if (col + k < ncols_total) bv[k] = g.bias[coff + col + k];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
        const int rl = 8 * (p0 + pp) + (lane >> 3), row = row0 + rl;
        if (row >= g.M) continue;
        f32x4 v = ldsum<NPART>(t, pstride, rl * EPI_LD + c4);
        if (row >= g.a_rows) v = f32x4{0.f, 0.f, 0.f, 0.f};
        float* cp = g.C + (long long)row * g.ldc + coff + col;
        if (vec) {
            f32x4 o = {v[0] + bv[0], v[1] + bv[1], v[2] + bv[2], v[3] + bv[3]};
            if (g.beta != 0.f) { const f32x4 c = ld4(cp); for (int k = 0; k < 4; ++k) o[k] = fmaf(g.beta, c[k], o[k]); }
            st4(cp, o);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (col + k < ncols_total) {
                    float o = v[k] + bv[k];
                    if (g.beta != 0.f) o = fmaf(g.beta, cp[k], o);
                    cp[k] = o;
                }
        }
    }
}

// Epilogue operands are PREFETCHED while the K loop runs (the epilogue used to start its loads only
// after the last MFMA: ~10 us of exposed memory latency per launch).  The main loops count outstanding
// loads (s_waitcnt vmcnt(N)), so the number of prefetch loads per lane must be a compile-time constant:
// every kernel body is instantiated per MODE and a workgroup branches ONCE, at kernel entry, into the
// body of its problem (no register merges between the variants, no dummy loads).
//   MODE 0  plain store          MODE 1  GRU epilogue           MODE 2  GRU epilogue with the optional operand
//                                        (backward: no ext;              (backward: + ext; forward: + table[token])
//                                         forward: no table)
// Rows / columns are clamped for the loads; stores are guarded.
template <int NP> struct GruBwdPf { f32x4 cr[NP], ex[NP], gr[NP], gz[NP], gn[NP], gh[NP], hp[NP]; };
template <int NP> struct GruFwdPf { f32x4 hr[NP], hz[NP], hn[NP], hp[NP], tr[NP], tz[NP], tn[NP]; };
constexpr int bwd_pf_loads(int mode) { return mode == 0 ? 0 : mode == 1 ? 6 : 7; }    // 16-byte loads per pass
constexpr int fwd_pf_loads(int mode) { return mode == 0 ? 0 : mode == 1 ? 4 : 7; }

template <int NP, bool EXT>
__device__ __forceinline__ void gru_bwd_prefetch(const ppk::NtProblem& g, int p0, int row0, int col0, int lane, GruBwdPf<NP>& f)
{
    const int H = g.H, mlast = g.M - 1;
    int col = col0 + (lane & 7) * 4; col = col < H ? col : H - 4;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        int row = row0 + 8 * (p0 + p) + (lane >> 3); row = row < mlast ? row : mlast;
        const long long rh = (long long)row * H + col;
        f.cr[p] = ld4(g.carry + rh);
        if constexpr (EXT) f.ex[p] = ld4(g.ext + rh);
        const float* gt = g.gates_in + (long long)row * 4 * H + col;
        f.gr[p] = ld4(gt); f.gz[p] = ld4(gt + H); f.gn[p] = ld4(gt + 2 * H); f.gh[p] = ld4(gt + 3 * H);
        f.hp[p] = ld4(g.hp_in + rh);
    }
}

// backward of the GRU gate math: dh = acc + carry (+ ext) ; H % 4 == 0
template <int NP, int NPART, bool EXT>
__device__ __forceinline__ void epi_gru_bwd(const ppk::NtProblem& g, const float* t, int pstride, int p0, int row0, int col0, int lane,
                                            const GruBwdPf<NP>& f)
{
    const int H = g.H, c4 = (lane & 7) * 4, col = col0 + c4;
    if (col >= H) return;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int rl = 8 * (p0 + p) + (lane >> 3), row = row0 + rl;
        if (row >= g.M) continue;
        f32x4 a = ldsum<NPART>(t, pstride, rl * EPI_LD + c4);
        if (row >= g.a_rows) a = f32x4{0.f, 0.f, 0.f, 0.f};
        if (g.drop) a *= ld4(g.drop + (long long)row * H + col);       // the layer above read hnew * drop (problem-uniform branch)
        f32x4 dr, dz, dn, dnr, c;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float dh = a[k] + f.cr[p][k];
            if constexpr (EXT) dh += f.ex[p][k];
            const float rr = f.gr[p][k], z = f.gz[p][k], n = f.gn[p][k];
            dn[k] = dh * (1.0f - z) * (1.0f - n * n);
            dz[k] = dh * (f.hp[p][k] - n) * z * (1.0f - z);
            dr[k] = dn[k] * f.gh[p][k] * rr * (1.0f - rr);
            dnr[k] = dn[k] * rr;
            c[k] = dh * z;
        }
        float* x = g.dgi + (long long)row * 3 * H + col;
        float* y = g.dgh + (long long)row * 3 * H + col;
        st4(x, dr); st4(x + H, dz); st4(x + 2 * H, dn);
        st4(y, dr); st4(y + H, dz); st4(y + 2 * H, dnr);
        st4(g.carry + (long long)row * H + col, c);
    }
}

// Token ids of the rows of this lane's passes.  They are fetched with SCALAR loads (wave-uniform
// addresses, issued before the first staging load) and handed to the lanes with selects: a vector
// load here would make the compiler wait for vmcnt(0) — i.e. for the staging loads in flight — before
// the table addresses can be formed.  rbase = first row of the wave's 32-row tile (wave-uniform).
template <int NP> struct TokS { int v[NP * 8]; };
template <int NP>
__device__ __forceinline__ void gru_fwd_tokens_issue(const ppk::NtProblem& g, int p0, int rbase, TokS<NP>& ts)
{
    const int mlast = g.M - 1;
    // constant address space: the tokens are never written while these kernels run -> s_load
    typedef const __attribute__((address_space(4))) long long* ctok_t;
    const ctok_t tk = (ctok_t)(uintptr_t)g.tok;
#pragma unroll
    for (int i = 0; i < NP * 8; ++i) {
        int row = rbase + 8 * p0 + i; row = row < mlast ? row : mlast;
        int v = (int)tk[row];
        v = v < 0 ? 0 : v; v = v < g.V ? v : g.V - 1;              // never index past the table (scalar ops)
        ts.v[i] = v;
    }
}
template <int NP>
__device__ __forceinline__ void gru_fwd_tokens_select(const TokS<NP>& ts, int lane, long long (&tok)[NP])
{
    const int j = lane >> 3;
    int mask[8];                                   // and/or instead of selects: a select chain is turned into an indexed (scratch) array
#pragma unroll
    for (int k = 0; k < 8; ++k) mask[k] = j == k ? -1 : 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        int t = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t |= ts.v[p * 8 + k] & mask[k];
        tok[p] = t;
    }
}

template <int NP, bool TABLE>
__device__ __forceinline__ void gru_fwd_prefetch(const ppk::NtProblem& g, int p0, int row0, int u0, int lane, const long long (&tok)[NP],
                                                 GruFwdPf<NP>& f)
{
    const int H = g.H, mlast = g.M - 1;
    int u = u0 + (lane & 7) * 4; u = u < H ? u : H - 4;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        int row = row0 + 8 * (p0 + p) + (lane >> 3); row = row < mlast ? row : mlast;
        const float* ghr = g.gh + (long long)row * 3 * H + u;
        f.hr[p] = ld4(ghr); f.hz[p] = ld4(ghr + H); f.hn[p] = ld4(ghr + 2 * H);
        f.hp[p] = ld4(g.hprev + (long long)row * H + u);
    }
    if constexpr (TABLE) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const float* tp = g.table + tok[p] * 3 * H + u;
            f.tr[p] = ld4(tp); f.tz[p] = ld4(tp + H); f.tn[p] = ld4(tp + 2 * H);
        }
    }
}

// GRU cell on transposed r / z / n tiles (t, t + EPI_TILE, t + 2 EPI_TILE); u0 = first hidden unit of the tile; H % 4 == 0
template <int NP, int NPART, bool TABLE>
__device__ __forceinline__ void epi_gru_fwd(const ppk::NtProblem& g, const float* t, int pstride, int p0, int row0, int u0, int lane,
                                            const GruFwdPf<NP>& f)
{
    const int H = g.H, c4 = (lane & 7) * 4, u = u0 + c4;
    if (u >= H) return;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0, b2 = b0;
    if (g.bias) { b0 = ld4(g.bias + u); b1 = ld4(g.bias + H + u); b2 = ld4(g.bias + 2 * H + u); }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int rl = 8 * (p0 + p) + (lane >> 3), row = row0 + rl;
        if (row >= g.M) continue;
        const int o = rl * EPI_LD + c4;
        const f32x4 ar = ldsum<NPART>(t, pstride, o), az = ldsum<NPART>(t + EPI_TILE, pstride, o), an = ldsum<NPART>(t + 2 * EPI_TILE, pstride, o);
        f32x4 rr, zz, nn, hh;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float ir = ar[k] + b0[k], iz = az[k] + b1[k], in_ = an[k] + b2[k];
            if constexpr (TABLE) { ir += f.tr[p][k]; iz += f.tz[p][k]; in_ += f.tn[p][k]; }
            rr[k] = sigmoidf_(ir + f.hr[p][k]);
            zz[k] = sigmoidf_(iz + f.hz[p][k]);
            nn[k] = tanhf(in_ + rr[k] * f.hn[p][k]);
            hh[k] = (1.0f - zz[k]) * nn[k] + zz[k] * f.hp[p][k];
        }
        st4(g.hnew + (long long)row * H + u, hh);
        if (g.drop) st4(g.hdrop + (long long)row * H + u, hh * ld4(g.drop + (long long)row * H + u));      // problem-uniform branch
        st4(g.hp_out + (long long)row * H + u, f.hp[p]);
        float* gt = g.gates_out + (long long)row * 4 * H + u;
        st4(gt, rr); st4(gt + H, zz); st4(gt + 2 * H, nn); st4(gt + 3 * H, f.hn[p]);
    }
}

// ---------------------------------------------------------------------------
// k_nt64: generic 64 x 64 tiles
// ---------------------------------------------------------------------------
constexpr int NT64_BUF = (64 + 64) * BK, NT64_NBUF = 4;

template <int MODE>
__device__ __forceinline__ void nt64_body(const ppk::NtProblem& g, float* lds, const int m0, const int n0)
{
    constexpr int BM = 64;
    constexpr int PPW = 4;                         // LDS-DMA instructions per wave per K tile (2 A + 2 B pieces)
    constexpr int BUF = NT64_BUF;
    constexpr int PF = 4 * bwd_pf_loads(MODE);     // prefetch loads per lane

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;

    const int r8 = lane >> 3, pc = lane & 7;
    const float* ga[2];
    const float* gb[2];
    const int arows = g.a_rows < g.M ? g.a_rows : g.M;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = (wave + 4 * p) * 8 + r8;
        const int lc = pc ^ ((row >> 1) & 7);
        int gr = m0 + row; gr = gr < arows ? gr : arows - 1; gr = gr < 0 ? 0 : gr;
        ga[p] = g.A + (long long)gr * g.lda + lc * 4;
        int gc = n0 + row; gc = gc < g.N ? gc : g.N - 1;
        gb[p] = g.B + (long long)gc * g.ldb + lc * 4;
    }
    auto stage = [&](int buf) {
        float* la = lds + buf * BUF;
        float* lb = la + BM * BK;
#pragma unroll
        for (int p = 0; p < 2; ++p) { __builtin_amdgcn_global_load_lds(ga[p], PPK_LDS_PTR(la + (wave + 4 * p) * 256), 16, 0, 0); ga[p] += BK; }
#pragma unroll
        for (int p = 0; p < 2; ++p) { __builtin_amdgcn_global_load_lds(gb[p], PPK_LDS_PTR(lb + (wave + 4 * p) * 256), 16, 0, 0); gb[p] += BK; }
    };
    const int sw = (li >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xo[c] = 4 * ((2 * c + h) ^ sw);
    const int arow = (wm * 32 + li) * BK;
    const int brow = BM * BK + (wn * 32 + li) * BK;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int row0 = m0 + wm * 32, col0 = n0 + wn * 32;
    GruBwdPf<4> f;

    const int nk = arows > 0 ? g.K / BK : 0;
    if (nk > 0) stage(0);
    if (nk > 1) stage(1);
    if (nk > 2) stage(2);
    if constexpr (MODE != 0) gru_bwd_prefetch<4, MODE == 2>(g, 0, row0, col0, lane, f);     // newer than stages 0..2
    int cur = 0;
    for (int it = 0; it < nk; ++it) {
        // stage `it` must have landed; newer in flight: two stages (+ the prefetch during iterations 0..2)
        if (it + 2 < nk) { if (PF != 0 && it <= 2) wait_vmcnt<2 * PPW + PF>(); else wait_vmcnt<2 * PPW>(); }
        else if (it + 1 < nk) wait_vmcnt<PPW>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (it + 3 < nk) stage((cur + 3) & 3);
        const float* base = lds + cur * BUF;
        // fragment reads run one chunk ahead of the MFMAs that consume them
        f32x4 a[2], b[2];
        a[0] = *(const f32x4*)(base + arow + xo[0]);
        b[0] = *(const f32x4*)(base + brow + xo[0]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < 3) {
                a[(c + 1) & 1] = *(const f32x4*)(base + arow + xo[c + 1]);
                b[(c + 1) & 1] = *(const f32x4*)(base + brow + xo[c + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the reads ahead: the scheduler would sink them to their use
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c & 1][s], b[c & 1][s], acc, 0, 0, 0);
        }
        cur = (cur + 1) & 3;
    }

    __syncthreads();                                 // every wave is done with the staging buffers
    float* t = lds + wave * EPI_TILE;
    tile_to_lds(t, acc, li, h);
    if constexpr (MODE != 0) epi_gru_bwd<4, 1, MODE == 2>(g, t, 0, 0, row0, col0, lane, f);
    else epi_store<4, 1>(g, t, 0, 0, row0, col0, 0, g.N, lane);
}

__global__ __launch_bounds__(256) void k_nt64(const NtBatch nb)
{
    __shared__ __align__(1024) float lds[NT64_NBUF * NT64_BUF];
    int z, tm, tn;
    if (!nt_tile(nb, z, tm, tn)) return;
    const ppk::NtProblem& g = nb.p[z];
    const int m0 = tm * 64, n0 = tn * 64;
    if (g.epi == ppk::NT_GRU_BWD) { if (g.ext) nt64_body<2>(g, lds, m0, n0); else nt64_body<1>(g, lds, m0, n0); }
    else nt64_body<0>(g, lds, m0, n0);
}

// ---------------------------------------------------------------------------
// k_ntF: gate-layout tiles.  N = 3H columns laid out [r | z | n]; n-tile j holds hidden units
// 32 j .. 32 j + 31 of all three gates, so B rows / C columns are g * H + 32 j + lane.
// ---------------------------------------------------------------------------
constexpr int NTF_BUF = (128 + 96) * BK, NTF_NBUF = 2;

// waits until at most R newer stages of PPW loads each (R < 4, run time) are in flight
template <int PPW, int EXTRA> __device__ __forceinline__ void wait_stages(int r)
{
    if (r <= 0) wait_vmcnt<EXTRA>();
    else if (r == 1) wait_vmcnt<PPW + EXTRA>();
    else if (r == 2) wait_vmcnt<2 * PPW + EXTRA>();
    else wait_vmcnt<3 * PPW + EXTRA>();
}

template <int MODE>
__device__ __forceinline__ void ntF_body(const ppk::NtProblem& g, float* lds, const int m0, const int j0)
{
    constexpr int BM = 128;
    constexpr int BUF = NTF_BUF;
    constexpr int PF = 4 * fwd_pf_loads(MODE);

    const int H = g.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int row0 = m0 + wave * 32;

    GruFwdPf<4> f;
    long long tok[4];
    TokS<4> ts;
    if constexpr (MODE == 2) gru_fwd_tokens_issue<4>(g, 0, m0 + __builtin_amdgcn_readfirstlane(wave) * 32, ts);

    const int r8 = lane >> 3, pc = lane & 7;
    const float* ga[4];
    const float* gb[3];
    const int arows = g.a_rows < g.M ? g.a_rows : g.M;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = (wave + 4 * p) * 8 + r8;
        const int lc = pc ^ ((row >> 1) & 7);
        int gr = m0 + row; gr = gr < arows ? gr : arows - 1; gr = gr < 0 ? 0 : gr;
        ga[p] = g.A + (long long)gr * g.lda + lc * 4;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int row = (wave + 4 * p) * 8 + r8;          // 0..95: gate = row / 32, unit = row % 32
        const int lc = pc ^ ((row >> 1) & 7);
        int u = j0 + (row & 31); u = u < H ? u : H - 1;
        gb[p] = g.B + ((long long)(row >> 5) * H + u) * g.ldb + lc * 4;
    }
    auto stage = [&](int buf) {
        float* la = lds + buf * BUF;
        float* lb = la + BM * BK;
#pragma unroll
        for (int p = 0; p < 4; ++p) { __builtin_amdgcn_global_load_lds(ga[p], PPK_LDS_PTR(la + (wave + 4 * p) * 256), 16, 0, 0); ga[p] += BK; }
#pragma unroll
        for (int p = 0; p < 3; ++p) { __builtin_amdgcn_global_load_lds(gb[p], PPK_LDS_PTR(lb + (wave + 4 * p) * 256), 16, 0, 0); gb[p] += BK; }
    };
    const int sw = (li >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xo[c] = 4 * ((2 * c + h) ^ sw);
    const int arow = (wave * 32 + li) * BK;
    const int brow = BM * BK + li * BK;

    f32x16 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    // D = NBUF - 1 K tiles in flight.  Measured: 5 buffers (4 tiles = 115 KB in flight, one workgroup per CU)
    // are NOT faster than 2 buffers with two co-resident workgroups (74 vs 66 us at 512 tiles, the same
    // ~35 us for a lone tile): the loop is not bound by memory latency x bytes in flight.
    constexpr int D = NTF_NBUF - 1, PPW = 7;
    static_assert(D >= 1 && D <= 5 && (D - 1) * PPW + PF < 64, "wait counts");
    const int nk = arows > 0 ? g.K / BK : 0;
#pragma unroll
    for (int d = 0; d < D; ++d) if (d < nk) stage(d);
    if constexpr (MODE == 2) gru_fwd_tokens_select<4>(ts, lane, tok);
    if constexpr (MODE != 0) gru_fwd_prefetch<4, MODE == 2>(g, 0, row0, j0, lane, tok, f);   // newer than stages 0 .. D-1
    int cur = 0, nxt = D;                            // buffer of stage it / of stage it + D
    for (int it = 0; it < nk; ++it) {
        // stage `it` must have landed; newer in flight: D - 1 stages (+ the prefetch while it < D), fewer at the end
        const int r = nk - 1 - it;
        if (r >= D - 1) { if (PF != 0 && it < D) wait_vmcnt<(D - 1) * PPW + PF>(); else wait_vmcnt<(D - 1) * PPW>(); }
        else if (PF != 0 && it < D) wait_stages<PPW, PF>(r);
        else wait_stages<PPW, 0>(r);
        __builtin_amdgcn_s_barrier();
        if (it + D < nk) stage(nxt);
        nxt = nxt + 1 == NTF_NBUF ? 0 : nxt + 1;
        const float* base = lds + cur * BUF;
        // fragment reads run one chunk ahead of the MFMAs that consume them
        f32x4 a[2], b[2][3];
        a[0] = *(const f32x4*)(base + arow + xo[0]);
#pragma unroll
        for (int q = 0; q < 3; ++q) b[0][q] = *(const f32x4*)(base + brow + q * 32 * BK + xo[0]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < 3) {
                a[(c + 1) & 1] = *(const f32x4*)(base + arow + xo[c + 1]);
#pragma unroll
                for (int q = 0; q < 3; ++q) b[(c + 1) & 1][q] = *(const f32x4*)(base + brow + q * 32 * BK + xo[c + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the reads ahead: the scheduler would sink them to their use
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c & 1][s], b[c & 1][q][s], acc[q], 0, 0, 0);
        }
        cur = cur + 1 == NTF_NBUF ? 0 : cur + 1;
    }

    __syncthreads();                                 // every wave is done with the staging buffers
    float* t = lds + wave * 3 * EPI_TILE;
#pragma unroll
    for (int q = 0; q < 3; ++q) tile_to_lds(t + q * EPI_TILE, acc[q], li, h);
    if constexpr (MODE != 0) {
        epi_gru_fwd<4, 1, MODE == 2>(g, t, 0, 0, row0, j0, lane, f);
    } else {
        // plain store in gate layout: C[row, q * H + unit]
#pragma unroll
        for (int q = 0; q < 3; ++q) epi_store<4, 1>(g, t + q * EPI_TILE, 0, 0, row0, j0, (long long)q * H, H, lane);
    }
}

__global__ __launch_bounds__(256) void k_ntF(const NtBatch nb)
{
    __shared__ __align__(1024) float lds[NTF_NBUF * NTF_BUF];
    int z, tm, tn;
    if (!nt_tile(nb, z, tm, tn)) return;
    const ppk::NtProblem& g = nb.p[z];
    const int m0 = tm * 128, j0 = tn * 32;
    if (g.epi == ppk::NT_GRU_FWD) { if (g.table) ntF_body<2>(g, lds, m0, j0); else ntF_body<1>(g, lds, m0, j0); }
    else ntF_body<0>(g, lds, m0, j0);
}

// ---------------------------------------------------------------------------
// Small-grid variants for the short tail steps of a packed batch (and for the reference's own
// bundles of <= 21 sequences): one 32-row output tile per workgroup and K SPLIT OVER THE FOUR
// WAVES (wave w takes K tiles w, w + 4, ...).  Each wave stages and reads only its own LDS
// region, so the main loop has no barrier at all; the four partial tiles are added in wave
// order in the epilogue.  A 64 x 64 x 1536 tile keeps one CU's matrix pipes busy for >= 20 us
// whatever the grid; this shape spreads the same work over 4x (k_nt32) / 4x (k_ntFs) more CUs.
// ---------------------------------------------------------------------------
template <int BROWS>      // B rows per tile: 32 (k_nt32) or 96 (k_ntFs)
struct SmallTile {
    static constexpr int PA = 4, PB = BROWS / 8;            // 8-row LDS-DMA pieces per K tile
    static constexpr int BUF = (32 + BROWS) * BK;           // floats per buffer
    static constexpr int WAVE_LDS = 2 * BUF;                // two buffers per wave
};

template <int MODE>
__device__ __forceinline__ void nt32_body(const ppk::NtProblem& g, float* lds, const int m0, const int n0)
{
    using ST = SmallTile<32>;
    constexpr int PF = bwd_pf_loads(MODE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    float* wl = lds + wave * ST::WAVE_LDS;

    const int r8 = lane >> 3, pc = lane & 7;
    const float* ga[4];
    const float* gb[4];
    const int arows = g.a_rows < g.M ? g.a_rows : g.M;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = p * 8 + r8;
        const int lc = pc ^ ((row >> 1) & 7);
        int gr = m0 + row; gr = gr < arows ? gr : arows - 1; gr = gr < 0 ? 0 : gr;
        ga[p] = g.A + (long long)gr * g.lda + lc * 4 + wave * BK;
        int gc = n0 + row; gc = gc < g.N ? gc : g.N - 1;
        gb[p] = g.B + (long long)gc * g.ldb + lc * 4 + wave * BK;
    }
    auto stage = [&](int buf) {
        float* la = wl + buf * ST::BUF;
        float* lb = la + 32 * BK;
#pragma unroll
        for (int p = 0; p < 4; ++p) { __builtin_amdgcn_global_load_lds(ga[p], PPK_LDS_PTR(la + p * 256), 16, 0, 0); ga[p] += 4 * BK; }
#pragma unroll
        for (int p = 0; p < 4; ++p) { __builtin_amdgcn_global_load_lds(gb[p], PPK_LDS_PTR(lb + p * 256), 16, 0, 0); gb[p] += 4 * BK; }
    };
    const int sw = (li >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xo[c] = 4 * ((2 * c + h) ^ sw);
    const int arow = li * BK, brow = 32 * BK + li * BK;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int nk = arows > 0 ? g.K / BK : 0;
    const int nkw = (nk - wave + 3) >> 2;           // K tiles of this wave
    GruBwdPf<1> f;
    if (nkw > 0) stage(0);
    if (nkw > 1) stage(1);
    if constexpr (MODE != 0) gru_bwd_prefetch<1, MODE == 2>(g, wave, m0, n0, lane, f);     // newer than stages 0 and 1
    for (int i = 0; i < nkw; ++i) {
        // buffer i must have landed; newer in flight: one stage (+ the prefetch while i < 2)
        if (i + 1 < nkw) { if (PF != 0 && i < 2) wait_vmcnt<8 + PF>(); else wait_vmcnt<8>(); }
        else wait_vmcnt<0>();
        const float* base = wl + (i & 1) * ST::BUF;
        // fragment reads run one chunk ahead of the MFMAs that consume them
        f32x4 a[2], b[2];
        a[0] = *(const f32x4*)(base + arow + xo[0]);
        b[0] = *(const f32x4*)(base + brow + xo[0]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < 3) {
                a[(c + 1) & 1] = *(const f32x4*)(base + arow + xo[c + 1]);
                b[(c + 1) & 1] = *(const f32x4*)(base + brow + xo[c + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the reads ahead: the scheduler would sink them to their use
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c & 1][s], b[c & 1][s], acc, 0, 0, 0);
        }
        // this wave's reads of the buffer have returned (the MFMAs consumed them): refill it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (i + 2 < nkw) stage(i & 1);
    }
    __syncthreads();
    tile_to_lds(lds + wave * EPI_TILE, acc, li, h);
    __syncthreads();
    if constexpr (MODE != 0) epi_gru_bwd<1, 4, MODE == 2>(g, lds, EPI_TILE, wave, m0, n0, lane, f);
    else epi_store<1, 4>(g, lds, EPI_TILE, wave, m0, n0, 0, g.N, lane);
}

__global__ __launch_bounds__(256) void k_nt32(const NtBatch nb)
{
    __shared__ __align__(1024) float lds[4 * SmallTile<32>::WAVE_LDS];
    int z, tm, tn;
    if (!nt_tile(nb, z, tm, tn)) return;
    const ppk::NtProblem& g = nb.p[z];
    const int m0 = tm * 32, n0 = tn * 32;
    if (g.epi == ppk::NT_GRU_BWD) { if (g.ext) nt32_body<2>(g, lds, m0, n0); else nt32_body<1>(g, lds, m0, n0); }
    else nt32_body<0>(g, lds, m0, n0);
}

template <int MODE>
__device__ __forceinline__ void ntFs_body(const ppk::NtProblem& g, float* lds, const int m0, const int j0)
{
    using ST = SmallTile<96>;
    constexpr int PF = fwd_pf_loads(MODE);
    const int H = g.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    float* wl = lds + wave * ST::WAVE_LDS;

    GruFwdPf<1> f;
    long long tok[1];
    TokS<1> ts;
    if constexpr (MODE == 2) gru_fwd_tokens_issue<1>(g, __builtin_amdgcn_readfirstlane(wave), m0, ts);

    const int r8 = lane >> 3, pc = lane & 7;
    const float* ga[4];
    const float* gb[12];
    const int arows = g.a_rows < g.M ? g.a_rows : g.M;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = p * 8 + r8;
        const int lc = pc ^ ((row >> 1) & 7);
        int gr = m0 + row; gr = gr < arows ? gr : arows - 1; gr = gr < 0 ? 0 : gr;
        ga[p] = g.A + (long long)gr * g.lda + lc * 4 + wave * BK;
    }
#pragma unroll
    for (int p = 0; p < 12; ++p) {
        const int row = p * 8 + r8;                       // 0..95: gate = row / 32, unit = row % 32
        const int lc = pc ^ ((row >> 1) & 7);
        int u = j0 + (row & 31); u = u < H ? u : H - 1;
        gb[p] = g.B + ((long long)(row >> 5) * H + u) * g.ldb + lc * 4 + wave * BK;
    }
    auto stage = [&](int buf) {
        float* la = wl + buf * ST::BUF;
        float* lb = la + 32 * BK;
#pragma unroll
        for (int p = 0; p < 4; ++p) { __builtin_amdgcn_global_load_lds(ga[p], PPK_LDS_PTR(la + p * 256), 16, 0, 0); ga[p] += 4 * BK; }
#pragma unroll
        for (int p = 0; p < 12; ++p) { __builtin_amdgcn_global_load_lds(gb[p], PPK_LDS_PTR(lb + p * 256), 16, 0, 0); gb[p] += 4 * BK; }
    };
    const int sw = (li >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xo[c] = 4 * ((2 * c + h) ^ sw);
    const int arow = li * BK, brow = 32 * BK + li * BK;

    f32x16 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    const int nk = arows > 0 ? g.K / BK : 0;
    const int nkw = (nk - wave + 3) >> 2;
    if (nkw > 0) stage(0);
    if (nkw > 1) stage(1);
    if constexpr (MODE == 2) gru_fwd_tokens_select<1>(ts, lane, tok);
    if constexpr (MODE != 0) gru_fwd_prefetch<1, MODE == 2>(g, wave, m0, j0, lane, tok, f);   // newer than stages 0 and 1
    for (int i = 0; i < nkw; ++i) {
        if (i + 1 < nkw) { if (PF != 0 && i < 2) wait_vmcnt<16 + PF>(); else wait_vmcnt<16>(); }
        else wait_vmcnt<0>();
        const float* base = wl + (i & 1) * ST::BUF;
        // fragment reads run one chunk ahead of the MFMAs that consume them
        f32x4 a[2], b[2][3];
        a[0] = *(const f32x4*)(base + arow + xo[0]);
#pragma unroll
        for (int q = 0; q < 3; ++q) b[0][q] = *(const f32x4*)(base + brow + q * 32 * BK + xo[0]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < 3) {
                a[(c + 1) & 1] = *(const f32x4*)(base + arow + xo[c + 1]);
#pragma unroll
                for (int q = 0; q < 3; ++q) b[(c + 1) & 1][q] = *(const f32x4*)(base + brow + q * 32 * BK + xo[c + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the reads ahead: the scheduler would sink them to their use
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c & 1][s], b[c & 1][q][s], acc[q], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (i + 2 < nkw) stage(i & 1);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 3; ++q) tile_to_lds(lds + (wave * 3 + q) * EPI_TILE, acc[q], li, h);
    __syncthreads();
    if constexpr (MODE != 0) {
        epi_gru_fwd<1, 4, MODE == 2>(g, lds, 3 * EPI_TILE, wave, m0, j0, lane, f);
    } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) epi_store<1, 4>(g, lds + q * EPI_TILE, 3 * EPI_TILE, wave, m0, j0, (long long)q * H, H, lane);
    }
}

__global__ __launch_bounds__(256) void k_ntFs(const NtBatch nb)
{
    __shared__ __align__(1024) float lds[4 * SmallTile<96>::WAVE_LDS];
    int z, tm, tn;
    if (!nt_tile(nb, z, tm, tn)) return;
    const ppk::NtProblem& g = nb.p[z];
    const int m0 = tm * 32, j0 = tn * 32;
    if (g.epi == ppk::NT_GRU_FWD) { if (g.table) ntFs_body<2>(g, lds, m0, j0); else ntFs_body<1>(g, lds, m0, j0); }
    else ntFs_body<0>(g, lds, m0, j0);
}

inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }

// Which tile family runs a launch.  Fitted to isolated launches on MI355X (tools/bench_nt_families.py):
//   64 x 64 tiles   time is a STEP function of the grid: a CU holds two workgroups that share its matrix
//                   pipes, so ceil(tiles / 256) "rounds" of  2.4 + 0.0141 K us  on top of  -0.5 + 0.0088 K us
//                   (K = 1536: 37 us up to 256 tiles, 61 us up to 512; K = 512: 14 / 23 / 33 / 42 us)
//   32 x 32 K-split linear in the grid:  4 us + (2.9 + 0.0181 K) ns per tile  (4 x the tiles, 2 x the L2 operand
//                   traffic per flop) — wins below ~1.9 rounds of the big tiles at K = 1536, ~2.3 rounds at K = 512.
// ppk_nt_small_grid_limit(L >= 0) or env PPK_NT_SMALL / PPK_NTF_SMALL replace the model by a plain tile-count
// threshold (tests force either family with 0 / 1 << 30).
long long g_small_limit = -2, g_small_limit_f = -2;
long long small_grid_limit()
{
    if (g_small_limit == -2) { const char* e = getenv("PPK_NT_SMALL"); g_small_limit = e ? atoll(e) : -1; }
    return g_small_limit;
}
bool use_small_64(long long tiles64, double tileK64, double tileK32, long long tiles32)
{
    const long long lim = small_grid_limit();
    if (lim >= 0) return tiles64 <= lim;
    const double kbar = tileK64 / (double)tiles64;                    // tile-weighted mean K
    const double rounds = (double)((tiles64 + 255) / 256);
    const double t64 = -0.5 + 0.0088 * kbar + rounds * (2.4 + 0.0141 * kbar);
    const double t32 = 4.0 + 1e-3 * (2.9 * (double)tiles32 + 0.0181 * tileK32);
    return t32 < t64;
}
// env PPK_NT_FAMILY = 32 / 64 forces a family (tools/bench_nt_families.py).  A 128 x 64 tile with two accumulators
// per wave (k_nt128) was built and measured: 57 us for one 128 x 64 x 1536 tile per CU against 38 us for the
// 64 x 64 tile and 58.9 us for two of them — never faster, removed.
int pick_family(int fam)
{
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("PPK_NT_FAMILY"); forced = e ? atoi(e) : 0; }
    return (forced == 32 || forced == 64) ? forced : fam;
}
bool use_small_F(long long tilesF)
{
    if (g_small_limit_f == -2) { const char* e = getenv("PPK_NTF_SMALL"); g_small_limit_f = e ? atoll(e) : -1; }
    const long long lim = small_grid_limit() >= 0 && g_small_limit_f < 0 ? small_grid_limit() : g_small_limit_f;
    return tilesF <= (lim >= 0 ? lim : 128);
}

int check_common(const ppk::NtProblem& p, const char* who)
{
    if (p.M < 0 || p.N < 0 || p.K < 0) PPK_FAIL(PPK_ERR_SHAPE, "%s: negative size", who);
    if (!p.A || !p.B) PPK_FAIL(PPK_ERR_NULL, "%s: null operand", who);
    if (p.K % BK) PPK_FAIL(PPK_ERR_UNSUP, "%s: K must be a multiple of %d (pad with zeros)", who, BK);
    if (p.lda < p.K || p.ldb < p.K) PPK_FAIL(PPK_ERR_SHAPE, "%s: leading dimension smaller than K", who);
    if ((p.lda & 3) || (p.ldb & 3) || ((uintptr_t)p.A & 15) || ((uintptr_t)p.B & 15))
        PPK_FAIL(PPK_ERR_ALIGN, "%s: operands must be 16-byte aligned with leading dimensions %% 4 == 0", who);
    if (p.a_rows < 0) PPK_FAIL(PPK_ERR_SHAPE, "%s: a_rows < 0", who);
    return PPK_OK;
}

}  // namespace

namespace ppk {

bool gemm_nt_fast_ok(int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb)
{
    return K > 0 && K % BK == 0 && !(lda & 3) && !(ldb & 3) && !((uintptr_t)A & 15) && !((uintptr_t)B & 15);
}

// fills the tile map of a launch for (tile_m x tile_n) tiles; returns the grid size (0 = nothing to do)
static unsigned nt_tile_map(NtBatch& nb, int count, int tile_m, int tile_n, bool gate_layout)
{
    long long slots = 0;
    nb.count = count;
    for (int i = 0; i < NT_MAX_GROUP; ++i) {
        long long tn = 1, tm = 0;
        if (i < count) {
            const NtProblem& p = nb.p[i];
            tn = cdiv(gate_layout ? p.H : p.N, tile_n);
            tm = cdiv(p.M, tile_m);
            if (tn < 1) { tn = 1; tm = 0; }
        }
        nb.tn[i] = (int)tn;
        nb.tiles[i] = (int)(tn * tm);
        nb.cp[i] = (int)cdiv(tn * tm, 8);
        nb.coff[i] = (int)slots;
        slots += nb.cp[i];
    }
    nb.coff[NT_MAX_GROUP] = (int)slots;
    return (unsigned)(8 * slots);
}

int gemm_nt64(const NtProblem* ps, int count, hipStream_t stream)
{
    if (count < 1 || count > NT_MAX_GROUP) PPK_FAIL(PPK_ERR_UNSUP, "gemm_nt64: 1..%d problems", NT_MAX_GROUP);
    NtBatch nb;
    long long tiles64 = 0, tiles32 = 0;
    double flops = 0, tileK64 = 0, tileK32 = 0;
    for (int i = 0; i < count; ++i) {
        const NtProblem& p = ps[i];
        int rc = check_common(p, "gemm_nt64");
        if (rc) return rc;
        if (p.epi == NT_STORE) {
            if (!p.C) PPK_FAIL(PPK_ERR_NULL, "gemm_nt64: null C");
            if (p.ldc < p.N) PPK_FAIL(PPK_ERR_SHAPE, "gemm_nt64: ldc < N");
        } else if (p.epi == NT_GRU_BWD) {
            if (!p.carry || !p.gates_in || !p.hp_in || !p.dgi || !p.dgh) PPK_FAIL(PPK_ERR_NULL, "gemm_nt64: null GRU-backward operand");
            if (p.N != p.H || (p.H % 4)) PPK_FAIL(PPK_ERR_SHAPE, "gemm_nt64: GRU-backward epilogue needs N == H, H %% 4 == 0");
        } else {
            PPK_FAIL(PPK_ERR_UNSUP, "gemm_nt64: epilogue %d not available on 64x64 tiles", p.epi);
        }
        if (p.M > (1 << 24) || p.N > (1 << 24)) PPK_FAIL(PPK_ERR_UNSUP, "gemm_nt64: M, N <= 2^24");
        nb.p[i] = p;
        const long long t64 = cdiv(p.N, 64) * cdiv(p.M, 64), t32 = cdiv(p.N, 32) * cdiv(p.M, 32);
        tiles64 += t64;
        tiles32 += t32;
        tileK64 += (double)t64 * (double)p.K;
        tileK32 += (double)t32 * (double)p.K;
        flops += 2.0 * (double)p.M * (double)p.N * (double)p.K;
    }
    if (tiles64 == 0) return PPK_OK;
    if (tiles64 > 0x0fffffff) PPK_FAIL(PPK_ERR_UNSUP, "gemm_nt64: too many tiles");
    int family = use_small_64(tiles64, tileK64, tileK32, tiles32) ? 32 : 64;
    if (small_grid_limit() < 0) family = pick_family(family);
    if (family == 32) {
        // 32 x 32 tiles with K split over the waves
        const unsigned grid = nt_tile_map(nb, count, 32, 32, false);
        ProfScope prof(PK_GEMM_NT32, flops, stream);
        k_nt32<<<dim3(grid), dim3(256), 0, stream>>>(nb);
        PPK_CHECK_LAUNCH("k_nt32");
        return PPK_OK;
    }
    const unsigned grid = nt_tile_map(nb, count, 64, 64, false);
    ProfScope prof(PK_GEMM_NT64, flops, stream);
    k_nt64<<<dim3(grid), dim3(256), 0, stream>>>(nb);
    PPK_CHECK_LAUNCH("k_nt64");
    return PPK_OK;
}

int gemm_ntF(const NtProblem* ps, int count, hipStream_t stream)
{
    if (count < 1 || count > NT_MAX_GROUP) PPK_FAIL(PPK_ERR_UNSUP, "gemm_ntF: 1..%d problems", NT_MAX_GROUP);
    NtBatch nb;
    long long tilesF = 0;
    double flops = 0;
    for (int i = 0; i < count; ++i) {
        const NtProblem& p = ps[i];
        int rc = check_common(p, "gemm_ntF");
        if (rc) return rc;
        if (p.H < 1 || p.N != 3 * p.H) PPK_FAIL(PPK_ERR_SHAPE, "gemm_ntF: gate layout needs N == 3 H");
        if (p.H % 4) PPK_FAIL(PPK_ERR_UNSUP, "gemm_ntF: H must be a multiple of 4");
        if (p.epi == NT_STORE) {
            if (!p.C) PPK_FAIL(PPK_ERR_NULL, "gemm_ntF: null C");
            if (p.ldc < p.N) PPK_FAIL(PPK_ERR_SHAPE, "gemm_ntF: ldc < N");
        } else if (p.epi == NT_GRU_FWD) {
            if (!p.gh || !p.hprev || !p.hnew || !p.hp_out || !p.gates_out) PPK_FAIL(PPK_ERR_NULL, "gemm_ntF: null GRU-forward operand");
            if (p.drop && !p.hdrop) PPK_FAIL(PPK_ERR_NULL, "gemm_ntF: dropout mask without an output for the masked hidden state");
            if (p.table && !p.tok) PPK_FAIL(PPK_ERR_NULL, "gemm_ntF: table without tokens");
            if (p.table && p.V < 1) PPK_FAIL(PPK_ERR_SHAPE, "gemm_ntF: table without its row count");
        } else {
            PPK_FAIL(PPK_ERR_UNSUP, "gemm_ntF: epilogue %d not available on gate-layout tiles", p.epi);
        }
        if (p.M > (1 << 24) || p.N > (1 << 24)) PPK_FAIL(PPK_ERR_UNSUP, "gemm_ntF: M, N <= 2^24");
        nb.p[i] = p;
        tilesF += cdiv(p.H, 32) * cdiv(p.M, 128);
        flops += 2.0 * (double)p.M * (double)p.N * (double)p.K;
    }
    if (tilesF == 0) return PPK_OK;
    if (tilesF > 0x0fffffff) PPK_FAIL(PPK_ERR_UNSUP, "gemm_ntF: too many tiles");
    if (use_small_F(tilesF)) {
        const unsigned grid = nt_tile_map(nb, count, 32, 32, true);
        ProfScope prof(PK_GEMM_NTFS, flops, stream);
        k_ntFs<<<dim3(grid), dim3(256), 0, stream>>>(nb);
        PPK_CHECK_LAUNCH("k_ntFs");
        return PPK_OK;
    }
    const unsigned grid = nt_tile_map(nb, count, 128, 32, true);
    ProfScope prof(PK_GEMM_NTF, flops, stream);
    k_ntF<<<dim3(grid), dim3(256), 0, stream>>>(nb);
    PPK_CHECK_LAUNCH("k_ntF");
    return PPK_OK;
}

}  // namespace ppk

extern "C" {

// Tuning / test hook: NT GEMM launches whose big-tile grid has at most `limit` tiles use the
// small-tile kernels (K split over the waves); -1 restores the built-in thresholds (the default);
// any other negative value only queries.  Returns the value in force.
int64_t ppk_nt_small_grid_limit(int64_t limit)
{
    if (limit >= -1) g_small_limit = limit;
    return small_grid_limit();
}

}  // extern "C"
