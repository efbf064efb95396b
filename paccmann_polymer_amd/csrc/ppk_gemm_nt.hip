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
// Two tile shapes (measured on MI355X, tools/gemm_lab3.hip):
//   k_nt64  64 x 64, 4 waves 2x2, 1 accumulator per wave, 4 LDS buffers    — generic, many small tiles
//   k_ntF   128 x (3 x 32) "gate layout", 4 waves along M, 3 accumulators   — a lane holds r, z, n of one
//           (row, hidden unit), so the GRU cell is evaluated in registers.
// Measured limiter: every ds_read_b128 costs the matrix pipe ~18 cycles (the
// LDS return shares the register-file write path with the MFMA write-back), so
// fewer LDS bytes per MFMA (3 accumulators per A fragment) is what pays.
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
    static_assert(N == 0 || N == 4 || N == 7 || N == 8 || N == 16, "add the count");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct NtBatch { ppk::NtProblem p[ppk::NT_MAX_GROUP]; };

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

// backward of the GRU gate math: dh = acc + carry + ext ; H % 4 == 0
template <int NP, int NPART>
__device__ __forceinline__ void epi_gru_bwd(const ppk::NtProblem& g, const float* t, int pstride, int p0, int row0, int col0, int lane)
{
    const int H = g.H, c4 = (lane & 7) * 4, col = col0 + c4;
    if (col >= H) return;
    const int mlast = g.M - 1;
    f32x4 cr[NP], ex[NP], gr[NP], gz[NP], gn[NP], gh[NP], hp[NP];
    // issue every load of all passes before the first use (rows clamped, stores guarded)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        int row = row0 + 8 * (p0 + p) + (lane >> 3); row = row < mlast ? row : mlast;
        const long long rh = (long long)row * H + col;
        cr[p] = ld4(g.carry + rh);
        ex[p] = g.ext ? ld4(g.ext + rh) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float* gt = g.gates_in + (long long)row * 4 * H + col;
        gr[p] = ld4(gt); gz[p] = ld4(gt + H); gn[p] = ld4(gt + 2 * H); gh[p] = ld4(gt + 3 * H);
        hp[p] = ld4(g.hp_in + rh);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int rl = 8 * (p0 + p) + (lane >> 3), row = row0 + rl;
        if (row >= g.M) continue;
        f32x4 a = ldsum<NPART>(t, pstride, rl * EPI_LD + c4);
        if (row >= g.a_rows) a = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 dr, dz, dn, dnr, c;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dh = a[k] + cr[p][k] + ex[p][k];
            const float rr = gr[p][k], z = gz[p][k], n = gn[p][k];
            dn[k] = dh * (1.0f - z) * (1.0f - n * n);
            dz[k] = dh * (hp[p][k] - n) * z * (1.0f - z);
            dr[k] = dn[k] * gh[p][k] * rr * (1.0f - rr);
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

// GRU cell on transposed r / z / n tiles (t, t + EPI_TILE, t + 2 EPI_TILE); u0 = first hidden unit of the tile; H % 4 == 0
template <int NP, int NPART>
__device__ __forceinline__ void epi_gru_fwd(const ppk::NtProblem& g, const float* t, int pstride, int p0, int row0, int u0, int lane)
{
    const int H = g.H, c4 = (lane & 7) * 4, u = u0 + c4;
    if (u >= H) return;
    const int mlast = g.M - 1;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0, b2 = b0;
    if (g.bias) { b0 = ld4(g.bias + u); b1 = ld4(g.bias + H + u); b2 = ld4(g.bias + 2 * H + u); }
    f32x4 hr[NP], hz[NP], hn[NP], hp[NP], tr[NP], tz[NP], tn[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        int row = row0 + 8 * (p0 + p) + (lane >> 3); row = row < mlast ? row : mlast;
        const float* ghr = g.gh + (long long)row * 3 * H + u;
        hr[p] = ld4(ghr); hz[p] = ld4(ghr + H); hn[p] = ld4(ghr + 2 * H);
        hp[p] = ld4(g.hprev + (long long)row * H + u);
        if (g.table) {
            const float* tp = g.table + g.tok[row] * 3 * H + u;
            tr[p] = ld4(tp); tz[p] = ld4(tp + H); tn[p] = ld4(tp + 2 * H);
        } else {
            tr[p] = tz[p] = tn[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int rl = 8 * (p0 + p) + (lane >> 3), row = row0 + rl;
        if (row >= g.M) continue;
        const int o = rl * EPI_LD + c4;
        const f32x4 ar = ldsum<NPART>(t, pstride, o), az = ldsum<NPART>(t + EPI_TILE, pstride, o), an = ldsum<NPART>(t + 2 * EPI_TILE, pstride, o);
        f32x4 rr, zz, nn, hh;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float ir = ar[k] + b0[k] + tr[p][k], iz = az[k] + b1[k] + tz[p][k], in_ = an[k] + b2[k] + tn[p][k];
            rr[k] = sigmoidf_(ir + hr[p][k]);
            zz[k] = sigmoidf_(iz + hz[p][k]);
            nn[k] = tanhf(in_ + rr[k] * hn[p][k]);
            hh[k] = (1.0f - zz[k]) * nn[k] + zz[k] * hp[p][k];
        }
        st4(g.hnew + (long long)row * H + u, hh);
        st4(g.hp_out + (long long)row * H + u, hp[p]);
        float* gt = g.gates_out + (long long)row * 4 * H + u;
        st4(gt, rr); st4(gt + H, zz); st4(gt + 2 * H, nn); st4(gt + 3 * H, hn[p]);
    }
}

// ---------------------------------------------------------------------------
// k_nt64: generic 64 x 64 tiles
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nt64(const NtBatch nb)
{
    constexpr int BM = 64, BN = 64, NBUF = 4;
    constexpr int PPW = 4;                         // LDS-DMA instructions per wave per K tile (2 A + 2 B pieces)
    constexpr int BUF = (BM + BN) * BK;
    __shared__ __align__(1024) float lds[NBUF * BUF];

    const ppk::NtProblem& g = nb.p[blockIdx.z];
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (m0 >= g.M || n0 >= g.N) return;             // grouped launch: grid is the max over problems
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

    const int nk = arows > 0 ? g.K / BK : 0;
    if (nk > 0) stage(0);
    if (nk > 1) stage(1);
    if (nk > 2) stage(2);
    int cur = 0;
    for (int it = 0; it < nk; ++it) {
        if (it + 2 < nk) wait_vmcnt<2 * PPW>();
        else if (it + 1 < nk) wait_vmcnt<PPW>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (it + 3 < nk) stage((cur + 3) & 3);
        const float* base = lds + cur * BUF;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 a = *(const f32x4*)(base + arow + xo[c]);
            const f32x4 b = *(const f32x4*)(base + brow + xo[c]);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        }
        cur = (cur + 1) & 3;
    }

    __syncthreads();                                 // every wave is done with the staging buffers
    float* t = lds + wave * EPI_TILE;
    tile_to_lds(t, acc, li, h);
    const int row0 = m0 + wm * 32, col0 = n0 + wn * 32;
    if (g.epi == ppk::NT_GRU_BWD) epi_gru_bwd<4, 1>(g, t, 0, 0, row0, col0, lane);
    else epi_store<4, 1>(g, t, 0, 0, row0, col0, 0, g.N, lane);
}

// ---------------------------------------------------------------------------
// k_ntF: gate-layout tiles.  N = 3H columns laid out [r | z | n]; n-tile j holds hidden units
// 32 j .. 32 j + 31 of all three gates, so B rows / C columns are g * H + 32 j + lane.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ntF(const NtBatch nb)
{
    constexpr int BM = 128, BNR = 96, NBUF = 2;
    constexpr int BUF = (BM + BNR) * BK;
    __shared__ __align__(1024) float lds[NBUF * BUF];

    const ppk::NtProblem& g = nb.p[blockIdx.z];
    const int H = g.H;
    const int m0 = blockIdx.y * BM, j0 = blockIdx.x * 32;
    if (m0 >= g.M || j0 >= H) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;

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

    const int nk = arows > 0 ? g.K / BK : 0;
    if (nk > 0) stage(0);
    int cur = 0;
    for (int it = 0; it < nk; ++it) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (it + 1 < nk) stage(cur ^ 1);
        const float* base = lds + cur * BUF;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 a = *(const f32x4*)(base + arow + xo[c]);
            f32x4 b[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) b[q] = *(const f32x4*)(base + brow + q * 32 * BK + xo[c]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[q][s], acc[q], 0, 0, 0);
        }
        cur ^= 1;
    }

    __syncthreads();                                 // every wave is done with the staging buffers
    float* t = lds + wave * 3 * EPI_TILE;
#pragma unroll
    for (int q = 0; q < 3; ++q) tile_to_lds(t + q * EPI_TILE, acc[q], li, h);
    const int row0 = m0 + wave * 32;
    if (g.epi == ppk::NT_GRU_FWD) {
        epi_gru_fwd<4, 1>(g, t, 0, 0, row0, j0, lane);
    } else {
        // plain store in gate layout: C[row, q * H + unit]
#pragma unroll
        for (int q = 0; q < 3; ++q) epi_store<4, 1>(g, t + q * EPI_TILE, 0, 0, row0, j0, (long long)q * H, H, lane);
    }
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

__global__ __launch_bounds__(256) void k_nt32(const NtBatch nb)
{
    using ST = SmallTile<32>;
    __shared__ __align__(1024) float lds[4 * ST::WAVE_LDS];

    const ppk::NtProblem& g = nb.p[blockIdx.z];
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    if (m0 >= g.M || n0 >= g.N) return;
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
    if (nkw > 0) stage(0);
    if (nkw > 1) stage(1);
    for (int i = 0; i < nkw; ++i) {
        if (i + 1 < nkw) wait_vmcnt<8>(); else wait_vmcnt<0>();
        const float* base = wl + (i & 1) * ST::BUF;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 a = *(const f32x4*)(base + arow + xo[c]);
            const f32x4 b = *(const f32x4*)(base + brow + xo[c]);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        }
        // this wave's reads of the buffer have returned (the MFMAs consumed them): refill it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (i + 2 < nkw) stage(i & 1);
    }
    __syncthreads();
    tile_to_lds(lds + wave * EPI_TILE, acc, li, h);
    __syncthreads();
    if (g.epi == ppk::NT_GRU_BWD) epi_gru_bwd<1, 4>(g, lds, EPI_TILE, wave, m0, n0, lane);
    else epi_store<1, 4>(g, lds, EPI_TILE, wave, m0, n0, 0, g.N, lane);
}

__global__ __launch_bounds__(256) void k_ntFs(const NtBatch nb)
{
    using ST = SmallTile<96>;
    __shared__ __align__(1024) float lds[4 * ST::WAVE_LDS];

    const ppk::NtProblem& g = nb.p[blockIdx.z];
    const int H = g.H;
    const int m0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    if (m0 >= g.M || j0 >= H) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    float* wl = lds + wave * ST::WAVE_LDS;

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
    for (int i = 0; i < nkw; ++i) {
        if (i + 1 < nkw) wait_vmcnt<16>(); else wait_vmcnt<0>();
        const float* base = wl + (i & 1) * ST::BUF;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 a = *(const f32x4*)(base + arow + xo[c]);
            f32x4 b[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) b[q] = *(const f32x4*)(base + brow + q * 32 * BK + xo[c]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[q][s], acc[q], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (i + 2 < nkw) stage(i & 1);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 3; ++q) tile_to_lds(lds + (wave * 3 + q) * EPI_TILE, acc[q], li, h);
    __syncthreads();
    if (g.epi == ppk::NT_GRU_FWD) {
        epi_gru_fwd<1, 4>(g, lds, 3 * EPI_TILE, wave, m0, j0, lane);
    } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) epi_store<1, 4>(g, lds + q * EPI_TILE, 3 * EPI_TILE, wave, m0, j0, (long long)q * H, H, lane);
    }
}

inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }

// Which tile family runs a launch: the small-tile kernels take over when the big-tile grid has at
// most this many tiles (rocprofv3 traces of the batch-2048 step, profiles/: below ~half a tile per CU a
// big-tile launch sits on the latency floor of one tile's K loop, ~34 us at K = 1536, while 32-row
// tiles finish in ~12 us; above it the small tiles lose to their 2-4x higher L2 operand traffic).
// ppk_nt_small_grid_limit(L >= 0) or env PPK_NT_SMALL / PPK_NTF_SMALL override (tests, tuning).
long long g_small_limit = -2, g_small_limit_f = -2;
long long small_grid_limit()
{
    if (g_small_limit == -2) { const char* e = getenv("PPK_NT_SMALL"); g_small_limit = e ? atoll(e) : -1; }
    return g_small_limit;
}
bool use_small_64(long long tiles64)
{
    const long long lim = small_grid_limit();
    return tiles64 <= (lim >= 0 ? lim : 224);
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

int gemm_nt64(const NtProblem* ps, int count, hipStream_t stream)
{
    if (count < 1 || count > NT_MAX_GROUP) PPK_FAIL(PPK_ERR_UNSUP, "gemm_nt64: 1..%d problems", NT_MAX_GROUP);
    NtBatch nb;
    long long gx = 0, gy = 0, gx32 = 0, gy32 = 0, tiles64 = 0, tiles32 = 0, tilesF = 0, tilesFs = 0, kmax = 0;
    double flops = 0;
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
        nb.p[i] = p;
        gx = gx > cdiv(p.N, 64) ? gx : cdiv(p.N, 64);
        gy = gy > cdiv(p.M, 64) ? gy : cdiv(p.M, 64);
        gx32 = gx32 > cdiv(p.N, 32) ? gx32 : cdiv(p.N, 32);
        gy32 = gy32 > cdiv(p.M, 32) ? gy32 : cdiv(p.M, 32);
        tiles64 += cdiv(p.N, 64) * cdiv(p.M, 64);
        tiles32 += cdiv(p.N, 32) * cdiv(p.M, 32);
        kmax = kmax > p.K ? kmax : p.K;
        flops += 2.0 * (double)p.M * (double)p.N * (double)p.K;
    }
    (void)tilesF; (void)tilesFs; (void)tiles32; (void)kmax;
    if (gx == 0 || gy == 0) return PPK_OK;
    if (use_small_64(tiles64)) {
        // 32 x 32 tiles with K split over the waves
        ProfScope prof(PK_GEMM_NT32, flops, stream);
        k_nt32<<<dim3((unsigned)gx32, (unsigned)gy32, (unsigned)count), dim3(256), 0, stream>>>(nb);
        PPK_CHECK_LAUNCH("k_nt32");
        return PPK_OK;
    }
    ProfScope prof(PK_GEMM_NT64, flops, stream);
    k_nt64<<<dim3((unsigned)gx, (unsigned)gy, (unsigned)count), dim3(256), 0, stream>>>(nb);
    PPK_CHECK_LAUNCH("k_nt64");
    return PPK_OK;
}

int gemm_ntF(const NtProblem* ps, int count, hipStream_t stream)
{
    if (count < 1 || count > NT_MAX_GROUP) PPK_FAIL(PPK_ERR_UNSUP, "gemm_ntF: 1..%d problems", NT_MAX_GROUP);
    NtBatch nb;
    long long gx = 0, gy = 0, gx32 = 0, gy32 = 0, tiles64 = 0, tiles32 = 0, tilesF = 0, tilesFs = 0, kmax = 0;
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
            if (p.table && !p.tok) PPK_FAIL(PPK_ERR_NULL, "gemm_ntF: table without tokens");
        } else {
            PPK_FAIL(PPK_ERR_UNSUP, "gemm_ntF: epilogue %d not available on gate-layout tiles", p.epi);
        }
        nb.p[i] = p;
        gx = gx > cdiv(p.H, 32) ? gx : cdiv(p.H, 32);
        gy = gy > cdiv(p.M, 128) ? gy : cdiv(p.M, 128);
        gy32 = gy32 > cdiv(p.M, 32) ? gy32 : cdiv(p.M, 32);
        tilesF += cdiv(p.H, 32) * cdiv(p.M, 128);
        tilesFs += cdiv(p.H, 32) * cdiv(p.M, 32);
        kmax = kmax > p.K ? kmax : p.K;
        flops += 2.0 * (double)p.M * (double)p.N * (double)p.K;
    }
    (void)gx32; (void)tiles64; (void)tiles32; (void)tilesFs; (void)kmax;
    if (gx == 0 || gy == 0) return PPK_OK;
    if (use_small_F(tilesF)) {
        ProfScope prof(PK_GEMM_NTFS, flops, stream);
        k_ntFs<<<dim3((unsigned)gx, (unsigned)gy32, (unsigned)count), dim3(256), 0, stream>>>(nb);
        PPK_CHECK_LAUNCH("k_ntFs");
        return PPK_OK;
    }
    ProfScope prof(PK_GEMM_NTF, flops, stream);
    k_ntF<<<dim3((unsigned)gx, (unsigned)gy, (unsigned)count), dim3(256), 0, stream>>>(nb);
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
