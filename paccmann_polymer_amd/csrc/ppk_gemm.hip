// fp32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32: exact fp32, fmaf-chain numerics).
//
//   C[M,N] = alpha * opA(A) . opB(B) + beta * C + bias[N]        (row-major)
//
// Used for every dense contraction of the StackGRU VAE step (GRU gate GEMMs,
// stack heads, mu/logvar heads, latent->hidden, logits, all weight gradients)
// — the arithmetic that reaches cuDNN/MIOpen/ATen through torch.nn.GRU /
// nn.Linear in the reference (paccmann_encoder.py:166-168 -> paccmann_chemistry).
//
// Structure: 256 threads = 4 waves in a 2x2 grid, block tile BM x BN x 32,
// register-prefetched single LDS buffer (global loads of tile k+1 are in
// flight while tile k is multiplied).  LDS images:
//   k-contiguous operand  ->  [row][32 + 4]   read by ds_read_b128 (conflict-free: stride 36)
//   x-contiguous operand  ->  [k][rows]       read by ds_read_b32  (32 consecutive dwords)
// Per 8-deep K chunk lane (i, h = lane>>5) owns k = 4h..4h+3, so one 16-byte
// read feeds four MFMA steps; A and B use the same k permutation.
// Small grids with a long K are split along K into slabs that a second kernel
// sums in slice order (bitwise reproducible, no atomics).
#include "ppk_common.h"
#include "ppk_kernels.h"
#include "ppk_seq_common.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int KC_STRIDE = BK + 4;   // floats

// ---- global -> registers ---------------------------------------------------
// k-contiguous source: element (row, k) at src[row * ld + k]
template <int ROWS>
__device__ __forceinline__ void gload_kc(f32x4 (&reg)[ROWS / 32], const float* __restrict__ src, long long ld,
                                         int row0, int nrows, int k0, int kend, int tid, bool vec)
{
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx >> 3, kq = (idx & 7) << 2;
        const int gr = row0 + row, gk = k0 + kq;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gr < nrows) {
            const float* p = src + (long long)gr * ld + gk;
            if (vec && gk + 3 < kend) {
                v = *reinterpret_cast<const f32x4*>(p);
            } else {
                if (gk < kend) v.x = p[0];
                if (gk + 1 < kend) v.y = p[1];
                if (gk + 2 < kend) v.z = p[2];
                if (gk + 3 < kend) v.w = p[3];
            }
        }
        reg[i] = v;
    }
}
// x-contiguous source: element (x, k) at src[k * ld + x]
template <int ROWS>
__device__ __forceinline__ void gload_xc(f32x4 (&reg)[ROWS / 32], const float* __restrict__ src, long long ld,
                                         int x0, int nx, int k0, int kend, int tid, bool vec)
{
    constexpr int QPR = ROWS / 4;   // float4 per k-row
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int idx = tid + 256 * i;
        const int k = idx / QPR, xq = (idx % QPR) << 2;
        const int gk = k0 + k, gx = x0 + xq;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (gk < kend) {
            const float* p = src + (long long)gk * ld + gx;
            if (vec && gx + 3 < nx) {
                v = *reinterpret_cast<const f32x4*>(p);
            } else {
                if (gx < nx) v.x = p[0];
                if (gx + 1 < nx) v.y = p[1];
                if (gx + 2 < nx) v.z = p[2];
                if (gx + 3 < nx) v.w = p[3];
            }
        }
        reg[i] = v;
    }
}
// unguarded variants for interior tiles (full rows, K a multiple of BK, float4-aligned operands)
template <int ROWS>
__device__ __forceinline__ void gload_kc_fast(f32x4 (&reg)[ROWS / 32], const float* __restrict__ src, long long ld,
                                              int row0, int k0, int tid)
{
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int idx = tid + 256 * i;
        reg[i] = *reinterpret_cast<const f32x4*>(src + (long long)(row0 + (idx >> 3)) * ld + k0 + ((idx & 7) << 2));
    }
}
template <int ROWS>
__device__ __forceinline__ void gload_xc_fast(f32x4 (&reg)[ROWS / 32], const float* __restrict__ src, long long ld,
                                              int x0, int k0, int tid)
{
    constexpr int QPR = ROWS / 4;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int idx = tid + 256 * i;
        reg[i] = *reinterpret_cast<const f32x4*>(src + (long long)(k0 + idx / QPR) * ld + x0 + ((idx % QPR) << 2));
    }
}
// ---- registers -> LDS --------------------------------------------------------
template <int ROWS>
__device__ __forceinline__ void sstore_kc(float* lds, const f32x4 (&reg)[ROWS / 32], int tid)
{
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx >> 3, kq = (idx & 7) << 2;
        *reinterpret_cast<f32x4*>(lds + row * KC_STRIDE + kq) = reg[i];
    }
}
template <int ROWS>
__device__ __forceinline__ void sstore_xc(float* lds, const f32x4 (&reg)[ROWS / 32], int tid)
{
    constexpr int QPR = ROWS / 4;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int idx = tid + 256 * i;
        const int k = idx / QPR, xq = (idx % QPR) << 2;
        *reinterpret_cast<f32x4*>(lds + k * ROWS + xq) = reg[i];
    }
}
// ---- LDS -> MFMA fragment (4 k-steps of one 32-row tile) ---------------------
template <int ROWS, bool KC>
__device__ __forceinline__ f32x4 frag(const float* lds, int row, int chunk, int h)
{
    if constexpr (KC) {
        return *reinterpret_cast<const f32x4*>(lds + row * KC_STRIDE + chunk * 8 + 4 * h);
    } else {
        const float* p = lds + (chunk * 8 + 4 * h) * ROWS + row;
        f32x4 v = {p[0], p[ROWS], p[2 * ROWS], p[3 * ROWS]};
        return v;
    }
}

template <int ROWS, bool KC>
constexpr int lds_floats() { return KC ? ROWS * KC_STRIDE : BK * ROWS; }

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias;
    long long lda, ldb, ldc;
    int M, N, K;
    float alpha, beta;
    int ksplit;          // number of K slices (grid.z) — single-problem launches only
    int kslice;          // K elements per slice (multiple of BK)
    float* slabs;        // [ksplit][M][N] when ksplit > 1
    int a_vec, b_vec;    // operand base 16-B aligned and leading dimension % 4 == 0 -> float4 loads
    float* csum;         // k_tn128 only: csum[m] = sum_k A[k, m] (column sums of the x-contiguous A), or null
    float* csum_slabs;   // [ksplit][M] when ksplit > 1
};
struct GemmBatch { GemmArgs p[1]; };

template <int BM, int BN, bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void k_gemm(const GemmBatch gb)
{
    constexpr int TM = BM / 64, TN = BN / 64;       // 32x32 tiles per wave
    __shared__ __align__(16) float lds[lds_floats<BM, A_KC>() + lds_floats<BN, B_KC>()];
    float* ldsA = lds;
    float* ldsB = lds + lds_floats<BM, A_KC>();

    const GemmArgs& g = gb.p[0];
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, h = lane >> 5;
    const int kbeg = blockIdx.z * g.kslice;
    const int kend = min(g.K, kbeg + g.kslice);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[BM / 32], rb[BN / 32];
    // interior tile: no row / k guards, float4 loads (block-uniform choice)
    const bool fast = g.a_vec && g.b_vec && m0 + BM <= g.M && n0 + BN <= g.N && ((kend - kbeg) % BK) == 0;
    auto gload = [&](int k0) {
        if (fast) {
            if constexpr (A_KC) gload_kc_fast<BM>(ra, g.A, g.lda, m0, k0, tid);
            else                gload_xc_fast<BM>(ra, g.A, g.lda, m0, k0, tid);
            if constexpr (B_KC) gload_kc_fast<BN>(rb, g.B, g.ldb, n0, k0, tid);
            else                gload_xc_fast<BN>(rb, g.B, g.ldb, n0, k0, tid);
        } else {
            if constexpr (A_KC) gload_kc<BM>(ra, g.A, g.lda, m0, g.M, k0, kend, tid, g.a_vec);
            else                gload_xc<BM>(ra, g.A, g.lda, m0, g.M, k0, kend, tid, g.a_vec);
            if constexpr (B_KC) gload_kc<BN>(rb, g.B, g.ldb, n0, g.N, k0, kend, tid, g.b_vec);
            else                gload_xc<BN>(rb, g.B, g.ldb, n0, g.N, k0, kend, tid, g.b_vec);
        }
    };
    gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        if constexpr (A_KC) sstore_kc<BM>(ldsA, ra, tid); else sstore_xc<BM>(ldsA, ra, tid);
        if constexpr (B_KC) sstore_kc<BN>(ldsB, rb, tid); else sstore_xc<BN>(ldsB, rb, tid);
        __syncthreads();
        if (k0 + BK < kend) gload(k0 + BK);
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = frag<BM, A_KC>(ldsA, wm * (BM / 2) + i * 32 + li, c, h);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = frag<BN, B_KC>(ldsB, wn * (BN / 2) + j * 32 + li, c, h);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue.  C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const bool slab = g.ksplit > 1;
    float* Cout = slab ? g.slabs + (long long)blockIdx.z * g.M * g.N : g.C;
    const long long ldo = slab ? g.N : g.ldc;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / 2) + j * 32 + li;
            if (col >= g.N) continue;
            const float bv = (!slab && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= g.M) continue;
                float* p = Cout + (long long)row * ldo + col;
                float v = acc[i][j][r];
                if (!slab) {
                    v = g.alpha * v + bv;
                    if (g.beta != 0.f) v = fmaf(g.beta, *p, v);
                }
                *p = v;
            }
        }
}

// ---------------------------------------------------------------------------
// k_tn128: C = A^T . B with BOTH operands x-contiguous (A [K, M], B [K, N]) — the long-K weight
// gradients  dW = dG^T . X  over all tokens of the batch.  No transposition anywhere: rows of k are
// staged by LDS-DMA (global_load_lds_dwordx4, two 512-byte k rows per wave instruction) into a
// [k][128] image, double buffered, one raw barrier per K tile; an MFMA operand (row m = lane & 31,
// k = 2 s + (lane >> 5)) is one ds_read_b32 with consecutive lanes on consecutive dwords.  Odd k
// rows are stored with their 16-byte chunks XOR 8 (applied to the DMA source address) so the two
// half-waves of a read hit disjoint banks.  128 x 128 tile, 4 waves 2 x 2, four 32 x 32 accumulators
// per wave (64 MFMAs per wave between barriers keep the matrix pipe queued across them).
// Needs M % 4 == 0, N % 4 == 0, 16-byte aligned operands; slabs start at multiples of 32; the K tail
// (< 32 rows of the last slab) is multiplied straight from global memory with zero-filled lanes.
// ---------------------------------------------------------------------------
#define PPK_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr int TN_BUF = 2 * BK * 128;           // floats per stage (A image + B image)

template <bool BF>
__global__ __launch_bounds__(256) void k_tn128(const GemmArgs g)
{
    __shared__ __align__(1024) float lds[2 * TN_BUF];
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
    const int kbeg = blockIdx.z * g.kslice;
    const int kend = min(g.K, kbeg + g.kslice);
    const int nk = (kend - kbeg) / BK;

    // DMA piece q = wave + 4 p (p = 0..3) covers k rows 2 q, 2 q + 1 of the tile; lane -> row 2 q + h, physical chunk li
    const int lchunk = li ^ (8 * h);                                  // logical 16-byte chunk held by this lane's slot
    int ca = m0 + 4 * lchunk; ca = ca < g.M - 4 ? ca : g.M - 4;       // clamped columns (their outputs are never stored)
    int cb = n0 + 4 * lchunk; cb = cb < g.N - 4 ? cb : g.N - 4;
    const float* ga = g.A + (long long)(kbeg + 2 * wave + h) * g.lda + ca;
    const float* gbp = g.B + (long long)(kbeg + 2 * wave + h) * g.ldb + cb;
    const long long sa = 8 * g.lda, sb = 8 * g.ldb;                    // piece p -> + 8 k rows
    auto stage = [&](int buf) {
        float* la = lds + buf * TN_BUF;
        float* lb = la + BK * 128;
#pragma unroll
        for (int p = 0; p < 4; ++p) __builtin_amdgcn_global_load_lds(ga + p * sa, PPK_LDS_PTR(la + (wave + 4 * p) * 256), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < 4; ++p) __builtin_amdgcn_global_load_lds(gbp + p * sb, PPK_LDS_PTR(lb + (wave + 4 * p) * 256), 16, 0, 0);
        ga += (long long)BK * g.lda;
        gbp += (long long)BK * g.ldb;
    };
    // fragment element (k = 2 s + h, m): float offset k * 128 + 4 * ((m >> 2) ^ 8 h) + (m & 3)
    int offa[2], offb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ma = wm * 64 + i * 32 + li, mb = wn * 64 + i * 32 + li;
        offa[i] = h * 128 + 4 * ((ma >> 2) ^ (8 * h)) + (ma & 3);
        offb[i] = BK * 128 + h * 128 + 4 * ((mb >> 2) ^ (8 * h)) + (mb & 3);
    }

    int swa[2][2], swb[2][2];                       // bf16 variant: in-row offsets of (m, parity of k)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pz = 0; pz < 2; ++pz) {
            const int ma = wm * 64 + i * 32 + li, mb = wn * 64 + i * 32 + li;
            swa[i][pz] = 4 * ((ma >> 2) ^ (8 * pz)) + (ma & 3);
            swb[i][pz] = 4 * ((mb >> 2) ^ (8 * pz)) + (mb & 3);
        }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // column sums of A (the bias gradient that goes with a weight gradient): the A fragments pass through the
    // registers of the wn == 0 waves of the first n-tile anyway
    const bool do_cs = g.csum != nullptr && blockIdx.x == 0 && wn == 0;
    const float csw = do_cs ? 1.0f : 0.0f;
    float cs[2] = {0.f, 0.f};

    if (nk > 0) stage(0);
    int cur = 0;
    for (int it = 0; it < nk; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it + 1 < nk) stage(cur ^ 1);
        const float* base = lds + cur * TN_BUF;
        if constexpr (BF) {
            // bf16-operand variant (ppk_mfma_mode(1)): lane (m, h) takes k = 8 c + 4 h + 0..3 of every 8-row chunk,
            // rounds to bf16 in registers (round to nearest even), one v_mfma_f32_32x32x8_bf16 per chunk and tile pair
#pragma unroll
            for (int c = 0; c < BK / 8; ++c) {
                ppk_seq::s16x4 af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x4 av, bv;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        av[t] = base[(8 * c + 4 * h + t) * 128 + swa[i][t & 1]];
                        bv[t] = base[BK * 128 + (8 * c + 4 * h + t) * 128 + swb[i][t & 1]];
                    }
                    cs[i] = fmaf(csw, (av[0] + av[1]) + (av[2] + av[3]), cs[i]);
                    af[i] = ppk_seq::to_bf16x4(av);
                    bf[i] = ppk_seq::to_bf16x4(bv);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { a[i] = base[offa[i] + s * 256]; b[i] = base[offb[i] + s * 256]; }
            cs[0] = fmaf(csw, a[0], cs[0]);              // branch-free: csw = 1 on the waves that own the sums, else 0
            cs[1] = fmaf(csw, a[1], cs[1]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        }
        cur ^= 1;
    }
    // K tail: fewer than 32 rows, straight from global memory (lanes beyond the end contribute zeros)
    for (int k = kbeg + nk * BK; k < kend; k += 2) {
        const int kk = k + h;
        const bool ok = kk < kend;
        const int kr = ok ? kk : kend - 1;
        float a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int ma = m0 + wm * 64 + i * 32 + li; ma = ma < g.M ? ma : g.M - 1;
            int mb = n0 + wn * 64 + i * 32 + li; mb = mb < g.N ? mb : g.N - 1;
            const float va = g.A[(long long)kr * g.lda + ma], vb = g.B[(long long)kr * g.ldb + mb];
            a[i] = ok ? va : 0.f;
            b[i] = ok ? vb : 0.f;
        }
        if (do_cs) { cs[0] += a[0]; cs[1] += a[1]; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }

    const bool slab = g.ksplit > 1;
    float* Cout = slab ? g.slabs + (long long)blockIdx.z * g.M * g.N : g.C;
    const long long ldo = slab ? g.N : g.ldc;
    if (do_cs) {
        float* co = slab ? g.csum_slabs + (long long)blockIdx.z * g.M : g.csum;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float t = cs[i] + __shfl_xor(cs[i], 32, 64);                 // even + odd k rows
            const int m = m0 + wm * 64 + i * 32 + li;
            if (h == 0 && m < g.M) co[m] = t;
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + li;
            if (col >= g.N) continue;
            const float bv = (!slab && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= g.M) continue;
                float* p = Cout + (long long)row * ldo + col;
                float v = acc[i][j][r];
                if (!slab) {
                    v = g.alpha * v + bv;
                    if (g.beta != 0.f) v = fmaf(g.beta, *p, v);
                }
                *p = v;
            }
        }
}

// sums the K slabs in slice order, then applies alpha / beta / bias.
__global__ __launch_bounds__(256) void k_gemm_reduce(const GemmArgs g)
{
    const long long total = (long long)g.M * g.N;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        float s = 0.f;
        for (int z = 0; z < g.ksplit; ++z) s += g.slabs[(long long)z * total + e];
        const int row = (int)(e / g.N), col = (int)(e % g.N);
        float* p = g.C + (long long)row * g.ldc + col;
        float v = g.alpha * s + (g.bias ? g.bias[col] : 0.f);
        if (g.beta != 0.f) v = fmaf(g.beta, *p, v);
        *p = v;
    }
    if (g.csum)
        for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < g.M; m += (long long)gridDim.x * 256) {
            float s = 0.f;
            for (int z = 0; z < g.ksplit; ++z) s += g.csum_slabs[(long long)z * g.M + m];
            g.csum[m] = s;
        }
}

inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }

template <int BM, int BN>
void launch(const GemmBatch& gb, int count, bool a_kc, bool b_kc, hipStream_t stream)
{
    long long gx = 0, gy = 0;
    for (int i = 0; i < count; ++i) {
        gx = gx > cdiv(gb.p[i].N, BN) ? gx : cdiv(gb.p[i].N, BN);
        gy = gy > cdiv(gb.p[i].M, BM) ? gy : cdiv(gb.p[i].M, BM);
    }
    dim3 grid((unsigned)gx, (unsigned)gy, gb.p[0].ksplit), block(256);
    if (a_kc && b_kc)        k_gemm<BM, BN, true, true><<<grid, block, 0, stream>>>(gb);
    else if (a_kc && !b_kc)  k_gemm<BM, BN, true, false><<<grid, block, 0, stream>>>(gb);
    else if (!a_kc && b_kc)  k_gemm<BM, BN, false, true><<<grid, block, 0, stream>>>(gb);
    else                     k_gemm<BM, BN, false, false><<<grid, block, 0, stream>>>(gb);
}

// tile: 0 = 64x64, 1 = 128x64, 2 = 128x128 ; ksplit >= 1
struct Plan { int tile; int ksplit; };

// Measured on MI355X (tools/bench_gemm.py): the per-step shapes (M <= 2048, K <= 1536)
// are latency-bound and run best on 64x64 tiles at >= 3 blocks/CU; long-K
// contractions (weight gradients, K = sum of sequence lengths) want 128-row tiles
// plus enough K slices to put >= 2 blocks on every CU.
Plan make_plan(long long M, long long N, long long K, bool allow_split)
{
    Plan pl{0, 1};
    int ov = 0;
    if (const char* e = getenv("PPK_GEMM_TILE")) ov = atoi(e);       // tuning aid: 1 / 2 / 3 force 64x64 / 128x128 / 128x64
    const long long b22 = cdiv(M, 128) * cdiv(N, 128), b21 = cdiv(M, 128) * cdiv(N, 64), b11 = cdiv(M, 64) * cdiv(N, 64);
    if (ov == 1) pl.tile = 0;
    else if (ov == 2) pl.tile = 2;
    else if (ov == 3) pl.tile = 1;
    else if (K >= 8192) pl.tile = N > 192 ? 2 : 1;
    else if (b22 >= 768) pl.tile = 2;
    else if (b21 >= 768) pl.tile = 1;
    else pl.tile = 0;
    const long long blocks = pl.tile == 2 ? b22 : (pl.tile == 1 ? b21 : b11);
    if (allow_split && blocks < 256 && K >= 2048) {
        long long target = 768;                             // three resident blocks per CU
        if (const char* e = getenv("PPK_GEMM_WGS")) target = atoll(e);
        long long want = target / blocks > 0 ? target / blocks : 1;
        long long maxs = K / 1024;
        long long s = want < maxs ? want : maxs;
        if (s > 64) s = 64;
        if (s > 1) pl.ksplit = (int)s;
    }
    return pl;
}

void fill_args(GemmArgs& g, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
               const float* B, int64_t ldb, float beta, float* C, int64_t ldc, const float* bias)
{
    g.A = A; g.B = B; g.C = C; g.bias = bias;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.alpha = alpha; g.beta = beta;
    g.a_vec = !(lda & 3) && !((uintptr_t)A & 15);
    g.b_vec = !(ldb & 3) && !((uintptr_t)B & 15);
    g.ksplit = 1; g.kslice = (int)(cdiv(K > 0 ? K : 1, BK) * BK); g.slabs = nullptr;
    g.csum = nullptr; g.csum_slabs = nullptr;
}

int check_problem(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                  const float* B, int64_t ldb, const float* C, int64_t ldc)
{
    if (M < 0 || N < 0 || K < 0) PPK_FAIL(PPK_ERR_SHAPE, "gemm: negative size");
    if (!A || !B || !C) PPK_FAIL(PPK_ERR_NULL, "gemm: null operand");
    if (M > 0x7fffffff || N > 0x7fffffff || K > 0x7fffffff) PPK_FAIL(PPK_ERR_UNSUP, "gemm: dimension >= 2^31");
    const bool a_kc = !transA, b_kc = transB != 0;
    if (lda < (a_kc ? K : M) || ldb < (b_kc ? K : N) || ldc < N)
        PPK_FAIL(PPK_ERR_SHAPE, "gemm: leading dimension too small (lda=%lld ldb=%lld ldc=%lld)", (long long)lda, (long long)ldb, (long long)ldc);
    if (((uintptr_t)A & 3) || ((uintptr_t)B & 3) || ((uintptr_t)C & 3))
        PPK_FAIL(PPK_ERR_ALIGN, "gemm: operands must be 4-byte aligned");
    return PPK_OK;
}

void dispatch(const GemmBatch& gb, int count, int tile, bool a_kc, bool b_kc, double flops, hipStream_t stream)
{
    ppk::ProfScope prof((tile ? 4 : 0) + (a_kc ? 2 : 0) + (b_kc ? 1 : 0), flops, stream);
    if (tile == 2)      launch<128, 128>(gb, count, a_kc, b_kc, stream);
    else if (tile == 1) launch<128, 64>(gb, count, a_kc, b_kc, stream);
    else                launch<64, 64>(gb, count, a_kc, b_kc, stream);
}

}  // namespace

namespace ppk {

size_t gemm_workspace_bytes(int64_t M, int64_t N, int64_t K)
{
    if (M <= 0 || N <= 0) return 0;
    const Plan pl = make_plan(M, N, K, true);
    return pl.ksplit > 1 ? ppk_align_up(sizeof(float) * ((size_t)M * (size_t)N + (size_t)M) * (size_t)pl.ksplit, 256) : 0;
}

// csum != null: also csum[m] = sum_k A[k, m]; only with the TN fast path (returns PPK_ERR_UNSUP otherwise,
// before anything is launched, so that the caller can fall back to gemm + colsum)
int gemm_csum(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
              const float* A, int64_t lda, const float* B, int64_t ldb, float beta,
              float* C, int64_t ldc, const float* bias, float* csum, void* ws, size_t ws_bytes, hipStream_t stream);

int gemm(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
         const float* A, int64_t lda, const float* B, int64_t ldb, float beta,
         float* C, int64_t ldc, const float* bias, void* ws, size_t ws_bytes, hipStream_t stream)
{
    return gemm_csum(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, nullptr, ws, ws_bytes, stream);
}

int gemm_csum(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
              const float* A, int64_t lda, const float* B, int64_t ldb, float beta,
              float* C, int64_t ldc, const float* bias, float* csum, void* ws, size_t ws_bytes, hipStream_t stream)
{
    if (M == 0 || N == 0) return PPK_OK;
    int rc = check_problem(transA, transB, M, N, K, A, lda, B, ldb, C, ldc);
    if (rc) return rc;
    const bool a_kc = !transA, b_kc = transB != 0;
    if (csum && (a_kc || b_kc)) PPK_FAIL(PPK_ERR_UNSUP, "gemm_csum: TN layout only");
    GemmBatch gb;
    GemmArgs& g = gb.p[0];
    fill_args(g, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias);
    Plan pl = make_plan(M, N, K, ws != nullptr);
    if (a_kc && b_kc && alpha == 1.f && pl.ksplit == 1 && M <= 0x7fffffff && gemm_nt_fast_ok(K, A, lda, B, ldb)) {
        // LDS-DMA staged NT kernel (ppk_gemm_nt.hip)
        NtProblem p{};
        p.A = A; p.B = B; p.lda = lda; p.ldb = ldb;
        p.M = (int)M; p.N = (int)N; p.K = (int)K; p.a_rows = (int)M;
        p.epi = NT_STORE; p.H = 0;
        p.C = C; p.ldc = ldc; p.bias = bias; p.beta = beta;
        return gemm_nt64(&p, 1, stream);
    }
    if (pl.ksplit > 1) {
        const size_t need = sizeof(float) * ((size_t)M * (size_t)N + (size_t)M) * (size_t)pl.ksplit;
        if (need <= ws_bytes) {
            g.kslice = (int)(cdiv(cdiv(K, pl.ksplit), BK) * BK);
            g.ksplit = (int)cdiv(K, g.kslice);
            g.slabs = (float*)ws;
            g.csum_slabs = g.slabs + (size_t)M * (size_t)N * (size_t)pl.ksplit;
        }
    }
    const char* tn_env = getenv("PPK_GEMM_TN");                       // tuning aid: 0 keeps the register-staged kernel
    const bool tn_fast = !a_kc && !b_kc && pl.tile == 2 && g.a_vec && g.b_vec && !(M & 3) && !(N & 3) && M >= 4 && N >= 4 &&
                         !(tn_env && atoi(tn_env) == 0);
    if (csum && !tn_fast) PPK_FAIL(PPK_ERR_UNSUP, "gemm_csum: the fused column sum needs the TN fast path (long K, N > 192, M, N %% 4 == 0, aligned operands)");
    g.csum = csum;
    if (tn_fast) {
        ppk::ProfScope prof(PK_GEMM_TN128, 2.0 * (double)M * (double)N * (double)K, stream);
        if (ppk::mfma_mode() == 1) k_tn128<true><<<dim3((unsigned)cdiv(N, 128), (unsigned)cdiv(M, 128), (unsigned)g.ksplit), dim3(256), 0, stream>>>(g);
        else k_tn128<false><<<dim3((unsigned)cdiv(N, 128), (unsigned)cdiv(M, 128), (unsigned)g.ksplit), dim3(256), 0, stream>>>(g);
        PPK_CHECK_LAUNCH("k_tn128");
    } else {
        dispatch(gb, 1, pl.tile, a_kc, b_kc, 2.0 * (double)M * (double)N * (double)K, stream);
        PPK_CHECK_LAUNCH("k_gemm");
    }
    if (g.ksplit > 1) {
        long long total = (long long)M * N;
        unsigned nb = (unsigned)(cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048);
        k_gemm_reduce<<<dim3(nb), dim3(256), 0, stream>>>(g);
        PPK_CHECK_LAUNCH("k_gemm_reduce");
    }
    return PPK_OK;
}

}  // namespace ppk

extern "C" {

size_t ppk_gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K)
{
    return ppk::gemm_workspace_bytes(M, N, K);
}

// C = A^T . B  and  colsum[m] = sum_k A[k, m]  in one pass over A (A [K, M], B [K, N], both x-contiguous): a weight
// gradient and the bias gradient that goes with it.  Returns PPK_ERR_UNSUP (nothing launched) when the shape does
// not take the fused kernel; the caller then uses ppk_gemm_f32 + ppk_colsum.
int ppk_gemm_tn_colsum_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                           float* C, int64_t ldc, float* colsum, void* workspace, size_t ws_bytes, hipStream_t stream)
{
    if (!colsum) { ppk_set_error("gemm_tn_colsum: null colsum"); return PPK_ERR_NULL; }
    if (M == 0 || N == 0) return PPK_OK;
    return ppk::gemm_csum(1, 0, M, N, K, 1.0f, A, lda, B, ldb, 0.0f, C, ldc, nullptr, colsum, workspace, ws_bytes, stream);
}

int ppk_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                 const float* A, int64_t lda, const float* B, int64_t ldb, float beta,
                 float* C, int64_t ldc, const float* bias, void* workspace, size_t ws_bytes,
                 hipStream_t stream)
{
    return ppk::gemm(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, workspace, ws_bytes, stream);
}

}  // extern "C"
