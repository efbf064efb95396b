// Lab 3 (not part of libppk.so): NT fp32-MFMA GEMM with LDS-DMA staging (global_load_lds_dwordx4),
// 3 LDS buffers, one raw barrier per K tile, counted vmcnt.  C[M,N] = A[M,K] . B[N,K]^T
//   hipcc -O3 --offload-arch=gfx950 tools/gemm_lab3.hip -o tools/gemm_lab3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else static_assert(N < 0, "add the count");
}

// BM x 64 block tile, BK = 32, 4 waves as 2 x 2, wave tile (BM/2) x 32.
template <int BM, int NBUF>
__device__ __forceinline__ void nt_glds_body(const float* __restrict__ A, long long lda, const float* __restrict__ B, long long ldb,
                                             float* __restrict__ C, long long ldc, int M, int N, int K)
{
    constexpr int BN = 64, BK = 32;
    constexpr int TM = BM / 64;                    // 32x32 accumulators per wave
    constexpr int PA = BM / 32, PB = BN / 32;      // 8-row pieces per wave (pieces w, w+4, ...)
    constexpr int PPW = PA + PB;                   // LDS-DMA instructions per wave per K tile
    constexpr int BUF = (BM + BN) * BK;            // floats per buffer
    __shared__ __align__(1024) float lds[NBUF * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // ---- staging addresses: lane -> (row r8 = lane>>3 of the piece, physical 16-B chunk lane&7)
    const int r8 = lane >> 3, pc = lane & 7;
    const float* ga[PA];
    const float* gb[PB];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int row = (wave + 4 * p) * 8 + r8;                     // row inside the A tile
        const int lc = pc ^ ((row >> 1) & 7);                        // logical chunk stored at this physical slot
        int gr = m0 + row; gr = gr < M ? gr : M - 1;
        ga[p] = A + (long long)gr * lda + lc * 4;
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int row = (wave + 4 * p) * 8 + r8;
        const int lc = pc ^ ((row >> 1) & 7);
        int gr = n0 + row; gr = gr < N ? gr : N - 1;
        gb[p] = B + (long long)gr * ldb + lc * 4;
    }
    auto stage = [&](int buf) {
        float* la = lds + buf * BUF;
        float* lb = la + BM * BK;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            __builtin_amdgcn_global_load_lds(ga[p], LDS_PTR(la + (wave + 4 * p) * 256), 16, 0, 0);
            ga[p] += BK;
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            __builtin_amdgcn_global_load_lds(gb[p], LDS_PTR(lb + (wave + 4 * p) * 256), 16, 0, 0);
            gb[p] += BK;
        }
    };

    // ---- fragment read offsets (floats): row * 32 + 4 * ((2c + h) ^ sw), sw = (li >> 1) & 7
    const int sw = (li >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xo[c] = 4 * ((2 * c + h) ^ sw);
    const int arow = (wm * (BM / 2) + li) * BK;
    const int brow = BM * BK + (wn * 32 + li) * BK;

    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int nk = K / BK;
    stage(0);
    if (nk > 1) stage(1);
    int cur = 0;
    for (int it = 0; it < nk; ++it) {
        if (it + 1 < nk) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (it + 2 < nk) { int nb = cur + 2; nb = nb >= NBUF ? nb - NBUF : nb; stage(nb); }
        const float* base = lds + cur * BUF;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 a[TM], b;
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const f32x4*)(base + arow + i * 32 * BK + xo[c]);
            b = *(const f32x4*)(base + brow + xo[c]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[s], acc[i], 0, 0, 0);
        }
        cur = cur + 1 == NBUF ? 0 : cur + 1;
    }

    const int col = n0 + wn * 32 + li;
    if (col < N) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < M) C[(long long)row * ldc + col] = acc[i][r];
            }
    }
}

#define GLDS_KERNEL(name, BM, NBUF) \
__global__ __launch_bounds__(256) void name(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, int M, int N, int K) \
{ nt_glds_body<BM, NBUF>(A, lda, B, ldb, C, ldc, M, N, K); }
GLDS_KERNEL(k_glds_64_3, 64, 3)
GLDS_KERNEL(k_glds_64_4, 64, 4)
GLDS_KERNEL(k_glds_128_3, 128, 3)

// F config: 128 x 96 block tile (4 waves along M, each 32 rows x 3 accumulators of 32 columns), BK = 32, NBUF LDS buffers.
// column map: n-tile j, accumulator g, lane li  ->  B row / C column  j * tstride + g * gstride + li
__device__ unsigned long long g_clk[4096 * 2];
template <int NBUF, int ABL = 0>
__device__ __forceinline__ void nt_f_body(const float* __restrict__ A, long long lda, const float* __restrict__ B, long long ldb,
                                          float* __restrict__ C, long long ldc, int M, int N, int K, int tstride, int gstride)
{
    constexpr int BM = 128, BNR = 96, BK = 32;
    constexpr int PA = 4, PB = 3;                  // 8-row pieces per wave: A 16 pieces, B 12 pieces
    constexpr int PPW = PA + PB;
    constexpr int BUF = (BM + BNR) * BK;
    __shared__ __align__(1024) float lds[NBUF * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, j = blockIdx.x;
    const int r8 = lane >> 3, pc = lane & 7;
    const float* ga[PA];
    const float* gb[PB];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int row = (wave + 4 * p) * 8 + r8;
        const int lc = pc ^ ((row >> 1) & 7);
        int gr = m0 + row; gr = gr < M ? gr : M - 1;
        ga[p] = A + (long long)gr * lda + lc * 4;
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int row = (wave + 4 * p) * 8 + r8;     // 0..95 inside the B tile: accumulator g = row / 32
        const int lc = pc ^ ((row >> 1) & 7);
        int gr = j * tstride + (row >> 5) * gstride + (row & 31); gr = gr < N ? gr : N - 1;
        gb[p] = B + (long long)gr * ldb + lc * 4;
    }
    auto stage = [&](int buf) {
        float* la = lds + buf * BUF;
        float* lb = la + BM * BK;
#pragma unroll
        for (int p = 0; p < PA; ++p) { __builtin_amdgcn_global_load_lds(ga[p], LDS_PTR(la + (wave + 4 * p) * 256), 16, 0, 0); ga[p] += BK; }
#pragma unroll
        for (int p = 0; p < PB; ++p) { __builtin_amdgcn_global_load_lds(gb[p], LDS_PTR(lb + (wave + 4 * p) * 256), 16, 0, 0); gb[p] += BK; }
    };
    const int sw = (li >> 1) & 7;
    int xo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xo[c] = 4 * ((2 * c + h) ^ sw);
    const int arow = (wave * 32 + li) * BK;
    const int brow = BM * BK + li * BK;
    f32x16 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    f32x4 a, b[3];
    const int nk = K / BK;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    stage(0);
    if (NBUF > 2 && nk > 1) stage(1);
    int cur = 0;
    for (int it = 0; it < nk; ++it) {
        if (!(ABL & 2) || it == 0) { if (NBUF > 2 && it + 1 < nk) wait_vmcnt<PPW>(); else wait_vmcnt<0>(); }
        if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
        if (!(ABL & 2) && it + NBUF - 1 < nk) { int nb = cur + NBUF - 1; nb = nb >= NBUF ? nb - NBUF : nb; stage(nb); }
        const float* base = lds + cur * BUF;
        if (ABL & 8) {
            f32x4 a0, a1, b0[3], b1[3];
            a0 = *(const f32x4*)(base + arow + xo[0]);
#pragma unroll
            for (int g = 0; g < 3; ++g) b0[g] = *(const f32x4*)(base + brow + g * 32 * BK + xo[0]);
            __builtin_amdgcn_sched_barrier(0);
            // chunk 0 (regs 0), prefetch chunk 1 into regs 1
            a1 = *(const f32x4*)(base + arow + xo[1]);
#pragma unroll
            for (int g = 0; g < 3; ++g) b1[g] = *(const f32x4*)(base + brow + g * 32 * BK + xo[1]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[g][s], acc[g], 0, 0, 0);
            for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_barrier(0);
            a0 = *(const f32x4*)(base + arow + xo[2]);
#pragma unroll
            for (int g = 0; g < 3; ++g) b0[g] = *(const f32x4*)(base + brow + g * 32 * BK + xo[2]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[g][s], acc[g], 0, 0, 0);
            for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_barrier(0);
            a1 = *(const f32x4*)(base + arow + xo[3]);
#pragma unroll
            for (int g = 0; g < 3; ++g) b1[g] = *(const f32x4*)(base + brow + g * 32 * BK + xo[3]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[g][s], acc[g], 0, 0, 0);
            for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[g][s], acc[g], 0, 0, 0);
        } else {
        if (!(ABL & 1) || it == 0) {
            a = *(const f32x4*)(base + arow + xo[0]);
#pragma unroll
            for (int g = 0; g < 3; ++g) b[g] = *(const f32x4*)(base + brow + g * 32 * BK + xo[0]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 an, bn[3];
            if (c < 3 && !(ABL & 1)) {
                an = *(const f32x4*)(base + arow + xo[c + 1]);
#pragma unroll
                for (int g = 0; g < 3; ++g) bn[g] = *(const f32x4*)(base + brow + g * 32 * BK + xo[c + 1]);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[g][s], acc[g], 0, 0, 0);
            if (c < 3 && !(ABL & 1)) { a = an; b[0] = bn[0]; b[1] = bn[1]; b[2] = bn[2]; }
        }
        }
        cur = cur + 1 == NBUF ? 0 : cur + 1;
    }
    if (tid == 0) {
        const int bid = blockIdx.y * gridDim.x + blockIdx.x;
        if (bid < 4096) { g_clk[2 * bid] = __builtin_amdgcn_s_memtime() - t0; g_clk[2 * bid + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const int col = j * tstride + g * gstride + li;
        if (col >= N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < M) C[(long long)row * ldc + col] = acc[g][r];
        }
    }
}
#define F_KERNEL(name, NBUF) \
__global__ __launch_bounds__(256) void name(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, int M, int N, int K, int ts, int gs) \
{ nt_f_body<NBUF>(A, lda, B, ldb, C, ldc, M, N, K, ts, gs); }
F_KERNEL(k_f_2, 2)
F_KERNEL(k_f_3, 3)
#define FA_KERNEL(name, ABL) \
__global__ __launch_bounds__(256) void name(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, int M, int N, int K, int ts, int gs) \
{ nt_f_body<2, ABL>(A, lda, B, ldb, C, ldc, M, N, K, ts, gs); }
FA_KERNEL(k_fa_1, 1)
FA_KERNEL(k_fa_2, 2)
FA_KERNEL(k_fa_3, 3)
FA_KERNEL(k_fa_7, 7)
FA_KERNEL(k_fa_8, 8)

// staging only: the LDS-DMA fill stream of the 64x64 kernel with nothing consuming it (per-CU fill rate)
template <int NBUF>
__device__ __forceinline__ void stage_only_body(const float* __restrict__ A, long long lda, const float* __restrict__ B, long long ldb,
                                                float* __restrict__ C, int M, int N, int K, int aux_nt)
{
    constexpr int BM = 64, BN = 64, BK = 32, BUF = (BM + BN) * BK, PPW = 4;
    __shared__ __align__(1024) float lds[NBUF * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int r8 = lane >> 3, pc = lane & 7;
    const float* ga[2];
    const float* gb[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = (wave + 4 * p) * 8 + r8;
        const int lc = pc ^ ((row >> 1) & 7);
        int gr = m0 + row; gr = gr < M ? gr : M - 1;
        ga[p] = A + (long long)gr * lda + lc * 4;
        int gc = n0 + row; gc = gc < N ? gc : N - 1;
        gb[p] = B + (long long)gc * ldb + lc * 4;
    }
    const int nk = K / BK;
    int cur = 0;
    for (int it = 0; it < nk; ++it) {
        float* la = lds + cur * BUF;
        float* lb = la + BM * BK;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (aux_nt) __builtin_amdgcn_global_load_lds(ga[p], LDS_PTR(la + (wave + 4 * p) * 256), 16, 0, 2);
            else __builtin_amdgcn_global_load_lds(ga[p], LDS_PTR(la + (wave + 4 * p) * 256), 16, 0, 0);
            ga[p] += BK;
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (aux_nt) __builtin_amdgcn_global_load_lds(gb[p], LDS_PTR(lb + (wave + 4 * p) * 256), 16, 0, 2);
            else __builtin_amdgcn_global_load_lds(gb[p], LDS_PTR(lb + (wave + 4 * p) * 256), 16, 0, 0);
            gb[p] += BK;
        }
        if (it >= NBUF - 1) wait_vmcnt<(NBUF - 1) * PPW>();
        cur = cur + 1 == NBUF ? 0 : cur + 1;
    }
    wait_vmcnt<0>();
    __syncthreads();
    if (tid == 0) C[(long long)blockIdx.y * gridDim.x + blockIdx.x] = lds[17];
}
__global__ __launch_bounds__(256) void k_stage4(const float* A, long long lda, const float* B, long long ldb, float* C, int M, int N, int K, int nt)
{ stage_only_body<4>(A, lda, B, ldb, C, M, N, K, nt); }
__global__ __launch_bounds__(256) void k_stage2(const float* A, long long lda, const float* B, long long ldb, float* C, int M, int N, int K, int nt)
{ stage_only_body<2>(A, lda, B, ldb, C, M, N, K, nt); }

// the baseline from the library (register staged, guarded) for comparison
__global__ __launch_bounds__(256) void k_base(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K)
{
    constexpr int BM = 64, BN = 64, BK = 32, STR = 36;
    __shared__ __align__(16) float lds[(BM + BN) * STR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 ra[2], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i, row = idx >> 3, kq = (idx & 7) << 2;
            ra[i] = (m0 + row < M) ? *(const f32x4*)(A + (long long)(m0 + row) * K + k0 + kq) : f32x4{0, 0, 0, 0};
            rb[i] = (n0 + row < N) ? *(const f32x4*)(B + (long long)(n0 + row) * K + k0 + kq) : f32x4{0, 0, 0, 0};
        }
    };
    float* la = lds; float* lb = lds + BM * STR;
    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i, row = idx >> 3, kq = (idx & 7) << 2;
            *(f32x4*)(la + row * STR + kq) = ra[i];
            *(f32x4*)(lb + row * STR + kq) = rb[i];
        }
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 a = *(const f32x4*)(la + (wm * 32 + li) * STR + c * 8 + 4 * h), b = *(const f32x4*)(lb + (wn * 32 + li) * STR + c * 8 + 4 * h);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wn * 32 + li;
    if (col < N)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < M) C[(long long)row * N + col] = acc[r];
        }
}

template <int BM, int NBUF>
void launch_glds(dim3 g, const float* A, const float* B, float* C, int M, int N, int K)
{
    if (BM == 64 && NBUF == 3) k_glds_64_3<<<g, 256>>>(A, K, B, K, C, N, M, N, K);
    else if (BM == 64) k_glds_64_4<<<g, 256>>>(A, K, B, K, C, N, M, N, K);
    else k_glds_128_3<<<g, 256>>>(A, K, B, K, C, N, M, N, K);
}
void launch_base(dim3 g, const float* A, const float* B, float* C, int M, int N, int K) { k_base<<<g, 256>>>(A, B, C, M, N, K); }
static std::vector<float> hA, hB;
static double check(const float* C, int M, int N, int K)
{
    std::vector<float> out((size_t)M * N);
    CK(hipMemcpy(out.data(), C, out.size() * 4, hipMemcpyDeviceToHost));
    double e = 0;
    for (int t = 0; t < 400; ++t) {
        int i = t < 8 ? M - 1 - t % M : rand() % M, j = t < 8 ? N - 1 : rand() % N;
        if (i < 0) i = 0;
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += (double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k];
        e = fmax(e, fabs(acc - out[(size_t)i * N + j]));
    }
    return e;
}

template <typename F>
static void bench(const char* name, F launch, float* C, int M, int N, int K, int blocks)
{
    CK(hipMemset(C, 0xff, (size_t)M * N * 4));
    launch();
    CK(hipDeviceSynchronize());
    const double err = check(C, M, N, K);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9, tot = 0;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        best = ms < best ? ms : best; tot += ms;
    }
    printf("  %-26s %7.1f us (min %6.1f) %6.1f TF  blocks %5d  maxerr %.2e\n", name, tot / 5 * 1e3, best * 1e3, 2.0 * M * N * K / (tot / 5) / 1e9, blocks, err);
}

int main()
{
    const int shapes[][3] = {{2048, 1536, 512}, {2048, 512, 1536}, {64, 512, 1536}, {64, 64, 512}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        hA.resize((size_t)M * K); hB.resize((size_t)N * K);
        for (auto& v : hA) v = (float)rand() / RAND_MAX - 0.5f;
        for (auto& v : hB) v = (float)rand() / RAND_MAX - 0.5f;
        float *A, *B, *C;
        CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
        CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
        printf("M=%d N=%d K=%d\n", M, N, K);
        const dim3 g64((N + 63) / 64, (M + 63) / 64), g128((N + 63) / 64, (M + 127) / 128);
        bench("base 64x64 regstage", [&] { launch_base(g64, A, B, C, M, N, K); }, C, M, N, K, g64.x * g64.y);
        bench("glds 64x64 3buf", [&] { launch_glds<64, 3>(g64, A, B, C, M, N, K); }, C, M, N, K, g64.x * g64.y);
        bench("glds 64x64 4buf", [&] { launch_glds<64, 4>(g64, A, B, C, M, N, K); }, C, M, N, K, g64.x * g64.y);
        bench("glds 128x64 3buf", [&] { launch_glds<128, 3>(g128, A, B, C, M, N, K); }, C, M, N, K, g128.x * g128.y);
        {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int v = 0; v < 4; ++v) {
                float ms;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < 20; ++i) {
                        if (v < 2) k_stage4<<<g64, 256>>>(A, K, B, K, C, M, N, K, v & 1);
                        else k_stage2<<<g64, 256>>>(A, K, B, K, C, M, N, K, v & 1);
                    }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                }
                const double bytes = (double)g64.x * g64.y * (64 + 64) * K * 4.0;
                printf("  stage-only %s %s   %7.1f us  %.2f TB/s  %.1f GB/s per CU (blocks %d)\n", v < 2 ? "4buf" : "2buf", (v & 1) ? "nt " : "def", ms / 20 * 1e3,
                       bytes / (ms / 20 * 1e-3) / 1e12, bytes / (ms / 20 * 1e-3) / 1e9 / (g64.x * g64.y < 256 ? g64.x * g64.y : 256), g64.x * g64.y);
            }
        }
        const dim3 gf((N + 95) / 96, (M + 127) / 128);
        bench("F 128x96 2buf plain", [&] { k_f_2<<<gf, 256>>>(A, K, B, K, C, N, M, N, K, 96, 32); }, C, M, N, K, gf.x * gf.y);
        bench("F 128x96 3buf plain", [&] { k_f_3<<<gf, 256>>>(A, K, B, K, C, N, M, N, K, 96, 32); }, C, M, N, K, gf.x * gf.y);
        auto census = [&](const char* nm) {
            const int nb = gf.x * gf.y < 4096 ? gf.x * gf.y : 4096;
            std::vector<unsigned long long> st(2 * nb);
            CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_clk), st.size() * 8));
            std::vector<double> clk, cyc;
            for (int b = 0; b < nb; ++b) { clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 100.0); cyc.push_back((double)st[2 * b]); }
            std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
            printf("    %s main loop: median clock %.0f MHz, median cycles %.0f (%.1f per MFMA of one wave), max cycles %.0f\n", nm, clk[nb / 2], cyc[nb / 2], cyc[nb / 2] / (3.0 * K / 2), cyc[nb - 1]);
        };
        census("3buf");
        bench("F 2buf again", [&] { k_f_2<<<gf, 256>>>(A, K, B, K, C, N, M, N, K, 96, 32); }, C, M, N, K, gf.x * gf.y); census("2buf");
        bench("F 2buf pinned reads", [&] { k_fa_8<<<gf, 256>>>(A, K, B, K, C, N, M, N, K, 96, 32); }, C, M, N, K, gf.x * gf.y); census("pinned");
        bench("F abl no-ds_read", [&] { k_fa_1<<<gf, 256>>>(A, K, B, K, C, N, M, N, K, 96, 32); }, C, M, N, K, gf.x * gf.y); census("no-ds_read");
        bench("F abl no-stage", [&] { k_fa_2<<<gf, 256>>>(A, K, B, K, C, N, M, N, K, 96, 32); }, C, M, N, K, gf.x * gf.y); census("no-stage");
        bench("F abl no-read no-stage", [&] { k_fa_3<<<gf, 256>>>(A, K, B, K, C, N, M, N, K, 96, 32); }, C, M, N, K, gf.x * gf.y); census("no-read-no-stage");
        bench("F abl mfma only", [&] { k_fa_7<<<gf, 256>>>(A, K, B, K, C, N, M, N, K, 96, 32); }, C, M, N, K, gf.x * gf.y); census("mfma-only");
        if (N % 96 == 0) bench("F 128x96 2buf gate-layout", [&] { k_f_2<<<gf, 256>>>(A, K, B, K, C, N, M, N, K, 32, N / 3); }, C, M, N, K, gf.x * gf.y);
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
    }
    return 0;
}
