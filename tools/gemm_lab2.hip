// Lab 2 (not part of libppk.so): where do the cycles of one K-iteration of the NT fp32-MFMA tile loop go?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ unsigned long long g_st[64 * 16];
#define STAMP(i) do { if (STAMPS && it == WHICH) { st##i = __builtin_amdgcn_s_memtime(); } } while (0)

template <bool STAMPS, int WHICH>
__global__ __launch_bounds__(256) void k_nt(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K, int ld, int rot, int nomfma)
{
    constexpr int BM = 64, BN = 64, BK = 32, STR = 36;
    __shared__ __align__(16) float lds[(BM + BN) * STR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0, st6 = 0, st7 = 0;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 ra[2], rb[2];
    const float* pa = A + (long long)(m0 + (tid >> 3)) * ld + ((tid & 7) << 2);
    const float* pb = B + (long long)(n0 + (tid >> 3)) * ld + ((tid & 7) << 2);
    auto gload = [&](int k0) {
        ra[0] = *(const f32x4*)(pa + k0); ra[1] = *(const f32x4*)(pa + k0 + 32LL * ld);
        rb[0] = *(const f32x4*)(pb + k0); rb[1] = *(const f32x4*)(pb + k0 + 32LL * ld);
    };
    float* la = lds; float* lb = lds + BM * STR;
    float* wa = la + (tid >> 3) * STR + ((tid & 7) << 2);
    float* wb = lb + (tid >> 3) * STR + ((tid & 7) << 2);
    const float* fa = la + (wm * 32 + li) * STR + 4 * h;
    const float* fb = lb + (wn * 32 + li) * STR + 4 * h;
    const int kr = rot ? ((blockIdx.x + blockIdx.y) * rot * BK) % K : 0;
    gload(kr);
    int it = 0;
    for (int k0 = 0; k0 < K; k0 += BK, ++it) {
        STAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(6);
        *(f32x4*)wa = ra[0]; *(f32x4*)(wa + 32 * STR) = ra[1];
        *(f32x4*)wb = rb[0]; *(f32x4*)(wb + 32 * STR) = rb[1];
        STAMP(1);
        __syncthreads();
        STAMP(2);
        if (k0 + BK < K) gload((kr + k0 + BK) % K);
        STAMP(3);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 a = *(const f32x4*)(fa + c * 8), b = *(const f32x4*)(fb + c * 8);
            if (nomfma) { acc[c] += a[0] * b[1] + a[2] * b[3]; continue; }
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        }
        STAMP(4);
        __syncthreads();
        STAMP(5);
        if (STAMPS && it == WHICH + 1) st7 = __builtin_amdgcn_s_memtime();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, col = n0 + wn * 32 + li;
        C[(long long)row * N + col] = acc[r];
    }
    if (STAMPS && tid == 0) {
        unsigned long long* o = g_st + 8 * (blockIdx.y * gridDim.x + blockIdx.x < 64 ? blockIdx.y * gridDim.x + blockIdx.x : 63);
        o[0] = st0; o[1] = st6; o[2] = st1; o[3] = st2; o[4] = st3; o[5] = st4; o[6] = st5; o[7] = st7;
    }
}

int main()
{
    const int M = 2048, N = 1536, K = 512;
    const int LDP = K + 32;
    std::vector<float> hA((size_t)M * LDP), hB((size_t)N * LDP);
    for (auto& v : hA) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : hB) v = (float)rand() / RAND_MAX - 0.5f;
    float *A, *B, *C;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    for (int full = 0; full < 2; ++full) {
        dim3 grid(full ? N / 64 : 1, full ? M / 64 : 1);
        for (int i = 0; i < 5; ++i) k_nt<true, 8><<<grid, 256>>>(A, B, C, M, N, K, K, 0, 0);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> st(64 * 8);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_st), st.size() * 8));
        printf("%s grid, iteration 8: cycles since top: vmcnt0, lds-written, barrier1, gload-issued, mfma-issued, barrier2 | next top\n", full ? "full" : "single-block");
        for (int b = 0; b < (full ? 6 : 1); ++b) {
            printf("  block %2d:", b);
            for (int j = 1; j < 8; ++j) printf(" %6llu", st[b * 8 + j] - st[b * 8]);
            printf("\n");
        }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        const int lds_[4] = {K, K, LDP, K}, rots[4] = {0, 0, 0, 1};
        for (int v = 0; v < 4; ++v) {
            const int nomfma = v >= 1;
            k_nt<false, 0><<<grid, 256>>>(A, B, C, M, N, K, lds_[v], rots[v], nomfma);
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) k_nt<false, 0><<<grid, 256>>>(A, B, C, M, N, K, lds_[v], rots[v], nomfma);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  unstamped kernel ld=%d rot=%d nomfma=%d: %.1f us  %.1f TF\n", lds_[v], rots[v], nomfma, ms * 1e3 / 20, 2.0 * grid.x * 64 * grid.y * 64 * K / (ms / 20) / 1e9);
        }
    }
    return 0;
}
