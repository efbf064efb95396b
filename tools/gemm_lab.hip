// Tuning lab (not part of libppk.so): NT fp32-MFMA GEMM variants at the GRU step shapes.
//   hipcc -O3 --offload-arch=gfx950 tools/gemm_lab.hip -o tools/gemm_lab && ./tools/gemm_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ unsigned long long g_stamps[4 * 8192];

template <int BM, int BN, int WM, int WN, int BK, bool DB>
__global__ __launch_bounds__(64 * WM * WN) void k_nt(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                     int M, int N, int K)
{
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int STR = BK + 4;
    constexpr int QK = BK / 4;                        // float4 per row
    constexpr int RA = BM * QK / NT, RB = BN * QK / NT;
    __shared__ __align__(16) float lds[(DB ? 2 : 1) * (BM + BN) * STR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const unsigned long long st0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[RA], rb[RB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int idx = tid + NT * i, row = idx / QK, kq = (idx % QK) * 4;
            ra[i] = (m0 + row < M) ? *(const f32x4*)(A + (long long)(m0 + row) * K + k0 + kq) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int idx = tid + NT * i, row = idx / QK, kq = (idx % QK) * 4;
            rb[i] = (n0 + row < N) ? *(const f32x4*)(B + (long long)(n0 + row) * K + k0 + kq) : f32x4{0, 0, 0, 0};
        }
    };
    auto sstore = [&](float* la, float* lb) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int idx = tid + NT * i, row = idx / QK, kq = (idx % QK) * 4;
            *(f32x4*)(la + row * STR + kq) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int idx = tid + NT * i, row = idx / QK, kq = (idx % QK) * 4;
            *(f32x4*)(lb + row * STR + kq) = rb[i];
        }
    };
    auto compute = [&](const float* la, const float* lb) {
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const f32x4*)(la + (wm * (BM / WM) + i * 32 + li) * STR + c * 8 + 4 * h);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(const f32x4*)(lb + (wn * (BN / WN) + j * 32 + li) * STR + c * 8 + 4 * h);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (!DB) {
        float* la = lds; float* lb = lds + BM * STR;
        gload(0);
        for (int k0 = 0; k0 < K; k0 += BK) {
            sstore(la, lb);
            __syncthreads();
            if (k0 + BK < K) gload(k0 + BK);
            compute(la, lb);
            __syncthreads();
        }
    } else {
        constexpr int BUF = (BM + BN) * STR;
        gload(0);
        sstore(lds, lds + BM * STR);
        __syncthreads();
        if (BK < K) gload(BK);
        int cur = 0;
        for (int k0 = 0; k0 < K; k0 += BK) {
            float* la = lds + cur * BUF;
            compute(la, la + BM * STR);
            if (k0 + BK < K) {
                float* na = lds + (cur ^ 1) * BUF;
                sstore(na, na + BM * STR);
            }
            __syncthreads();
            if (k0 + 2 * BK < K) gload(k0 + 2 * BK);
            cur ^= 1;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / WN) + j * 32 + li;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < M) C[(long long)row * N + col] = acc[i][j][r];
            }
        }
    if (tid == 0) {
        const int bid = blockIdx.y * gridDim.x + blockIdx.x;
        if (bid < 8192) {
            g_stamps[4 * bid] = st0;
            g_stamps[4 * bid + 1] = __builtin_amdgcn_s_memrealtime();
            g_stamps[4 * bid + 2] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
            g_stamps[4 * bid + 3] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
        }
    }
}

template <int BM, int BN, int WM, int WN, int BK, bool DB>
__global__ __launch_bounds__(64 * WM * WN) void k_nt16(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                       int M, int N, int K)
{
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int STR = BK + 4;
    constexpr int QK = BK / 4;
    constexpr int RA = BM * QK / NT, RB = BN * QK / NT;
    __shared__ __align__(16) float lds[(DB ? 2 : 1) * (BM + BN) * STR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const unsigned long long st0 = __builtin_amdgcn_s_memrealtime();
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    f32x4 ra[RA], rb[RB];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int idx = tid + NT * i, row = idx / QK, kq = (idx % QK) * 4;
            ra[i] = (m0 + row < M) ? *(const f32x4*)(A + (long long)(m0 + row) * K + k0 + kq) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int idx = tid + NT * i, row = idx / QK, kq = (idx % QK) * 4;
            rb[i] = (n0 + row < N) ? *(const f32x4*)(B + (long long)(n0 + row) * K + k0 + kq) : f32x4{0, 0, 0, 0};
        }
    };
    auto sstore = [&](float* la, float* lb) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int idx = tid + NT * i, row = idx / QK, kq = (idx % QK) * 4;
            *(f32x4*)(la + row * STR + kq) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int idx = tid + NT * i, row = idx / QK, kq = (idx % QK) * 4;
            *(f32x4*)(lb + row * STR + kq) = rb[i];
        }
    };
    auto compute = [&](const float* la, const float* lb) {
#pragma unroll
        for (int c = 0; c < BK / 16; ++c) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const f32x4*)(la + (wm * (BM / WM) + i * 16 + li) * STR + c * 16 + 4 * q);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(const f32x4*)(lb + (wn * (BN / WN) + j * 16 + li) * STR + c * 16 + 4 * q);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (!DB) {
        float* la = lds; float* lb = lds + BM * STR;
        gload(0);
        for (int k0 = 0; k0 < K; k0 += BK) {
            sstore(la, lb);
            __syncthreads();
            if (k0 + BK < K) gload(k0 + BK);
            compute(la, lb);
            __syncthreads();
        }
    } else {
        constexpr int BUF = (BM + BN) * STR;
        gload(0);
        sstore(lds, lds + BM * STR);
        __syncthreads();
        if (BK < K) gload(BK);
        int cur = 0;
        for (int k0 = 0; k0 < K; k0 += BK) {
            float* la = lds + cur * BUF;
            compute(la, la + BM * STR);
            if (k0 + BK < K) {
                float* na = lds + (cur ^ 1) * BUF;
                sstore(na, na + BM * STR);
            }
            __syncthreads();
            if (k0 + 2 * BK < K) gload(k0 + 2 * BK);
            cur ^= 1;
        }
    }
    // C/D map 16x16: col = lane & 15, row = 4 * (lane >> 4) + r
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / WN) + j * 16 + li;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * (BM / WM) + i * 16 + 4 * q + r;
                if (row < M) C[(long long)row * N + col] = acc[i][j][r];
            }
        }
    if (tid == 0) {
        const int bid = blockIdx.y * gridDim.x + blockIdx.x;
        if (bid < 8192) {
            g_stamps[4 * bid] = st0;
            g_stamps[4 * bid + 1] = __builtin_amdgcn_s_memrealtime();
            g_stamps[4 * bid + 2] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
            g_stamps[4 * bid + 3] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
        }
    }
}

template <int BM, int BN, int WM, int WN, int BK, bool DB, bool M16 = false>
float run(const char* name, const float* A, const float* B, float* C, int M, int N, int K, const std::vector<float>& ref, int pad = 0)
{
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM), block(64 * WM * WN);
    for (int i = 0; i < 3; ++i) { if constexpr (M16) k_nt16<BM, BN, WM, WN, BK, DB><<<grid, block, pad>>>(A, B, C, M, N, K); else k_nt<BM, BN, WM, WN, BK, DB><<<grid, block, pad>>>(A, B, C, M, N, K); }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 40;
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) { if constexpr (M16) k_nt16<BM, BN, WM, WN, BK, DB><<<grid, block, pad>>>(A, B, C, M, N, K); else k_nt<BM, BN, WM, WN, BK, DB><<<grid, block, pad>>>(A, B, C, M, N, K); }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    std::vector<float> out((size_t)M * N);
    CK(hipMemcpy(out.data(), C, sizeof(float) * out.size(), hipMemcpyDeviceToHost));
    double err = 0;
    for (size_t i = 0; i < ref.size(); i += 97) err = fmax(err, fabs((double)out[i] - ref[i]));
    if (getenv("CENSUS")) {
        const int nb = grid.x * grid.y < 8192 ? grid.x * grid.y : 8192;
        std::vector<unsigned long long> st(4 * nb);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8));
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int b = 0; b < nb; ++b) { tmin = st[4 * b] < tmin ? st[4 * b] : tmin; tmax = st[4 * b + 1] > tmax ? st[4 * b + 1] : tmax; }
        std::map<unsigned long long, std::vector<int>> percu;
        double dsum = 0;
        for (int b = 0; b < nb; ++b) {
            unsigned long long hw = st[4 * b + 2], xcc = st[4 * b + 3] & 0xf;
            unsigned long long key = (xcc << 32) | (hw & 0xff00);   // cu_id, sh_id, se_id bits
            percu[key].push_back(b);
            dsum += (double)(st[4 * b + 1] - st[4 * b]) * 0.01;
        }
        int hist[16] = {0};
        for (auto& kv : percu) hist[kv.second.size() < 15 ? kv.second.size() : 15]++;
        printf("    census: span %.2f us, mean block %.2f us, CUs used %zu, blocks/CU histogram:", (double)(tmax - tmin) * 0.01, dsum / nb, percu.size());
        for (int i = 0; i < 16; ++i) if (hist[i]) printf(" %dx%d", i, hist[i]);
        printf("\n");
        // first CU timeline
        int shown = 0;
        for (auto& kv : percu) { if (shown++ >= 2) break; printf("      cu %llx:", kv.first); for (int b : kv.second) printf(" [%0.2f-%0.2f]", (double)(st[4 * b] - tmin) * 0.01, (double)(st[4 * b + 1] - tmin) * 0.01); printf("\n"); }
    }
    printf("  %-28s %7.1f us %6.1f TF  (blocks %d, maxerr %.2e)\n", name, ms * 1e3, 2.0 * M * N * K / ms / 1e9, grid.x * grid.y, err);
    return ms;
}

int main()
{
    const int shapes[][3] = {{2048, 1536, 512}, {2048, 3072, 512}, {2048, 512, 1536}, {1216, 1536, 512}, {512, 1536, 512}, {256, 1536, 512}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
        for (auto& v : hA) v = (float)rand() / RAND_MAX - 0.5f;
        for (auto& v : hB) v = (float)rand() / RAND_MAX - 0.5f;
        float *A, *B, *C;
        CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
        CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
        // reference from the baseline variant
        k_nt<64, 64, 2, 2, 32, false><<<dim3((N + 63) / 64, (M + 63) / 64), 256>>>(A, B, C, M, N, K);
        CK(hipDeviceSynchronize());
        std::vector<float> ref((size_t)M * N);
        CK(hipMemcpy(ref.data(), C, ref.size() * 4, hipMemcpyDeviceToHost));
        // spot-check the reference itself on the host
        double e = 0;
        for (int t = 0; t < 50; ++t) {
            int i = rand() % M, j = rand() % N;
            double acc = 0;
            for (int k = 0; k < K; ++k) acc += (double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k];
            e = fmax(e, fabs(acc - ref[(size_t)i * N + j]));
        }
        printf("M=%d N=%d K=%d (host check err %.2e)\n", M, N, K, e);
        run<64, 64, 2, 2, 32, false>("64x64 4w bk32", A, B, C, M, N, K, ref);
        run<64, 64, 2, 2, 32, false, true>("m16 64x64 4w bk32", A, B, C, M, N, K, ref);
        run<64, 64, 2, 2, 32, true, true>("m16 64x64 4w bk32 db", A, B, C, M, N, K, ref);
        run<64, 64, 2, 2, 64, false, true>("m16 64x64 4w bk64", A, B, C, M, N, K, ref);
        run<128, 64, 2, 2, 32, false, true>("m16 128x64 4w bk32", A, B, C, M, N, K, ref);
        run<128, 64, 2, 2, 32, true, true>("m16 128x64 4w bk32 db", A, B, C, M, N, K, ref);
        run<128, 128, 2, 2, 32, false, true>("m16 128x128 4w bk32", A, B, C, M, N, K, ref);
        run<128, 128, 2, 2, 32, true, true>("m16 128x128 4w bk32 db", A, B, C, M, N, K, ref);
        run<128, 128, 4, 2, 32, true, true>("m16 128x128 8w bk32 db", A, B, C, M, N, K, ref);
        run<64, 32, 2, 2, 32, false, true>("m16 64x32 4w bk32", A, B, C, M, N, K, ref);
        run<32, 64, 2, 2, 32, false, true>("m16 32x64 4w bk32", A, B, C, M, N, K, ref);
        run<32, 32, 2, 2, 32, false, true>("m16 32x32 4w bk32", A, B, C, M, N, K, ref);
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
    }
    return 0;
}
