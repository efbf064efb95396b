// Lab: what fp32-MFMA rate and clock does the chip sustain? (not part of libppk.so)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NACC, bool SHAPE16>
__global__ __launch_bounds__(256) void k_peak(const float* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* stamps)
{
    const int tid = threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(tid * 8 + i) & 4095]; b[i] = in[(tid * 8 + i + 2048) & 4095]; }
    f32x16 acc[NACC];
    f32x4 acc4[NACC];
    for (int n = 0; n < NACC; ++n) { for (int r = 0; r < 16; ++r) acc[n][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[n][r] = 0.f; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int n = 0; n < NACC; ++n) {
                if constexpr (SHAPE16) acc4[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[(s + n) & 7], acc4[n], 0, 0, 0);
                else acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[(s + n) & 7], acc[n], 0, 0, 0);
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int n = 0; n < NACC; ++n) { for (int r = 0; r < 16; ++r) s += acc[n][r]; for (int r = 0; r < 4; ++r) s += acc4[n][r]; }
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC, bool SHAPE16>
void run(const char* name, int blocks, const float* in, float* out, unsigned long long* stamps, bool zero, int pad = 0)
{
    const int iters = 4000;
    for (int i = 0; i < 2; ++i) k_peak<NACC, SHAPE16><<<blocks, 256, pad>>>(in, out, iters, stamps);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) k_peak<NACC, SHAPE16><<<blocks, 256, pad>>>(in, out, iters, stamps);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    std::vector<unsigned long long> st(2 * blocks);
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (int b = 0; b < blocks; ++b) clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 100.0);
    std::sort(clk.begin(), clk.end());
    const double flop_per = SHAPE16 ? 2.0 * 16 * 16 * 4 : 2.0 * 32 * 32 * 2;
    const double flops = (double)blocks * 4 * iters * 8 * NACC * flop_per;
    const double cyc_per_mfma = (double)st[0] / ((double)iters * 8 * NACC);
    printf("%-34s blocks %4d %s: %8.1f us  %6.1f TF  clock(median) %.0f MHz  cycles/MFMA/wave %.1f\n", name, blocks, zero ? "zeros " : "random", ms * 1e3,
           flops / ms / 1e9, clk[clk.size() / 2], cyc_per_mfma);
}

int main()
{
    float *in, *out; unsigned long long* stamps;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 4096 * 256 * 4)); CK(hipMalloc(&stamps, 4096 * 16));
    std::vector<float> h(4096);
    for (int z = 0; z < 1; ++z) {
        for (auto& v : h) v = z ? 0.f : (float)rand() / RAND_MAX - 0.5f;
        CK(hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
        run<1, false>("32x32x2 1 acc, 1 wave/SIMD", 256, in, out, stamps, z);
        run<4, false>("32x32x2 4 acc, 1 wave/SIMD", 256, in, out, stamps, z);
        run<1, false>("32x32x2 1 acc, 2 wave/SIMD", 512, in, out, stamps, z);
        run<1, false>("32x32x2 1 acc, 3 wave/SIMD", 768, in, out, stamps, z);
        run<1, false>("32x32x2 1 acc, 3 wave/SIMD capped", 768, in, out, stamps, z, 50 * 1024);
        run<2, false>("32x32x2 2 acc, 3 wave/SIMD capped", 768, in, out, stamps, z, 50 * 1024);
        run<1, false>("32x32x2 1 acc, 4 wave/SIMD capped", 1024, in, out, stamps, z, 36 * 1024);
        run<2, false>("32x32x2 2 acc, 2 wave/SIMD capped", 512, in, out, stamps, z, 70 * 1024);
        run<4, true>("16x16x4 4 acc, 1 wave/SIMD", 256, in, out, stamps, z);
        run<1, true>("16x16x4 1 acc, 3 wave/SIMD capped", 768, in, out, stamps, z, 50 * 1024);
        run<2, true>("16x16x4 2 acc, 3 wave/SIMD capped", 768, in, out, stamps, z, 50 * 1024);
        run<4, true>("16x16x4 4 acc, 2 wave/SIMD", 512, in, out, stamps, z);
    }
    return 0;
}
