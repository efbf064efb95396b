// How fast can a CU fill LDS from L2-resident data while every CU does the same?  LDS-DMA (global_load_lds, what the
// persistent sequence kernels' K loops use) against register staging (global_load_dwordx4 -> ds_write_b128).
//   hipcc -O3 --offload-arch=gfx950 tools/fill_lab.hip -o tools/fill_lab && tools/fill_lab
// Each workgroup (256 threads) streams `stages` stages of 16 KB (four blocks of 32 rows x 128 B, rows `pitch` bytes
// apart: the K-loop access pattern) out of a 640 KB slice it shares with the workgroups of the same n-tile, keeps
// `depth` stages in flight, and reads every staged byte back from LDS once (ds_read_b128, like the MFMA fragments).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef float f32x16 __attribute__((ext_vector_type(16)));
// the K loop of the sequence kernels in miniature: block 0 of a stage optionally comes from a cold buffer (the activation
// rows: first touch, beyond L2), NM MFMAs (32x32x2 fp32) per wave and stage on the fragments read from LDS
// k_loop with the stage's four loads issued BETWEEN its MFMAs (one after every NM / 4) instead of in a burst behind the barrier
template <int DEPTH, int NM>
__global__ __launch_bounds__(256, 2) void k_loop_spread(const float* src, const float* cold, int pitch_f, int ktiles, int stages, float* out)
{
    __shared__ __align__(1024) float lds[DEPTH + 1][4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = 8 * wave + (lane >> 3);
    const float* base = src + (size_t)(blockIdx.x % 16) * 160 * pitch_f + (size_t)i * pitch_f + 4 * ((lane & 7) ^ ((i >> 1) & 7));
    f32x16 acc[3];
    for (int q = 0; q < 3; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    int kt = 0;
    const int xo = (lane & 31) * 32 + 4 * ((2 * wave + (lane >> 5)) ^ (((lane & 31) >> 1) & 7));
    for (int p = 0; p < DEPTH; ++p) {
        const float* q = base + (size_t)(kt++ % ktiles) * 32;
#pragma unroll
        for (int b = 0; b < 4; ++b) __builtin_amdgcn_global_load_lds(q + (size_t)b * 32 * pitch_f, LDSP(&lds[p][b * 1024 + wave * 256]), 16, 0, 0);
    }
    int cur = 0, nxt = DEPTH;
    for (int it = 0; it < stages; ++it) {
        wait_vm<4 * (DEPTH - 1)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        f32x4 f[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) f[b] = *(const f32x4*)&lds[cur][b * 1024 + xo];
        const float* q = base + (size_t)(kt++ % ktiles) * 32;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            acc[m % 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[0][m & 3], f[1 + m % 3][m & 3], acc[m % 3], 0, 0, 0);
            if (m % (NM / 4) == NM / 4 - 1) {
                const int b = m / (NM / 4);
                __builtin_amdgcn_global_load_lds(q + (size_t)b * 32 * pitch_f, LDSP(&lds[nxt][b * 1024 + wave * 256]), 16, 0, 0);
            }
        }
        nxt = nxt == DEPTH ? 0 : nxt + 1;
        cur = cur == DEPTH ? 0 : cur + 1;
    }
    wait_vm<0>();
    float t = 0.f;
    for (int q = 0; q < 3; ++q) for (int r = 0; r < 16; ++r) t += acc[q][r];
    if (t == 12345.678f) out[tid] = t;
}

// k_loop with the fragments of stage it + 1 read from LDS BEFORE the MFMAs of stage it are issued (register double buffer):
// the barrier -> fragment read -> first MFMA chain leaves the critical path
template <int DEPTH, int NM>
__global__ __launch_bounds__(256, 2) void k_loop_pipe(const float* src, const float* cold, int pitch_f, int ktiles, int stages, float* out)
{
    __shared__ __align__(1024) float lds[DEPTH + 1][4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = 8 * wave + (lane >> 3);
    const float* base = src + (size_t)(blockIdx.x % 16) * 160 * pitch_f + (size_t)i * pitch_f + 4 * ((lane & 7) ^ ((i >> 1) & 7));
    f32x16 acc[3];
    for (int q = 0; q < 3; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    int kt = 0;
    auto stage = [&](int buf, int k) {
        const float* p = base + (size_t)(k % ktiles) * 32;
#pragma unroll
        for (int b = 0; b < 4; ++b)
            __builtin_amdgcn_global_load_lds(p + (size_t)b * 32 * pitch_f, LDSP(&lds[buf][b * 1024 + wave * 256]), 16, 0, 0);
    };
    const int xo = (lane & 31) * 32 + 4 * ((2 * wave + (lane >> 5)) ^ (((lane & 31) >> 1) & 7));
    for (int p = 0; p < DEPTH; ++p) stage(p, kt++);
    // stage 0 -> registers
    wait_vm<4 * (DEPTH - 1)>();
    __builtin_amdgcn_s_barrier();
    f32x4 f[4], g[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) f[b] = *(const f32x4*)&lds[0][b * 1024 + xo];
    int cur = 1, nxt = DEPTH;
    for (int it = 0; it < stages; ++it) {
        // stage it + 1 has landed; buffer of stage it is free once every wave holds its fragments (they do: read before the barrier)
        wait_vm<4 * (DEPTH - 2)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage(nxt, kt++);
        nxt = nxt == DEPTH ? 0 : nxt + 1;
#pragma unroll
        for (int b = 0; b < 4; ++b) g[b] = *(const f32x4*)&lds[cur][b * 1024 + xo];
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m % 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[0][m & 3], f[1 + m % 3][m & 3], acc[m % 3], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < 4; ++b) f[b] = g[b];
        cur = cur == DEPTH ? 0 : cur + 1;
    }
    wait_vm<0>();
    float t = 0.f;
    for (int q = 0; q < 3; ++q) for (int r = 0; r < 16; ++r) t += acc[q][r];
    if (t == 12345.678f) out[tid] = t;
}

template <int DEPTH, bool COLD, int NM, bool SAMEM0 = false>
__global__ __launch_bounds__(256, 2) void k_loop(const float* src, const float* cold, int pitch_f, int ktiles, int stages, float* out)
{
    __shared__ __align__(1024) float lds[DEPTH + 1][4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = 8 * wave + (lane >> 3);
    const float* base = src + (size_t)(blockIdx.x % 16) * 160 * pitch_f + (size_t)i * pitch_f + 4 * ((lane & 7) ^ ((i >> 1) & 7));
    // cold rows: 32 rows per (group of 4 workgroups, 16 k-tiles), never re-read
    const float* cbase = cold + ((size_t)(blockIdx.x / 4) * (stages / 16 + 2) * 32 + i) * pitch_f + 4 * ((lane & 7) ^ ((i >> 1) & 7));
    f32x16 acc[3];
    for (int q = 0; q < 3; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    int kt = 0;
    auto stage = [&](int buf, int k) {
        const float* p = base + (size_t)(k % ktiles) * 32;
        const float* pc = COLD ? cbase + (size_t)(k / 16) * 32 * pitch_f + (size_t)(k % 16) * 32 : p;
        __builtin_amdgcn_global_load_lds(pc, LDSP(&lds[buf][wave * 256]), 16, 0, 0);
#pragma unroll
        for (int b = 1; b < 4; ++b)
            __builtin_amdgcn_global_load_lds(p + (size_t)b * 32 * pitch_f, LDSP(&lds[buf][(SAMEM0 ? 0 : b * 1024) + wave * 256]), 16, 0, 0);
    };
    for (int p = 0; p < DEPTH; ++p) stage(p, kt++);
    int cur = 0, nxt = DEPTH;
    for (int it = 0; it < stages; ++it) {
        wait_vm<4 * (DEPTH - 1)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage(nxt, kt++);
        nxt = nxt == DEPTH ? 0 : nxt + 1;
        f32x4 f[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) f[b] = *(const f32x4*)&lds[cur][b * 1024 + (lane & 31) * 32 + 4 * ((2 * wave + (lane >> 5)) ^ (((lane & 31) >> 1) & 7))];
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m % 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[0][m & 3], f[1 + m % 3][m & 3], acc[m % 3], 0, 0, 0);
        cur = cur == DEPTH ? 0 : cur + 1;
    }
    wait_vm<0>();
    float t = 0.f;
    for (int q = 0; q < 3; ++q) for (int r = 0; r < 16; ++r) t += acc[q][r];
    if (t == 12345.678f) out[tid] = t;
}

template <int DEPTH>
__global__ __launch_bounds__(256, 2) void k_dma(const float* src, int pitch_f, int ktiles, int stages, float* out)
{
    __shared__ __align__(1024) float lds[DEPTH + 1][4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = 8 * wave + (lane >> 3);
    const float* base = src + (size_t)(blockIdx.x % 16) * 160 * pitch_f + (size_t)i * pitch_f + 4 * ((lane & 7) ^ ((i >> 1) & 7));
    f32x4 acc = {0, 0, 0, 0};
    int kt = 0;
    auto stage = [&](int buf, int k) {
        const float* p = base + (size_t)(k % ktiles) * 32;
#pragma unroll
        for (int b = 0; b < 4; ++b)
            __builtin_amdgcn_global_load_lds(p + (size_t)b * 32 * pitch_f, LDSP(&lds[buf][b * 1024 + wave * 256]), 16, 0, 0);
    };
    for (int p = 0; p < DEPTH; ++p) stage(p, kt++);
    int cur = 0, nxt = DEPTH;
    for (int it = 0; it < stages; ++it) {
        wait_vm<4 * (DEPTH - 1)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage(nxt, kt++);
        nxt = nxt == DEPTH ? 0 : nxt + 1;
#pragma unroll
        for (int b = 0; b < 4; ++b) acc += *(const f32x4*)&lds[cur][b * 1024 + (lane & 31) * 32 + 4 * ((2 * wave + (lane >> 5)) ^ (((lane & 31) >> 1) & 7))];
        cur = cur == DEPTH ? 0 : cur + 1;
    }
    wait_vm<0>();
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[tid] = acc[0];
}

// the same fill with NW waves per workgroup sharing the issue (NW = 4: k_dma): is the bytes-in-flight limit per wave or per CU?
template <int DEPTH, int NW>
__global__ __launch_bounds__(64 * NW) void k_dma_w(const float* src, int pitch_f, int ktiles, int stages, float* out)
{
    __shared__ __align__(1024) float lds[DEPTH + 1][4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc = {0, 0, 0, 0};
    int kt = 0;
    // 16 pieces of 1 KB per stage (piece p: block p / 4, rows 8 (p % 4) ..), dealt round-robin to the NW waves
    auto stage = [&](int buf, int k) {
#pragma unroll
        for (int p = wave; p < 16; p += NW) {
            const int b = p >> 2, i = 8 * (p & 3) + (lane >> 3);
            const float* q = src + (size_t)(blockIdx.x % 16) * 160 * pitch_f + (size_t)(b * 32 + i) * pitch_f + 4 * ((lane & 7) ^ ((i >> 1) & 7)) + (size_t)(k % ktiles) * 32;
            __builtin_amdgcn_global_load_lds(q, LDSP(&lds[buf][p * 256]), 16, 0, 0);
        }
    };
    constexpr int PPW = (16 + NW - 1) / NW;
    for (int p = 0; p < DEPTH; ++p) stage(p, kt++);
    int cur = 0, nxt = DEPTH;
    for (int it = 0; it < stages; ++it) {
        if (wave < 16) wait_vm<PPW * (DEPTH - 1)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage(nxt, kt++);
        nxt = nxt == DEPTH ? 0 : nxt + 1;
        if (wave < 4) {
#pragma unroll
            for (int b = 0; b < 4; ++b) acc += *(const f32x4*)&lds[cur][b * 1024 + (lane & 31) * 32 + 4 * ((2 * wave + (lane >> 5)) ^ (((lane & 31) >> 1) & 7))];
        }
        cur = cur == DEPTH ? 0 : cur + 1;
    }
    wait_vm<0>();
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[tid] = acc[0];
}

template <int DEPTH>
__global__ __launch_bounds__(256, 2) void k_reg(const float* src, int pitch_f, int ktiles, int stages, float* out)
{
    __shared__ __align__(1024) float lds[2][4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = 8 * wave + (lane >> 3);
    const float* base = src + (size_t)(blockIdx.x % 16) * 160 * pitch_f + (size_t)i * pitch_f + 4 * ((lane & 7) ^ ((i >> 1) & 7));
    f32x4 acc = {0, 0, 0, 0};
    f32x4 r[DEPTH][4];
    int kt = 0;
    auto load = [&](int slot, int k) {
        const float* p = base + (size_t)(k % ktiles) * 32;
#pragma unroll
        for (int b = 0; b < 4; ++b) r[slot][b] = *(const f32x4*)(p + (size_t)b * 32 * pitch_f);
    };
#pragma unroll
    for (int p = 0; p < DEPTH; ++p) load(p, kt++);
    int cur = 0;
    for (int it0 = 0; it0 < stages; it0 += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            wait_vm<4 * (DEPTH - 1)>();
#pragma unroll
            for (int b = 0; b < 4; ++b) *(f32x4*)&lds[cur][b * 1024 + wave * 256 + lane * 4] = r[s][b];
            load(s, kt++);
            __syncthreads();
#pragma unroll
            for (int b = 0; b < 4; ++b) acc += *(const f32x4*)&lds[cur][b * 1024 + (lane & 31) * 32 + 4 * ((2 * wave + (lane >> 5)) ^ (((lane & 31) >> 1) & 7))];
            cur ^= 1;
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[tid] = acc[0];
}

int main()
{
    const int pitch_f = 512, rows = 16 * 160, ktiles = 16, stages = 4096;
    float *src, *out;
    hipMalloc(&src, sizeof(float) * (size_t)rows * pitch_f);
    hipMalloc(&out, 4096);
    hipMemset(src, 0, sizeof(float) * (size_t)rows * pitch_f);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto runw = [&](const char* name, auto kern, int grid, int threads) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            kern<<<dim3(grid), dim3(threads)>>>(src, pitch_f, ktiles, stages, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-28s grid %4d: %8.3f ms  %6.1f GB/s per workgroup  (%.2f us per 16 KB stage)\n", name, grid, ms, 16384.0 * stages / (ms * 1e6), ms * 1e3 / stages);
        }
    };
    runw("DMA by 1 wave, depth 3", k_dma_w<3, 1>, 256, 64);
    runw("DMA by 2 waves, depth 3", k_dma_w<3, 2>, 256, 128);
    runw("DMA by 4 waves, depth 3", k_dma_w<3, 4>, 256, 256);
    runw("DMA by 8 waves, depth 3", k_dma_w<3, 8>, 256, 512);
    runw("DMA by 16 waves, depth 3", k_dma_w<3, 16>, 256, 1024);
    runw("DMA by 1 wave, depth 2", k_dma_w<2, 1>, 256, 64);
    runw("DMA by 8 waves, depth 2", k_dma_w<2, 8>, 256, 512);
    auto run = [&](const char* name, auto kern, int grid) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            kern<<<dim3(grid), dim3(256)>>>(src, pitch_f, ktiles, stages, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-28s grid %4d: %8.3f ms  %6.1f GB/s per workgroup  %6.2f TB/s chip  (%.2f us per 16 KB stage)\n", name, grid, ms,
                            16384.0 * stages / (ms * 1e6), 16384.0 * stages * grid / (ms * 1e9), ms * 1e3 / stages);
        }
    };
    for (int grid : {256, 512}) {
        run("LDS-DMA depth 2", k_dma<2>, grid);
        run("LDS-DMA depth 3", k_dma<3>, grid);
        run("LDS-DMA depth 5", k_dma<5>, grid);
        run("registers depth 1", k_reg<1>, grid);
        run("registers depth 2", k_reg<2>, grid);
        run("registers depth 4", k_reg<4>, grid);
    }
    {
        float* cold;
        const size_t cold_rows = (size_t)128 * (stages / 16 + 2) * 32;
        hipMalloc(&cold, sizeof(float) * cold_rows * pitch_f);
        hipMemset(cold, 0, sizeof(float) * cold_rows * pitch_f);
        auto run2 = [&](const char* name, auto kern, int grid) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                kern<<<dim3(grid), dim3(256)>>>(src, cold, pitch_f, ktiles, stages, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("%-44s grid %4d: %8.3f ms  (%.2f us per 16 KB stage)\n", name, grid, ms, ms * 1e3 / stages);
            }
        };
        for (int grid : {256, 512}) {
            run2("K loop, L2-resident, 12 MFMA, depth 3", k_loop<3, false, 12>, grid);
            run2("K loop, L2-resident, 24 MFMA, depth 2", k_loop<2, false, 24>, grid);
            run2("K loop SPREAD loads, 12 MFMA, depth 3", k_loop_spread<3, 12>, grid);
            run2("K loop SPREAD loads, 24 MFMA, depth 3", k_loop_spread<3, 24>, grid);
            run2("K loop SPREAD loads, 12 MFMA, depth 2", k_loop_spread<2, 12>, grid);
            run2("K loop PIPELINED frags, 12 MFMA, depth 3", k_loop_pipe<3, 12>, grid);
            run2("K loop PIPELINED frags, 24 MFMA, depth 3", k_loop_pipe<3, 24>, grid);
            run2("K loop, L2-resident, 0 MFMA, depth 3", k_loop<3, false, 0>, grid);
            run2("K loop, L2-res, 0 MFMA, d3, one M0", k_loop<3, false, 0, true>, grid);
            run2("K loop, L2-res, 12 MFMA, d3, one M0", k_loop<3, false, 12, true>, grid);
            run2("K loop, L2-res, 24 MFMA, d3, one M0", k_loop<3, false, 24, true>, grid);
            run2("K loop, cold A block, 0 MFMA, depth 2", k_loop<2, true, 0>, grid);
            run2("K loop, cold A block, 0 MFMA, depth 3", k_loop<3, true, 0>, grid);
            run2("K loop, cold A block, 12 MFMA, depth 2", k_loop<2, true, 12>, grid);
            run2("K loop, cold A block, 12 MFMA, depth 3", k_loop<3, true, 12>, grid);
            run2("K loop, cold A block, 24 MFMA, depth 2", k_loop<2, true, 24>, grid);
            run2("K loop, cold A block, 24 MFMA, depth 3", k_loop<3, true, 24>, grid);
        }
    }
    run("LDS-DMA depth 3, one CU", k_dma<3>, 1);
    run("registers depth 2, one CU", k_reg<2>, 1);
    return 0;
}
