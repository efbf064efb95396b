// Shared helpers for the ppk (paccmann-polymer kernels) C-ABI library.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define PPK_OK            0
#define PPK_ERR_SHAPE    -1   // bad / inconsistent sizes
#define PPK_ERR_ALIGN    -2   // misaligned pointer or leading dimension
#define PPK_ERR_WS       -3   // workspace too small
#define PPK_ERR_UNSUP    -4   // size beyond what the kernels support
#define PPK_ERR_HIP      -5   // a HIP runtime call / launch failed
#define PPK_ERR_NULL     -6   // required pointer is null

#define PPK_F32 0
#define PPK_F64 1

void ppk_set_error(const char* fmt, ...);

#define PPK_FAIL(code, ...)            \
    do {                               \
        ppk_set_error(__VA_ARGS__);    \
        return (code);                 \
    } while (0)

#define PPK_CHECK_LAUNCH(what)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) PPK_FAIL(PPK_ERR_HIP, "%s: %s", what, hipGetErrorString(e__)); \
    } while (0)

#define PPK_HIP(call)                                                            \
    do {                                                                         \
        hipError_t e__ = (call);                                                 \
        if (e__ != hipSuccess) PPK_FAIL(PPK_ERR_HIP, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

static inline size_t ppk_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// carve `bytes` (256-B aligned) out of a workspace cursor
static inline void* ppk_carve(char*& cur, size_t bytes)
{
    void* p = cur;
    cur += ppk_align_up(bytes, 256);
    return p;
}

__device__ __forceinline__ float ppk_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double ppk_wave_sum_d(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float ppk_wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
