// Device helpers shared by the persistent StackGRU sequence kernels (ppk_seq_persist.hip forward,
// ppk_seq_persist_bwd.hip backward): 16-byte accesses, write-through / L1-bypassing hand-off accesses (cache policy
// sc1) as raw buffer instructions, the group-local signal / wait pair, accumulator transposition, LDS-DMA sources.
//
// Cross-workgroup hand-offs follow MI355X_MICROARCH.md's valid forms: producer = write-through (sc1) 16-byte stores,
// every wave drains vmcnt, workgroup barrier, ONE agent-scope counter add; consumer = ONE lane polls (relaxed, agent),
// agent acquire (buffer_inv sc1), vmcnt(0), workgroup barrier, then plain loads / LDS-DMA (sc1 loads where a buffer is
// rewritten every step).  Polls are bounded: on a timeout the kernel raises a status word and every workgroup leaves.
#pragma once
#include <hip/hip_runtime.h>

namespace ppk_seq {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int EPI_LD = 32;                     // floats per transposed accumulator row: a thread reads (row = tid / 8, 4 columns), so
                                               // every 16-lane read group covers 256 contiguous bytes without padding
constexpr int EPI_TILE = 32 * EPI_LD;
#ifndef PPK_SPIN_LIMIT
#define PPK_SPIN_LIMIT (1 << 21)
#endif
constexpr int SPIN_LIMIT = PPK_SPIN_LIMIT;     // polls ~1.5 us apart (PPK_POLL_SLEEP + the load): ~3 s, then the status word is set, every workgroup
                                               // leaves at its next wait and k_seq_poison turns every output of the launch into NaN
                                               // (libppk_spin.so, built with -DPPK_SPIN_LIMIT=1, forces that path: tests/test_gpu_stackgru.py)
#define PPK_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
// ints between two group counters: every (group, phase) counter sits on a line of its own, so that the polls of one
// group's consumers do not queue with the adds and polls of the other groups (32 groups x 3 counters were 4 lines)
#ifndef PPK_CTR_STRIDE
#define PPK_CTR_STRIDE 1024
#endif
constexpr int CTR_STRIDE = PPK_CTR_STRIDE;
// cache policy of the LDS-DMA loads of HANDED-OFF operands (hidden rows, stack tops, gate gradients, DCS rows):
// PPK_NO_ACQUIRE: sc1 (L1-bypassing) loads on every handed-off byte replace the agent acquire after the poll
// (MI355X_MICROARCH.md, Valid forms: every such byte is stored sc1, every storing wave drains vmcnt, one lane signals
// behind a workgroup barrier, the polling wave's workgroup barrier precedes every load)
#ifdef PPK_NO_ACQUIRE
constexpr int DMA_AUX_HANDOFF = 16;
#else
constexpr int DMA_AUX_HANDOFF = 0;
#endif

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
// Write-through / L1-bypassing 16-byte accesses of hand-off data (cache policy sc1) as raw buffer instructions:
// base = a wave-uniform pointer, byte_off = this lane's offset (< 2 GiB).  The compiler counts them in vmcnt itself.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int CPOL_SC1 = 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void st4_sc1(const float* base, long long byte_off, const f32x4& v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), make_rsrc(base), (int)byte_off, 0, CPOL_SC1);
}
__device__ __forceinline__ f32x4 ld4_sc1(const float* base, long long byte_off)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(base), (int)byte_off, 0, CPOL_SC1));
}

// ---- group-local synchronisation ----------------------------------------------------------
// every wave has drained its stores -> barrier -> one counter add
__device__ __forceinline__ void group_signal(int* ctr)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane polls, acquires, then the workgroup barrier; false = timeout / another workgroup gave up
// cycles / 64 between two polls of a group counter.  Round 2 kept all counters back to back (32 groups x 3 words = 4 lines):
// every poll queued with every other group's polls and adds on those lines, and the best interval was 127 (k_seq_bwd
// 20.65 / 20.13 / 19.56 / 19.12 / 18.70 / 19.16 ms at 1 / 8 / 24 / 64 / 127 / 254).  With every counter on a line of its own
// (PPK_CTR_STRIDE, 4 KB apart) the interval no longer matters (same-box runs, profiles/r03_handoff_ab.txt: k_seq_bwd
// 18.7 -> 17.9 ms from the padding alone, 17.75 at intervals 32 and 8); 32 is kept.
#ifndef PPK_POLL_SLEEP
#define PPK_POLL_SLEEP 32
#endif
__device__ __forceinline__ bool group_wait(int* ctr, int target, int* status, volatile int* flag)
{
    if (threadIdx.x == 0) {
        int ok = 0;
        for (int it = 0; it < SPIN_LIMIT; ++it) {
            if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
            if ((it & 255) == 255 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
#if PPK_POLL_SLEEP == 0
            // back-off: 8, 16, 32, 64, then 127 x 64 cycles between polls (a poll is an agent-scope load that queues with
            // the co-resident workgroup's streaming)
            if (it == 0) __builtin_amdgcn_s_sleep(8);
            else if (it == 1) __builtin_amdgcn_s_sleep(16);
            else if (it == 2) __builtin_amdgcn_s_sleep(32);
            else if (it == 3) __builtin_amdgcn_s_sleep(64);
            else __builtin_amdgcn_s_sleep(127);
#elif PPK_POLL_SLEEP > 127
            __builtin_amdgcn_s_sleep(127);
            __builtin_amdgcn_s_sleep(PPK_POLL_SLEEP - 127);
#else
            __builtin_amdgcn_s_sleep(PPK_POLL_SLEEP);
#endif
        }
        if (!ok) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifndef PPK_NO_ACQUIRE
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        *flag = ok;
    }
    __syncthreads();
    const bool ok = *flag != 0;
    __syncthreads();
    return ok;
}

__device__ __forceinline__ void tile_to_lds(float* t, const f32x16& acc, int li, int h)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) t[((r & 3) + 8 * (r >> 2) + 4 * h) * EPI_LD + li] = acc[r];
}

// this lane's LDS-DMA source inside a [rows x K] row-major operand: row `row` (already clamped), swizzled 16-B chunk
__device__ __forceinline__ const float* dma_src(const float* base, long long row, long long ld, int i, int lane)
{
    return base + row * ld + 4 * ((lane & 7) ^ ((i >> 1) & 7));
}



// ---- MFMA over this lane's four K values of a chunk (fragment f32x4 = 4 consecutive k of one row) ----------------
// fp32: four v_mfma_f32_32x32x2_f32.  BF (the bf16-operand variant, ppk_mfma_mode(1)): operands rounded to bf16
// (round to nearest even, v_cvt_pk_bf16_f32) in registers, ONE v_mfma_f32_32x32x8_bf16, fp32 accumulation; memory
// stays fp32.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x4 to_bf16x4(const f32x4& x)
{
    const bf16x2_ lo = __builtin_convertvector(f32x2_{x[0], x[1]}, bf16x2_), hi = __builtin_convertvector(f32x2_{x[2], x[3]}, bf16x2_);
    const u32x2_ r = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
    return __builtin_bit_cast(s16x4, r);
}
template <bool BF> struct Frag;
template <> struct Frag<false> {
    f32x4 v;
    __device__ __forceinline__ void set(const f32x4& x) { v = x; }
};
template <> struct Frag<true> {
    s16x4 v;
    __device__ __forceinline__ void set(const f32x4& x) { v = to_bf16x4(x); }
};
__device__ __forceinline__ void mma_chunk(f32x16& acc, const Frag<false>& a, const Frag<false>& b, int st)
{
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[st], b.v[st], acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_chunk(f32x16& acc, const Frag<true>& a, const Frag<true>& b, int st)
{
    if (st == 0) acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a.v, b.v, acc, 0, 0, 0);
}

}  // namespace ppk_seq
