// K6 — CSR build (stable sort of the edge list by target) and
// K7 — CSR gather + segmented-reduce SpMM (forward and, on the transposed CSR, backward).
//
// Replaces the message passing of torch_geometric GCNConv / torch_scatter
// scatter_add that the reference calls at
//   .../experiments/chem_reactions/reactions_infomax.py:60,66   (GCNConv(256, 512)(x, edge_index))
// (gcn_norm + gather x_j + scatter-add at the target; semantics SURVEY.md Appendix C).
// scatter_add on a GPU uses float atomics (run-to-run different bits); here every
// output row is reduced by ONE wavefront in CSR order, so the result is deterministic
// and equals a CPU scatter_add that walks the edge list in order:
//   row i = edges with target i in ORIGINAL edge-list order, self loop last.
//
// CSR build without a global sort: integer histogram -> exclusive scan -> atomic
// fill (arbitrary order inside a row) -> per-row sort by edge id (unique keys, so the
// final order is deterministic).  Short rows are sorted by one lane each, long rows
// (hubs of power-law relation graphs) by a workgroup bitonic sort in LDS; rows beyond
// one LDS image (> 8192 entries) are sorted in 8192-entry chunks and the chunks merged by
// grid-wide rank-merge passes (every entry finds its place in the paired run by binary
// search; keys are unique), ping-ponging between `eid` and the not yet written `col`.
// k_finish and the SpMM give such rows to many lanes as well (k_finish_long, k_spmm_long):
// a hub with 5.5e5 entries took 170 s to build with one workgroup's transposition sort.
#include "ppk_common.h"
#include "ppk_kernels.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int SCAN_BLOCK = 1024;     // elements per scan block (256 threads x 4)
constexpr int SHORT_ROW = 16;
constexpr int LDS_SORT_MAX = 8192;
constexpr int FINISH_LONG = 64;      // rows beyond this are finished by k_finish_long
constexpr int SPMM_LONG = 1024;      // rows beyond this are reduced by a whole workgroup (k_spmm_long) or, with a workspace, in chunks
constexpr int SPMM_CHUNK = 1024;     // entries of a long row per wave of k_spmm_chunks
struct LongRow { long long row; long long base; };      // a long row and the id of its first chunk

inline unsigned grid_for(long long work, unsigned cap = 8192)
{
    long long b = (work + 255) / 256;
    if (b < 1) b = 1;
    return (unsigned)(b < cap ? b : cap);
}

// Block-level aggregation of same-target edges before they touch global memory.  On a power-law graph a third of the
// edges can point at a handful of hubs: 1.7 M global atomics on 3 addresses serialise (the 3-hub graph of configs[2]
// took 35 ms to build, nearly all of it in k_count / k_fill).  Each block of 256 edges first groups its targets in an
// LDS hash table (LDS atomics: a few cycles each), then issues ONE global atomic per distinct target.
constexpr int AGG_SLOTS = 512;                   // >= 2 x 256 distinct keys: linear probing always terminates
struct AggTable { int key[AGG_SLOTS]; int cnt[AGG_SLOTS]; int base[AGG_SLOTS]; };
__device__ __forceinline__ void agg_clear(AggTable& t)
{
    for (int s = threadIdx.x; s < AGG_SLOTS; s += 256) { t.key[s] = -1; t.cnt[s] = 0; }
}
// slot of key d (inserted if new); returns this lane's rank among the block's lanes with the same key
__device__ __forceinline__ int agg_insert(AggTable& t, int d, int& slot)
{
    slot = (int)(((unsigned)d * 2654435761u) >> 23) & (AGG_SLOTS - 1);
    for (;;) {
        const int old = atomicCAS(&t.key[slot], -1, d);
        if (old == -1 || old == d) break;
        slot = (slot + 1) & (AGG_SLOTS - 1);
    }
    return atomicAdd(&t.cnt[slot], 1);
}

// cnt[target] += 1 for every non-self-loop edge (integer atomics: order-independent result)
__global__ __launch_bounds__(256) void k_count(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, long long E,
                                               long long N, long long NS, int drop_self_loops, int* __restrict__ cnt, int* __restrict__ bad)
{
    __shared__ AggTable t;
    for (long long e0 = (long long)blockIdx.x * 256; e0 < E; e0 += (long long)gridDim.x * 256) {      // block-uniform trip count
        agg_clear(t);
        __syncthreads();
        const long long e = e0 + threadIdx.x;
        if (e < E) {
            const long long s = src[e], d = dst[e];
            if (s < 0 || s >= NS || d < 0 || d >= N) atomicOr(bad, 1);
            else if (!(drop_self_loops && s == d)) { int slot; agg_insert(t, (int)d, slot); }
        }
        __syncthreads();
        for (int s = threadIdx.x; s < AGG_SLOTS; s += 256) if (t.key[s] >= 0) atomicAdd(&cnt[t.key[s]], t.cnt[s]);
        __syncthreads();
    }
}

// ---- exclusive scan of (cnt[i] + extra) over N entries -> rowptr (int64) ------
__global__ __launch_bounds__(256) void k_scan_block(const int* __restrict__ cnt, int extra, long long N,
                                                    long long* __restrict__ rowptr, long long* __restrict__ block_sums)
{
    __shared__ long long wsum[4];
    const long long base = (long long)blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
    long long v[4], tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = (base + k < N) ? (long long)cnt[base + k] + extra : 0;
        tot += v[k];
    }
    // inclusive scan of `tot` across the block
    long long inc = tot;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        long long t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    long long woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    long long excl = woff + inc - tot;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < N) rowptr[base + k] = excl;     // block-local exclusive prefix
        excl += v[k];
    }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = woff + inc;
}
__global__ __launch_bounds__(256) void k_scan_sums(long long* __restrict__ block_sums, int nblocks, long long* __restrict__ total_out)
{
    // single block: sequential chunks of 256 with a running carry (nblocks <= a few thousand)
    __shared__ long long wsum[4];
    __shared__ long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblocks; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const long long v = i < nblocks ? block_sums[i] : 0;
        long long inc = v;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            long long t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        long long woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const long long carry = carry_s;
        if (i < nblocks) block_sums[i] = carry + woff + inc - v;    // exclusive
        __syncthreads();
        if (threadIdx.x == 255) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry_s;
}
__global__ __launch_bounds__(256) void k_scan_add(long long* __restrict__ rowptr, const long long* __restrict__ block_sums,
                                                  const long long* __restrict__ total, long long N)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < N) rowptr[i] += block_sums[i / SCAN_BLOCK];
    if (i == N) rowptr[N] = *total;
}

// fill: arbitrary position inside the row; the self loop gets edge id E + i (sorts last).  Same block-level grouping
// as k_count: one global cursor atomic per distinct target of the block, positions handed out from the LDS ranks.
__global__ __launch_bounds__(256) void k_fill(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, long long E,
                                              long long N, long long NS, int drop_self_loops, int add_self_loops,
                                              const long long* __restrict__ rowptr, int* __restrict__ cursor,
                                              long long* __restrict__ eid)
{
    __shared__ AggTable t;
    const long long total = E + (add_self_loops ? N : 0);
    for (long long e0 = (long long)blockIdx.x * 256; e0 < total; e0 += (long long)gridDim.x * 256) {  // block-uniform trip count
        agg_clear(t);
        __syncthreads();
        const long long e = e0 + threadIdx.x;
        long long d = -1;
        if (e < E) {
            const long long s = src[e];
            d = dst[e];
            if (s < 0 || s >= NS || d < 0 || d >= N) d = -1;
            else if (drop_self_loops && s == d) d = -1;
        } else if (e < total) {
            d = e - E;
        }
        int slot = 0, rank = 0;
        if (d >= 0) rank = agg_insert(t, (int)d, slot);
        __syncthreads();
        for (int s = threadIdx.x; s < AGG_SLOTS; s += 256) if (t.key[s] >= 0) t.base[s] = atomicAdd(&cursor[t.key[s]], t.cnt[s]);
        __syncthreads();
        if (d >= 0) eid[rowptr[d] + t.base[slot] + rank] = e;
        __syncthreads();
    }
}

// per-row sort by edge id.  Short rows: one lane each.  Long rows are queued.
__global__ __launch_bounds__(256) void k_sort_short(const long long* __restrict__ rowptr, long long N, long long* __restrict__ eid,
                                                    int* __restrict__ long_rows, int* __restrict__ n_long)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
        const long long b = rowptr[i], len = rowptr[i + 1] - b;
        if (len <= 1) continue;
        if (len > SHORT_ROW) { long_rows[atomicAdd(n_long, 1)] = (int)i; continue; }
        long long v[SHORT_ROW];
#pragma unroll
        for (int k = 0; k < SHORT_ROW; ++k) v[k] = k < len ? eid[b + k] : 0x7fffffffffffffffLL;
        // odd-even transposition network on registers (fully unrolled, no dynamic indexing)
#pragma unroll
        for (int pass = 0; pass < SHORT_ROW; ++pass)
#pragma unroll
            for (int k = pass & 1; k + 1 < SHORT_ROW; k += 2) {
                const long long lo = v[k] < v[k + 1] ? v[k] : v[k + 1];
                const long long hi = v[k] < v[k + 1] ? v[k + 1] : v[k];
                v[k] = lo; v[k + 1] = hi;
            }
#pragma unroll
        for (int k = 0; k < SHORT_ROW; ++k)
            if (k < len) eid[b + k] = v[k];
    }
}
// bitonic sort of a[0..len) (len <= LDS_SORT_MAX) through the LDS image `buf`, by the whole workgroup
__device__ inline void lds_sort(long long* __restrict__ buf, long long* __restrict__ a, int len)
{
    int n2 = 1;
    while (n2 < len) n2 <<= 1;
    for (int k = threadIdx.x; k < n2; k += 256) buf[k] = k < len ? a[k] : 0x7fffffffffffffffLL;
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int k = threadIdx.x; k < n2 / 2; k += 256) {
                const int lo = 2 * k - (k & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const long long x = buf[lo], y = buf[hi];
                if ((x > y) == up) { buf[lo] = y; buf[hi] = x; }
            }
            __syncthreads();
        }
    for (int k = threadIdx.x; k < len; k += 256) a[k] = buf[k];
    __syncthreads();
}
// long rows: one workgroup per row, bitonic sort in LDS.  Rows beyond one LDS image are sorted chunk by chunk
// and queued for the merge passes.
__global__ __launch_bounds__(256) void k_sort_long(const long long* __restrict__ rowptr, long long* __restrict__ eid,
                                                   const int* __restrict__ long_rows, const int* __restrict__ n_long,
                                                   int* __restrict__ vlong_rows, int* __restrict__ n_vlong)
{
    __shared__ long long buf[LDS_SORT_MAX];
    const int nl = *n_long;
    for (int q = blockIdx.x; q < nl; q += gridDim.x) {
        const long long i = long_rows[q];
        const long long b = rowptr[i];
        const long long len = rowptr[i + 1] - b;
        if (len <= LDS_SORT_MAX) lds_sort(buf, eid + b, (int)len);
        else if (threadIdx.x == 0) vlong_rows[atomicAdd(n_vlong, 1)] = (int)i;       // k_sort_vlong_chunks + merge passes
    }
}
// rows beyond one LDS image: every LDS_SORT_MAX-entry chunk is sorted by its own workgroup (grid.x over the chunks of a
// row, grid.y over the queued rows).  One workgroup per ROW walked a 5.5e5-entry hub's 68 chunks one after the other:
// 11 ms of the 14 ms that the 3-hub graph of configs[2] took to build.
__global__ __launch_bounds__(256) void k_sort_vlong_chunks(const long long* __restrict__ rowptr, long long* __restrict__ eid,
                                                           const int* __restrict__ vlong_rows, const int* __restrict__ n_vlong)
{
    __shared__ long long buf[LDS_SORT_MAX];
    const int nv = *n_vlong;
    for (int q = blockIdx.y; q < nv; q += gridDim.y) {
        const long long i = vlong_rows[q];
        const long long b = rowptr[i], len = rowptr[i + 1] - b;
        for (long long c0 = (long long)blockIdx.x * LDS_SORT_MAX; c0 < len; c0 += (long long)gridDim.x * LDS_SORT_MAX)
            lds_sort(buf, eid + b + c0, (int)(len - c0 < LDS_SORT_MAX ? len - c0 : LDS_SORT_MAX));
    }
}
// merge pass p over the queued rows: sorted runs of LDS_SORT_MAX << p entries are merged in pairs.  Every entry
// computes its own output position: its index in its run plus the number of smaller keys in the partner run.
// Even passes read `eid` and write `scratch`, odd passes the other way; a row whose runs already cover it is skipped.
__global__ __launch_bounds__(256) void k_merge_pass(const long long* __restrict__ rowptr, long long* __restrict__ eid,
                                                    long long* __restrict__ scratch, const int* __restrict__ vlong_rows,
                                                    const int* __restrict__ n_vlong, int p)
{
    const int nv = *n_vlong;
    const long long run = (long long)LDS_SORT_MAX << p;
    for (int q = blockIdx.y; q < nv; q += gridDim.y) {
        const long long i = vlong_rows[q];
        const long long b = rowptr[i], len = rowptr[i + 1] - b;
        if (run >= len) continue;
        const long long* __restrict__ s = (p & 1) ? scratch + b : eid + b;
        long long* __restrict__ d = (p & 1) ? eid + b : scratch + b;
        for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < len; k += (long long)gridDim.x * 256) {
            const long long v = s[k];
            const long long base = k / (2 * run) * (2 * run), mid = base + run;
            if (mid >= len) { d[k] = v; continue; }                 // a run without a partner is carried over
            const long long end = base + 2 * run < len ? base + 2 * run : len;
            long long lo = k < mid ? mid : base, hi = k < mid ? end : mid;
            const long long other0 = lo;
            while (lo < hi) {
                const long long m = (lo + hi) >> 1;
                if (s[m] < v) lo = m + 1; else hi = m;
            }
            d[base + (k < mid ? k - base : k - mid) + (lo - other0)] = v;
        }
    }
}
// rows that went through an odd number of passes have their sorted entries in `scratch`
__global__ __launch_bounds__(256) void k_merge_finish(const long long* __restrict__ rowptr, long long* __restrict__ eid,
                                                      const long long* __restrict__ scratch, const int* __restrict__ vlong_rows,
                                                      const int* __restrict__ n_vlong)
{
    const int nv = *n_vlong;
    for (int q = blockIdx.y; q < nv; q += gridDim.y) {
        const long long i = vlong_rows[q];
        const long long b = rowptr[i], len = rowptr[i + 1] - b;
        int passes = 0;
        while (((long long)LDS_SORT_MAX << passes) < len) ++passes;
        if (!(passes & 1)) continue;
        for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < len; k += (long long)gridDim.x * 256) eid[b + k] = scratch[b + k];
    }
}

// sources and symmetric-normalisation weights in CSR order
__global__ __launch_bounds__(256) void k_finish(const int64_t* __restrict__ src, long long E, long long N,
                                                const long long* __restrict__ rowptr, const long long* __restrict__ eid,
                                                const int* __restrict__ deg_cnt, int deg_extra, int normalize,
                                                const float* __restrict__ edge_weight,
                                                long long* __restrict__ col, float* __restrict__ val)
{
    // one thread per row walks its entries (rows are short on average); grid-stride
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
        if (rowptr[i + 1] - rowptr[i] > FINISH_LONG) continue;          // k_finish_long
        const float di = normalize ? rsqrtf((float)(deg_cnt[i] + deg_extra)) : 1.0f;
        for (long long j = rowptr[i]; j < rowptr[i + 1]; ++j) {
            const long long e = eid[j];
            const long long s = e < E ? src[e] : e - E;
            col[j] = s;
            float w = (e < E && edge_weight) ? edge_weight[e] : 1.0f;
            if (normalize) w *= di * rsqrtf((float)(deg_cnt[s] + deg_extra));
            val[j] = w;
        }
    }
}

// the same for rows beyond FINISH_LONG entries (taken from the long-row queue): blockIdx.y walks the rows,
// blockIdx.x and the lanes walk the entries
__global__ __launch_bounds__(256) void k_finish_long(const int64_t* __restrict__ src, long long E,
                                                     const long long* __restrict__ rowptr, const long long* __restrict__ eid,
                                                     const int* __restrict__ deg_cnt, int deg_extra, int normalize,
                                                     const float* __restrict__ edge_weight, const int* __restrict__ long_rows,
                                                     const int* __restrict__ n_long, long long* __restrict__ col, float* __restrict__ val)
{
    const int nl = *n_long;
    for (int q = blockIdx.y; q < nl; q += gridDim.y) {
        const long long i = long_rows[q];
        const long long b = rowptr[i], e1 = rowptr[i + 1];
        if (e1 - b <= FINISH_LONG) continue;
        const float di = normalize ? rsqrtf((float)(deg_cnt[i] + deg_extra)) : 1.0f;
        for (long long j = b + (long long)blockIdx.x * 256 + threadIdx.x; j < e1; j += (long long)gridDim.x * 256) {
            const long long e = eid[j];
            const long long s = e < E ? src[e] : e - E;
            col[j] = s;
            float w = (e < E && edge_weight) ? edge_weight[e] : 1.0f;
            if (normalize) w *= di * rsqrtf((float)(deg_cnt[s] + deg_extra));
            val[j] = w;
        }
    }
}

// ---- K7: out[i, :] = sum_j val[j] * x[col[j], :]   one wavefront per row -------
template <int VEC>   // floats per lane per pass (4 -> float4)
__global__ __launch_bounds__(256) void k_spmm(const long long* __restrict__ rowptr, const long long* __restrict__ col,
                                              const float* __restrict__ val, const float* __restrict__ x, long long ldx,
                                              long long N, int F, float* __restrict__ out, long long ldo,
                                              unsigned long long* __restrict__ long_ctr, LongRow* __restrict__ long_slots)
{
    const int lane = threadIdx.x & 63;
    const long long wave_global = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * 4;
    for (long long i = wave_global; i < N; i += nwaves) {
        const long long b = rowptr[i], e = rowptr[i + 1];
        if (e - b > SPMM_LONG) {                                        // k_spmm_long, or (with a list) the chunk kernels
            if (long_ctr && lane == 0) {
                // slot number and first chunk id from ONE 64-bit atomic: slots come out sorted by their first chunk
                const unsigned long long nch = (unsigned long long)((e - b + SPMM_CHUNK - 1) / SPMM_CHUNK);
                const unsigned long long old = atomicAdd(long_ctr, (1ull << 40) | nch);
                long_slots[old >> 40] = LongRow{i, (long long)(old & ((1ull << 40) - 1))};
            }
            continue;
        }
        for (int f0 = lane * VEC; f0 < F; f0 += 64 * VEC) {
            if constexpr (VEC == 4) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                long long j = b;
                for (; j + 1 < e; j += 2) {            // two gathers in flight
                    const long long c0 = col[j], c1 = col[j + 1];
                    const float w0 = val[j], w1 = val[j + 1];
                    const f32x4 v0 = *(const f32x4*)(x + c0 * ldx + f0);
                    const f32x4 v1 = *(const f32x4*)(x + c1 * ldx + f0);
                    acc += w0 * v0;
                    acc += w1 * v1;
                }
                if (j < e) acc += val[j] * *(const f32x4*)(x + col[j] * ldx + f0);
                *(f32x4*)(out + i * ldo + f0) = acc;
            } else {
                float acc = 0.f;
                for (long long j = b; j < e; ++j) acc = fmaf(val[j], x[col[j] * ldx + f0], acc);
                out[i * ldo + f0] = acc;
            }
        }
    }
}

// Rows beyond SPMM_LONG entries: each workgroup (16 waves) scans its stripe of rowptr, and reduces every long row it
// finds with all its waves: wave w takes the w-th contiguous sixteenth of the row, four gathers in flight, and the
// sixteen partial sums are added in wave order (a fixed association: bitwise reproducible, but not the
// entry-by-entry order of the short rows).
template <int VEC>
__global__ __launch_bounds__(1024) void k_spmm_long(const long long* __restrict__ rowptr, const long long* __restrict__ col,
                                                    const float* __restrict__ val, const float* __restrict__ x, long long ldx,
                                                    long long N, int F, float* __restrict__ out, long long ldo)
{
    __shared__ int list[1024];
    __shared__ int nlist;
    __shared__ float part[16][64 * VEC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long per = (N + gridDim.x - 1) / gridDim.x;
    const long long r_begin = per * blockIdx.x, r_end = r_begin + per < N ? r_begin + per : N;
    for (long long r0 = r_begin; r0 < r_end; r0 += 1024) {
        const long long i = r0 + threadIdx.x;
        const bool is_long = i < r_end && rowptr[i + 1] - rowptr[i] > SPMM_LONG;
        if (!__syncthreads_or(is_long)) continue;
        if (threadIdx.x == 0) nlist = 0;
        __syncthreads();
        if (is_long) list[atomicAdd(&nlist, 1)] = (int)(i - r0);
        __syncthreads();
        const int n = nlist;
        for (int q = 0; q < n; ++q) {
            const long long row = r0 + list[q];
            const long long b = rowptr[row], len = rowptr[row + 1] - b;
            const long long j0 = b + len * wave / 16, j1 = b + len * (wave + 1) / 16;
            for (int f0 = 0; f0 < F; f0 += 64 * VEC) {
                const int f = f0 + lane * VEC;
                if constexpr (VEC == 4) {
                    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
                    if (f < F) {
                        long long j = j0;
                        for (; j + 3 < j1; j += 4) {
                            const f32x4 v0 = *(const f32x4*)(x + col[j] * ldx + f);
                            const f32x4 v1 = *(const f32x4*)(x + col[j + 1] * ldx + f);
                            const f32x4 v2 = *(const f32x4*)(x + col[j + 2] * ldx + f);
                            const f32x4 v3 = *(const f32x4*)(x + col[j + 3] * ldx + f);
                            a0 += val[j] * v0; a1 += val[j + 1] * v1; a2 += val[j + 2] * v2; a3 += val[j + 3] * v3;
                        }
                        for (; j < j1; ++j) a0 += val[j] * *(const f32x4*)(x + col[j] * ldx + f);
                    }
                    *(f32x4*)&part[wave][lane * 4] = (a0 + a1) + (a2 + a3);
                } else {
                    float a0 = 0.f;
                    if (f < F)
                        for (long long j = j0; j < j1; ++j) a0 = fmaf(val[j], x[col[j] * ldx + f], a0);
                    part[wave][lane] = a0;
                }
                __syncthreads();
                if (wave == 0 && f < F) {
                    if constexpr (VEC == 4) {
                        f32x4 t = *(const f32x4*)&part[0][lane * 4];
                        for (int w = 1; w < 16; ++w) t += *(const f32x4*)&part[w][lane * 4];
                        *(f32x4*)(out + row * ldo + f) = t;
                    } else {
                        float t = part[0][lane];
                        for (int w = 1; w < 16; ++w) t += part[w][lane];
                        out[row * ldo + f] = t;
                    }
                }
                __syncthreads();
            }
        }
        __syncthreads();
    }
}

// Long rows with a workspace (ppk_spmm_csr_ws): a hub's entries are cut into SPMM_CHUNK-entry chunks, ONE WAVE PER CHUNK
// over the whole grid (a row holding a third of all edges no longer serialises behind one workgroup), partial rows
// go to part [chunks][F]; k_spmm_long_sum then adds a row's partials in chunk order (fixed association: bitwise
// reproducible whatever order the rows were listed in).
template <int VEC>
__global__ __launch_bounds__(256) void k_spmm_chunks(const long long* __restrict__ rowptr, const long long* __restrict__ col,
                                                     const float* __restrict__ val, const float* __restrict__ x, long long ldx, int F,
                                                     const unsigned long long* __restrict__ long_ctr, const LongRow* __restrict__ slots,
                                                     float* __restrict__ part)
{
    const int lane = threadIdx.x & 63;
    const long long wave_global = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const unsigned long long ctr = long_ctr[0];
    const long long total = (long long)(ctr & ((1ull << 40) - 1)), nslots = (long long)(ctr >> 40);
    for (long long c = wave_global; c < total; c += nwaves) {
        long long lo = 0, hi = nslots;                       // largest slot with base <= c
        while (hi - lo > 1) { const long long mid = (lo + hi) >> 1; if (slots[mid].base <= c) lo = mid; else hi = mid; }
        const long long row = slots[lo].row;
        const long long j0 = rowptr[row] + (c - slots[lo].base) * SPMM_CHUNK;
        const long long j1 = j0 + SPMM_CHUNK < rowptr[row + 1] ? j0 + SPMM_CHUNK : rowptr[row + 1];
        for (int f = lane * VEC; f < F; f += 64 * VEC) {
            if constexpr (VEC == 4) {
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
                long long j = j0;
                for (; j + 3 < j1; j += 4) {
                    const f32x4 v0 = *(const f32x4*)(x + col[j] * ldx + f);
                    const f32x4 v1 = *(const f32x4*)(x + col[j + 1] * ldx + f);
                    const f32x4 v2 = *(const f32x4*)(x + col[j + 2] * ldx + f);
                    const f32x4 v3 = *(const f32x4*)(x + col[j + 3] * ldx + f);
                    a0 += val[j] * v0; a1 += val[j + 1] * v1; a2 += val[j + 2] * v2; a3 += val[j + 3] * v3;
                }
                for (; j < j1; ++j) a0 += val[j] * *(const f32x4*)(x + col[j] * ldx + f);
                *(f32x4*)(part + c * F + f) = (a0 + a1) + (a2 + a3);
            } else {
                float a0 = 0.f;
                for (long long j = j0; j < j1; ++j) a0 = fmaf(val[j], x[col[j] * ldx + f], a0);
                part[c * F + f] = a0;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_spmm_long_sum(const long long* __restrict__ rowptr, const unsigned long long* __restrict__ long_ctr,
                                                       const LongRow* __restrict__ slots, const float* __restrict__ part, int F,
                                                       float* __restrict__ out, long long ldo)
{
    const int lane = threadIdx.x & 63;
    const long long wave_global = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const long long nslots = (long long)(long_ctr[0] >> 40);
    for (long long s = wave_global; s < nslots; s += nwaves) {
        const long long row = slots[s].row, base = slots[s].base;
        const long long nch = (rowptr[row + 1] - rowptr[row] + SPMM_CHUNK - 1) / SPMM_CHUNK;
        for (int f = lane; f < F; f += 64) {
            float acc = part[base * F + f];
            for (long long k = 1; k < nch; ++k) acc += part[(base + k) * F + f];
            out[row * ldo + f] = acc;
        }
    }
}

struct CsrWs { int* cnt; int* cursor; long long* block_sums; long long* total; int* long_rows; int* n_long; int* bad; };

size_t csr_ws_bytes(int64_t N)
{
    const size_t nb = (size_t)((N + SCAN_BLOCK - 1) / SCAN_BLOCK);
    return 3 * ppk_align_up(sizeof(int) * (size_t)N, 256) + ppk_align_up(sizeof(long long) * nb, 256) + 3 * 256;
}
CsrWs csr_ws_carve(void* ws, int64_t N)
{
    char* cur = (char*)ws;
    CsrWs w;
    w.cnt = (int*)ppk_carve(cur, sizeof(int) * (size_t)N);
    w.cursor = (int*)ppk_carve(cur, sizeof(int) * (size_t)N);
    w.long_rows = (int*)ppk_carve(cur, sizeof(int) * (size_t)N);
    w.block_sums = (long long*)ppk_carve(cur, sizeof(long long) * (size_t)((N + SCAN_BLOCK - 1) / SCAN_BLOCK));
    w.total = (long long*)ppk_carve(cur, 256);
    w.n_long = (int*)ppk_carve(cur, 256);
    w.bad = (int*)ppk_carve(cur, 256);
    return w;
}

}  // namespace

extern "C" {

size_t ppk_csr_build_workspace_bytes(int64_t N, int64_t E)
{
    (void)E;
    if (N < 1) return 0;
    return csr_ws_bytes(N);
}

// Builds the CSR of the (optionally self-loop-augmented) graph grouped by `dst`:
//   rowptr [N+1] int64, col [nnz] int64 (= src of each entry), eid [nnz] int64 (original edge id, E + i for
//   the appended self loop of node i), val [nnz] fp32, with nnz <= E + N (caller allocates E + N entries).
// Row i lists the edges with dst == i in ORIGINAL edge order, self loop last (stable sort by target).
//   drop_self_loops / add_self_loops : gcn_norm's remove_self_loops + add_self_loops
//   normalize : val = deg^-1/2[src] * w * deg^-1/2[dst], deg = weighted in-degree incl. self loop (unit weights only)
//   edge_weight : optional [E]; must be NULL when normalize != 0
// status[0] (device int) is set non-zero when an edge endpoint is out of range (such edges are skipped).
int ppk_csr_build_rect(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int64_t N_src, int drop_self_loops,
                       int add_self_loops, int normalize, const float* edge_weight, int64_t* rowptr, int64_t* col, int64_t* eid,
                       float* val, int* status, void* workspace, size_t ws_bytes, hipStream_t stream);
int ppk_csr_build(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int drop_self_loops, int add_self_loops,
                  int normalize, const float* edge_weight, int64_t* rowptr, int64_t* col, int64_t* eid, float* val,
                  int* status, void* workspace, size_t ws_bytes, hipStream_t stream)
{
    return ppk_csr_build_rect(src, dst, E, N, N, drop_self_loops, add_self_loops, normalize, edge_weight, rowptr, col, eid, val,
                              status, workspace, ws_bytes, stream);
}

// The same grouping for a RECTANGULAR incidence: dst in [0, N) names the CSR row, src in [0, N_src) the gathered row of
// another table (the embedding-table gradient groups the NT packed rows of a batch by their token, N = V).  Self loops
// and the symmetric normalisation only exist for square graphs (N_src == N).
int ppk_csr_build_rect(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int64_t N_src, int drop_self_loops,
                       int add_self_loops, int normalize, const float* edge_weight, int64_t* rowptr, int64_t* col, int64_t* eid,
                       float* val, int* status, void* workspace, size_t ws_bytes, hipStream_t stream)
{
    if (N < 1 || E < 0 || N_src < 1) PPK_FAIL(PPK_ERR_SHAPE, "csr_build: N < 1 or E < 0");
    if (N_src != N && (drop_self_loops || add_self_loops || normalize)) PPK_FAIL(PPK_ERR_UNSUP, "csr_build: self loops / normalisation need a square graph");
    if ((E > 0 && (!src || !dst)) || !rowptr || !col || !eid || !val || !status || !workspace) PPK_FAIL(PPK_ERR_NULL, "csr_build: null pointer");
    if (N > 0x7fffffff || E + N > 0x7fffffffLL * 4) PPK_FAIL(PPK_ERR_UNSUP, "csr_build: graph too large");
    if (normalize && edge_weight) PPK_FAIL(PPK_ERR_UNSUP, "csr_build: normalize with edge weights is not implemented");
    if (ws_bytes < csr_ws_bytes(N)) PPK_FAIL(PPK_ERR_WS, "csr_build: workspace too small");
    CsrWs w = csr_ws_carve(workspace, N);
    const int nb = (int)((N + SCAN_BLOCK - 1) / SCAN_BLOCK);
    PPK_HIP(hipMemsetAsync(w.cnt, 0, sizeof(int) * (size_t)N, stream));
    PPK_HIP(hipMemsetAsync(w.cursor, 0, sizeof(int) * (size_t)N, stream));
    PPK_HIP(hipMemsetAsync(w.n_long, 0, 256, stream));
    PPK_HIP(hipMemsetAsync(w.bad, 0, 256, stream));
    if (E > 0) {
        k_count<<<dim3(grid_for(E)), dim3(256), 0, stream>>>(src, dst, E, N, N_src, drop_self_loops, w.cnt, w.bad);
        PPK_CHECK_LAUNCH("k_count");
    }
    k_scan_block<<<dim3(nb), dim3(256), 0, stream>>>(w.cnt, add_self_loops ? 1 : 0, N, (long long*)rowptr, w.block_sums);
    PPK_CHECK_LAUNCH("k_scan_block");
    k_scan_sums<<<dim3(1), dim3(256), 0, stream>>>(w.block_sums, nb, w.total);
    PPK_CHECK_LAUNCH("k_scan_sums");
    k_scan_add<<<dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, stream>>>((long long*)rowptr, w.block_sums, w.total, N);
    PPK_CHECK_LAUNCH("k_scan_add");
    k_fill<<<dim3(grid_for(E + N)), dim3(256), 0, stream>>>(src, dst, E, N, N_src, drop_self_loops, add_self_loops,
                                                            (const long long*)rowptr, w.cursor, (long long*)eid);
    PPK_CHECK_LAUNCH("k_fill");
    k_sort_short<<<dim3(grid_for(N)), dim3(256), 0, stream>>>((const long long*)rowptr, N, (long long*)eid, w.long_rows, w.n_long);
    PPK_CHECK_LAUNCH("k_sort_short");
    // the fill is over: `cursor` now queues the rows beyond one LDS image, counted in n_long[1]; `col` is not
    // written before k_finish and serves as the merge passes' second buffer
    int* vlong_rows = w.cursor;
    int* n_vlong = w.n_long + 1;
    k_sort_long<<<dim3(512), dim3(256), 0, stream>>>((const long long*)rowptr, (long long*)eid, w.long_rows, w.n_long, vlong_rows, n_vlong);
    PPK_CHECK_LAUNCH("k_sort_long");
    if (E + 1 > LDS_SORT_MAX) {
        k_sort_vlong_chunks<<<dim3(128, 16), dim3(256), 0, stream>>>((const long long*)rowptr, (long long*)eid, vlong_rows, n_vlong);
        PPK_CHECK_LAUNCH("k_sort_vlong_chunks");
    }
    int passes = 0;                                   // a row holds at most E + 1 entries
    while (((long long)LDS_SORT_MAX << passes) < E + 1) ++passes;
    for (int p = 0; p < passes; ++p) {
        k_merge_pass<<<dim3(128, 16), dim3(256), 0, stream>>>((const long long*)rowptr, (long long*)eid, (long long*)col, vlong_rows, n_vlong, p);
        PPK_CHECK_LAUNCH("k_merge_pass");
    }
    if (passes) {
        k_merge_finish<<<dim3(128, 16), dim3(256), 0, stream>>>((const long long*)rowptr, (long long*)eid, (const long long*)col, vlong_rows, n_vlong);
        PPK_CHECK_LAUNCH("k_merge_finish");
    }
    k_finish<<<dim3(grid_for(N)), dim3(256), 0, stream>>>(src, E, N, (const long long*)rowptr, (const long long*)eid, w.cnt,
                                                          add_self_loops ? 1 : 0, normalize, edge_weight, (long long*)col, val);
    PPK_CHECK_LAUNCH("k_finish");
    k_finish_long<<<dim3(8, 256), dim3(256), 0, stream>>>(src, E, (const long long*)rowptr, (const long long*)eid, w.cnt, add_self_loops ? 1 : 0,
                                                          normalize, edge_weight, w.long_rows, w.n_long, (long long*)col, val);
    PPK_CHECK_LAUNCH("k_finish_long");
    PPK_HIP(hipMemcpyAsync(status, w.bad, sizeof(int), hipMemcpyDeviceToDevice, stream));
    return PPK_OK;
}

// out[i, 0:F] = sum_{j in row i} val[j] * x[col[j], 0:F]   (rows in CSR order: deterministic)
int ppk_spmm_csr(const int64_t* rowptr, const int64_t* col, const float* val, int64_t N, int64_t nnz,
                 const float* x, int64_t ldx, int64_t F, float* out, int64_t ldo, hipStream_t stream)
{
    if (N < 1 || F < 1 || nnz < 0) PPK_FAIL(PPK_ERR_SHAPE, "spmm_csr: bad sizes");
    if (!rowptr || !col || !val || !x || !out) PPK_FAIL(PPK_ERR_NULL, "spmm_csr: null pointer");
    if (ldx < F || ldo < F) PPK_FAIL(PPK_ERR_SHAPE, "spmm_csr: leading dimension < F");
    const unsigned nb = (unsigned)((N + 3) / 4 < 65536 * 4 ? (N + 3) / 4 : 65536 * 4);
    // algorithmic bytes: one F-row + col + val per entry, one F-row + rowptr per node
    ppk::ProfScope prof(ppk::PK_SPMM, (double)nnz * (4.0 * F + 12.0) + (double)N * (4.0 * F + 8.0), stream);
    const bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && !((uintptr_t)x & 15) && !((uintptr_t)out & 15);
    if (vec) k_spmm<4><<<dim3(nb), dim3(256), 0, stream>>>((const long long*)rowptr, (const long long*)col, val, x, ldx, N, (int)F, out, ldo, nullptr, nullptr);
    else     k_spmm<1><<<dim3(nb), dim3(256), 0, stream>>>((const long long*)rowptr, (const long long*)col, val, x, ldx, N, (int)F, out, ldo, nullptr, nullptr);
    PPK_CHECK_LAUNCH("k_spmm");
    if (vec) k_spmm_long<4><<<dim3(512), dim3(1024), 0, stream>>>((const long long*)rowptr, (const long long*)col, val, x, ldx, N, (int)F, out, ldo);
    else     k_spmm_long<1><<<dim3(512), dim3(1024), 0, stream>>>((const long long*)rowptr, (const long long*)col, val, x, ldx, N, (int)F, out, ldo);
    PPK_CHECK_LAUNCH("k_spmm_long");
    return PPK_OK;
}

// The same product with a workspace: rows beyond 1024 entries are cut into 1024-entry chunks spread over the whole
// grid (ppk_spmm_csr gives each such row to ONE workgroup: 35 ms when three hubs hold a third of 5 M edges).
size_t ppk_spmm_csr_workspace_bytes(int64_t N, int64_t nnz, int64_t F)
{
    if (N < 1 || nnz < 0 || F < 1) return 0;
    const size_t slots = (size_t)(nnz / SPMM_LONG + 1), chunks = (size_t)(nnz / SPMM_CHUNK) + slots + 1;
    return 256 + ppk_align_up(sizeof(LongRow) * slots, 256) + ppk_align_up(sizeof(float) * chunks * (size_t)F, 256);
}

int ppk_spmm_csr_ws(const int64_t* rowptr, const int64_t* col, const float* val, int64_t N, int64_t nnz,
                    const float* x, int64_t ldx, int64_t F, float* out, int64_t ldo, void* workspace, size_t ws_bytes, hipStream_t stream)
{
    if (N < 1 || F < 1 || nnz < 0) PPK_FAIL(PPK_ERR_SHAPE, "spmm_csr_ws: bad sizes");
    if (!rowptr || !col || !val || !x || !out || !workspace) PPK_FAIL(PPK_ERR_NULL, "spmm_csr_ws: null pointer");
    if (ldx < F || ldo < F) PPK_FAIL(PPK_ERR_SHAPE, "spmm_csr_ws: leading dimension < F");
    if (ws_bytes < ppk_spmm_csr_workspace_bytes(N, nnz, F)) PPK_FAIL(PPK_ERR_WS, "spmm_csr_ws: workspace too small");
    char* cur = (char*)workspace;
    unsigned long long* ctr = (unsigned long long*)ppk_carve(cur, 256);
    LongRow* slots = (LongRow*)ppk_carve(cur, sizeof(LongRow) * (size_t)(nnz / SPMM_LONG + 1));
    float* part = (float*)cur;
    PPK_HIP(hipMemsetAsync(ctr, 0, sizeof(unsigned long long), stream));
    const unsigned nb = (unsigned)((N + 3) / 4 < 65536 * 4 ? (N + 3) / 4 : 65536 * 4);
    ppk::ProfScope prof(ppk::PK_SPMM, (double)nnz * (4.0 * F + 12.0) + (double)N * (4.0 * F + 8.0), stream);
    const bool vec = (F % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && !((uintptr_t)x & 15) && !((uintptr_t)out & 15);
    if (vec) k_spmm<4><<<dim3(nb), dim3(256), 0, stream>>>((const long long*)rowptr, (const long long*)col, val, x, ldx, N, (int)F, out, ldo, ctr, slots);
    else     k_spmm<1><<<dim3(nb), dim3(256), 0, stream>>>((const long long*)rowptr, (const long long*)col, val, x, ldx, N, (int)F, out, ldo, ctr, slots);
    PPK_CHECK_LAUNCH("k_spmm");
    if (vec) k_spmm_chunks<4><<<dim3(2048), dim3(256), 0, stream>>>((const long long*)rowptr, (const long long*)col, val, x, ldx, (int)F, ctr, slots, part);
    else     k_spmm_chunks<1><<<dim3(2048), dim3(256), 0, stream>>>((const long long*)rowptr, (const long long*)col, val, x, ldx, (int)F, ctr, slots, part);
    PPK_CHECK_LAUNCH("k_spmm_chunks");
    k_spmm_long_sum<<<dim3(256), dim3(256), 0, stream>>>((const long long*)rowptr, ctr, slots, part, (int)F, out, ldo);
    PPK_CHECK_LAUNCH("k_spmm_long_sum");
    return PPK_OK;
}

}  // extern "C"
