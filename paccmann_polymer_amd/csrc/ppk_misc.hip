// Small fused kernels around the GRU core: row gather / scatter, deterministic
// column sums, per-key row sums (embedding-table gradient), logits cross-entropy (K2), reparameterise +
// KL (K3), flat Adam.  Every reduction has a fixed order (no float atomics).
#include "ppk_common.h"
#include <mutex>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Scratch of ppk_colsum_noalloc: the per-chunk partial column sums that ppk_colsum keeps in its caller-provided
// workspace.  Every GPU the library runs on holds its own copy of this array until the process ends.  In host code
// the NAME is only a handle for hipGetSymbolAddress — never a device address: kernels get the per-device pointer.
constexpr size_t NOALLOC_FLOATS = (size_t)1 << 18;             // 1 MiB
constexpr int NOALLOC_MAX_DEVICES = 64;
static __device__ __align__(256) float scratch[NOALLOC_FLOATS];
float* g_scratch_ptr[NOALLOC_MAX_DEVICES];
std::mutex g_scratch_mutex;

inline unsigned grid_for(long long work, unsigned cap = 4096)
{
    long long b = (work + 255) / 256;
    if (b < 1) b = 1;
    return (unsigned)(b < cap ? b : cap);
}

// dst[i, :] = src[idx[i], :]   (gather)      or   dst[idx[i], :] = src[i, :]   (scatter, distinct idx)
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_rows(const float* __restrict__ src, long long lds_, const int64_t* __restrict__ idx,
                                              long long n_rows, int width, long long src_rows, float* __restrict__ dst, long long ldd,
                                              long long dst_rows)
{
    const long long total = n_rows * width;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long i = e / width;
        const int c = (int)(e % width);
        const long long j = idx[i];
        if (SCATTER) {
            if (j >= 0 && j < dst_rows) dst[j * ldd + c] = src[i * lds_ + c];
        } else {
            dst[i * ldd + c] = (j >= 0 && j < src_rows) ? src[j * lds_ + c] : 0.0f;
        }
    }
}

// ---- column sums: out[c] = sum_r A[r, c] ; two passes, fixed order ----------
// pass 1: a block sums CS_CHUNK rows of 256 columns, 8 independent row streams per thread
// (HBM-bound read of A); pass 2: 32 columns per block, 8 slices of the partials per column, 8 loads
// in flight per thread, summed in double and combined in a fixed order.
constexpr int CS_CHUNK = 512;   // rows per partial
__global__ __launch_bounds__(256) void k_colsum_part(const float* __restrict__ A, long long lda, long long R, int C,
                                                     float* __restrict__ part)
{
    const long long r0 = (long long)blockIdx.y * CS_CHUNK;
    const long long r1 = r0 + CS_CHUNK < R ? r0 + CS_CHUNK : R;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    long long r = r0;
    for (; r + 8 <= r1; r += 8)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += A[(r + k) * lda + c];
    for (; r < r1; ++r) acc[0] += A[r * lda + c];
    part[(long long)blockIdx.y * C + c] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}
__global__ __launch_bounds__(256) void k_colsum_final(const float* __restrict__ part, int nparts, int C, float beta,
                                                      float* __restrict__ out)
{
    __shared__ double red[8][32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int per = (nparts + 7) / 8;
    const int p0 = sl * per, p1 = p0 + per < nparts ? p0 + per : nparts;
    double acc = 0.0;
    if (c < C) {
        int p = p0;
        for (; p + 8 <= p1; p += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = part[(long long)(p + k) * C + c];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += (double)v[k];
        }
        for (; p < p1; ++p) acc += (double)part[(long long)p * C + c];
    }
    red[sl][cl] = acc;
    __syncthreads();
    if (sl == 0 && c < C) {
        double t = red[0][cl];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][cl];
        out[c] = (beta != 0.f ? beta * out[c] : 0.f) + (float)t;
    }
}

// ---- rows of a table summed per key, in ascending row order (the embedding-table gradient of K1) ----------------
// out[k, :] = sum over the rows r with key[r] == k of x[r, :], added in ascending r inside fixed 64-row chunks and
// the chunks of a key in ascending order: one fixed association whatever the launch geometry (bitwise reproducible).
// Sort-free: a stable counting sort (per-1024-row-block key histograms in LDS -> per-key prefix over the blocks ->
// exclusive scan over the keys -> stable rank inside the block) lists every key's rows; a key's list is cut into
// chunks of RBK_CHUNK rows, one workgroup per (chunk, 1024-column block) gathers its rows with 16-byte loads (eight in
// flight per lane), and a last pass adds a key's chunk partials in order.
constexpr int RBK_BLOCK = 1024;       // rows per histogram block
constexpr int RBK_CHUNK = 64;         // rows per partial sum
constexpr int RBK_MAX_KEYS = 8192;    // histogram bins kept in LDS

__global__ __launch_bounds__(256) void k_rbk_hist(const int64_t* __restrict__ key, long long n, int nkeys, int* __restrict__ hist, int* __restrict__ bad)
{
    __shared__ int h[RBK_MAX_KEYS];
    for (int k = threadIdx.x; k < nkeys; k += 256) h[k] = 0;
    __syncthreads();
    const long long r0 = (long long)blockIdx.x * RBK_BLOCK;
    for (int i = threadIdx.x; i < RBK_BLOCK && r0 + i < n; i += 256) {
        const long long k = key[r0 + i];
        if (k < 0 || k >= nkeys) { atomicOr(bad, 1); continue; }
        atomicAdd(&h[(int)k], 1);                     // integer LDS atomics: order-independent result
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nkeys; k += 256) hist[(long long)blockIdx.x * nkeys + k] = h[k];
}
// per key: exclusive prefix of its block counts (in place), total -> cnt[k]
__global__ __launch_bounds__(256) void k_rbk_prefix(int* __restrict__ hist, int nblk, int nkeys, int* __restrict__ cnt)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nkeys) return;
    int run = 0;
    for (int b = 0; b < nblk; ++b) { const int c = hist[(long long)b * nkeys + k]; hist[(long long)b * nkeys + k] = run; run += c; }
    cnt[k] = run;
}
// one workgroup: rowptr = exclusive scan of cnt, chunkptr = exclusive scan of ceil(cnt / RBK_CHUNK)
__global__ __launch_bounds__(256) void k_rbk_scan(const int* __restrict__ cnt, int nkeys, long long* __restrict__ rowptr, int* __restrict__ chunkptr)
{
    __shared__ long long sr[256];
    __shared__ int sc[256];
    const int per = (nkeys + 255) / 256, k0 = threadIdx.x * per, k1 = k0 + per < nkeys ? k0 + per : nkeys;
    long long r = 0;
    int c = 0;
    for (int k = k0; k < k1; ++k) { r += cnt[k]; c += (cnt[k] + RBK_CHUNK - 1) / RBK_CHUNK; }
    sr[threadIdx.x] = r; sc[threadIdx.x] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long rr = 0;
        int cc = 0;
        for (int t = 0; t < 256; ++t) { const long long a = sr[t]; const int b = sc[t]; sr[t] = rr; sc[t] = cc; rr += a; cc += b; }
        rowptr[nkeys] = rr; chunkptr[nkeys] = cc;
    }
    __syncthreads();
    r = sr[threadIdx.x]; c = sc[threadIdx.x];
    for (int k = k0; k < k1; ++k) { rowptr[k] = r; chunkptr[k] = c; r += cnt[k]; c += (cnt[k] + RBK_CHUNK - 1) / RBK_CHUNK; }
}
// list[rowptr[k] + (rows of key k in earlier blocks) + (earlier rows of key k in this block)] = row
__global__ __launch_bounds__(256) void k_rbk_fill(const int64_t* __restrict__ key, long long n, int nkeys, const int* __restrict__ hist,
                                                  const long long* __restrict__ rowptr, int* __restrict__ list)
{
    __shared__ int kk[RBK_BLOCK];
    const long long r0 = (long long)blockIdx.x * RBK_BLOCK;
    const int m = (int)(n - r0 < RBK_BLOCK ? n - r0 : RBK_BLOCK);
    for (int i = threadIdx.x; i < m; i += 256) { const long long k = key[r0 + i]; kk[i] = (k < 0 || k >= nkeys) ? -1 : (int)k; }
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += 256) {
        const int k = kk[i];
        if (k < 0) continue;
        int rank = 0;
        for (int j = 0; j < i; ++j) rank += kk[j] == k;
        list[rowptr[k] + hist[(long long)blockIdx.x * nkeys + k] + rank] = (int)(r0 + i);
    }
}
// grid (chunk, column block of 1024): part[chunk][cols] = sum of the chunk's rows, ascending
__global__ __launch_bounds__(256) void k_rbk_partial(const long long* __restrict__ rowptr, const int* __restrict__ chunkptr, int nkeys,
                                                     const int* __restrict__ list, const float* __restrict__ x, long long ldx, int F,
                                                     float* __restrict__ part)
{
    const int c = blockIdx.x;
    if (c >= chunkptr[nkeys]) return;                 // the grid is an upper bound (the chunk count lives on the device)
    int lo = 0, hi = nkeys;                           // key of this chunk: last k with chunkptr[k] <= c
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (chunkptr[mid] <= c) lo = mid; else hi = mid; }
    const long long b = rowptr[lo] + (long long)(c - chunkptr[lo]) * RBK_CHUNK;
    const long long e = b + RBK_CHUNK < rowptr[lo + 1] ? b + RBK_CHUNK : rowptr[lo + 1];
    const int col = blockIdx.y * 1024 + threadIdx.x * 4;
    if (col >= F) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    long long i = b;
    for (; i + 8 <= e; i += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + (long long)list[i + u] * ldx + col);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; i < e; ++i) acc += *reinterpret_cast<const f32x4*>(x + (long long)list[i] * ldx + col);
    *reinterpret_cast<f32x4*>(part + (long long)c * F + col) = acc;
}
// grid (key, column block of 1024): out[key] = sum of its chunk partials, ascending
__global__ __launch_bounds__(256) void k_rbk_final(const int* __restrict__ chunkptr, const float* __restrict__ part, int F,
                                                   float* __restrict__ out, long long ldo)
{
    const int k = blockIdx.x, col = blockIdx.y * 1024 + threadIdx.x * 4;
    if (col >= F) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int c = chunkptr[k]; c < chunkptr[k + 1]; ++c) acc += *reinterpret_cast<const f32x4*>(part + (long long)c * F + col);
    *reinterpret_cast<f32x4*>(out + (long long)k * ldo + col) = acc;
}

// ---- dropout factors: Philox4x32-10, counter = (row lo, row hi, unit / 4, layer), key = seed ----------------------
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__global__ __launch_bounds__(256) void k_dropout_factors(unsigned long long seed, float p, long long layers, long long NT, int H, int Hv,
                                                         float* __restrict__ out)
{
    const int q = H >> 2;                              // groups of four units per row
    const long long total = layers * NT * q;
    const float scale = 1.0f / (1.0f - p);
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long lr = e / q;
        const int u4 = (int)(e - lr * q);
        const long long l = lr / NT, row = lr - l * NT;
        unsigned c[4] = {(unsigned)row, (unsigned)(row >> 32), (unsigned)u4, (unsigned)l};
        philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float uni = (float)(c[k] >> 8) * (1.0f / 16777216.0f);          // [0, 1)
            v[k] = (4 * u4 + k < Hv && uni >= p) ? scale : 0.f;
        }
        *reinterpret_cast<f32x4*>(out + e * 4) = v;
    }
}

// ---- K2: cross-entropy over packed logits, ignore_index, mean over valid ----
// one wave per row; block of 256 = 4 rows.  Writes lse[row] and per-block partials.
__global__ __launch_bounds__(256) void k_ce_rows(const float* __restrict__ logits, long long ldl, const int64_t* __restrict__ tgt,
                                                 long long rows, int V, long long ignore_index,
                                                 float* __restrict__ lse, double* __restrict__ part_loss,
                                                 unsigned int* __restrict__ part_cnt)
{
    __shared__ double sl[4];
    __shared__ unsigned int sc[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double myloss = 0.0;
    unsigned int mycnt = 0;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows; row += (long long)gridDim.x * 4) {
        const float* x = logits + row * ldl;
        float m = -INFINITY;
        for (int c = lane; c < V; c += 64) m = fmaxf(m, x[c]);
        m = ppk_wave_max(m);
        float s = 0.f;
        for (int c = lane; c < V; c += 64) s += expf(x[c] - m);
        s = ppk_wave_sum(s);
        const float l = m + logf(s);
        if (lane == 0) {
            lse[row] = l;
            const long long t = tgt[row];
            if (t != ignore_index && t >= 0 && t < V) {
                myloss += (double)(l - x[t]);
                mycnt += 1;
            }
        }
    }
    if (lane == 0) { sl[wave] = myloss; sc[wave] = mycnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part_loss[blockIdx.x] = (sl[0] + sl[1]) + (sl[2] + sl[3]);
        part_cnt[blockIdx.x] = sc[0] + sc[1] + sc[2] + sc[3];
    }
}
__global__ __launch_bounds__(256) void k_ce_final(const double* __restrict__ part_loss, const unsigned int* __restrict__ part_cnt,
                                                  int nparts, float* __restrict__ loss, float* __restrict__ count)
{
    __shared__ double sl[4];
    __shared__ unsigned int sc[4];
    double a = 0.0;
    unsigned int c = 0;
    for (int p = threadIdx.x; p < nparts; p += 256) { a += part_loss[p]; c += part_cnt[p]; }
    a = ppk_wave_sum_d(a);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) { sl[threadIdx.x >> 6] = a; sc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = (sl[0] + sl[1]) + (sl[2] + sl[3]);
        const unsigned int n = sc[0] + sc[1] + sc[2] + sc[3];
        loss[0] = (float)(tot / (double)n);     // n == 0 -> nan, like torch
        count[0] = (float)n;
    }
}
__global__ __launch_bounds__(256) void k_ce_bwd(const float* __restrict__ logits, long long ldl, const int64_t* __restrict__ tgt,
                                                long long rows, int V, long long ignore_index, const float* __restrict__ lse,
                                                const float* __restrict__ count, const float* __restrict__ dloss,
                                                float* __restrict__ dlogits, long long ldd)
{
    const float scale = dloss[0] / count[0];
    const long long total = rows * V;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long row = e / V;
        const int c = (int)(e % V);
        const long long t = tgt[row];
        float g = 0.f;
        if (t != ignore_index && t >= 0 && t < V) g = scale * (expf(logits[row * ldl + c] - lse[row]) - (c == t ? 1.f : 0.f));
        dlogits[row * ldd + c] = g;
    }
}

// ---- K3: reparameterise + KL ---------------------------------------------------
__global__ __launch_bounds__(256) void k_reparam_fwd(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                     const float* __restrict__ eps, long long n, float* __restrict__ z,
                                                     double* __restrict__ part)
{
    __shared__ double sl[4];
    double acc = 0.0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const float m = mu[e], lv = logvar[e];
        const float sd = expf(0.5f * lv);
        z[e] = fmaf(eps[e], sd, m);
        acc += (double)(1.0f + lv - m * m - expf(lv));
    }
    acc = ppk_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sl[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sl[0] + sl[1]) + (sl[2] + sl[3]);
}
__global__ __launch_bounds__(64) void k_kl_final(const double* __restrict__ part, int nparts, float* __restrict__ kl)
{
    double a = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 64) a += part[p];
    a = ppk_wave_sum_d(a);
    if (threadIdx.x == 0) kl[0] = (float)(-0.5 * a);
}
__global__ __launch_bounds__(256) void k_reparam_bwd(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                     const float* __restrict__ eps, long long n, const float* __restrict__ dz,
                                                     const float* __restrict__ dkl, float* __restrict__ dmu, float* __restrict__ dlogvar)
{
    const float gk = dkl ? dkl[0] : 0.f;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const float m = mu[e], lv = logvar[e];
        const float g = dz ? dz[e] : 0.f;
        dmu[e] = g + gk * m;
        dlogvar[e] = g * eps[e] * 0.5f * expf(0.5f * lv) + gk * 0.5f * (expf(lv) - 1.0f);
    }
}

// ---- flat Adam (torch.optim.Adam semantics, no amsgrad) --------------------------
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                              float wd, float bc1, float bc2_sqrt)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        float gr = g[e];
        const float pe = p[e];
        if (wd != 0.f) gr = fmaf(wd, pe, gr);
        const float me = b1 * m[e] + (1.0f - b1) * gr;
        const float ve = b2 * v[e] + (1.0f - b2) * gr * gr;
        m[e] = me;
        v[e] = ve;
        const float denom = sqrtf(ve) / bc2_sqrt + eps;
        p[e] = pe - (lr / bc1) * (me / denom);
    }
}

// ---- PReLU (per-channel slope), reactions_infomax.py:61,67 ----------------------
__global__ __launch_bounds__(256) void k_prelu_fwd(const float* __restrict__ x, const float* __restrict__ alpha, long long rows,
                                                   int C, float* __restrict__ y)
{
    const long long total = rows * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const float v = x[e];
        y[e] = v >= 0.f ? v : alpha[e % C] * v;
    }
}
// dx = dy * (x >= 0 ? 1 : alpha) ; t = (x < 0 ? x * dy : 0)  (column sums of t give dalpha)
__global__ __launch_bounds__(256) void k_prelu_bwd(const float* __restrict__ x, const float* __restrict__ alpha,
                                                   const float* __restrict__ dy, long long rows, int C,
                                                   float* __restrict__ dx, float* __restrict__ t)
{
    const long long total = rows * C;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const float v = x[e], g = dy[e];
        dx[e] = v >= 0.f ? g : alpha[e % C] * g;
        t[e] = v < 0.f ? v * g : 0.f;
    }
}

// ---- DeepGraphInfomax loss on discriminator logits (Appendix C) -----------------
// loss = -mean log(sigmoid(p) + 1e-15) - mean log(1 - sigmoid(n) + 1e-15)
__global__ __launch_bounds__(256) void k_dgi_part(const float* __restrict__ p, const float* __restrict__ n, long long N,
                                                  double* __restrict__ part)
{
    __shared__ double sl[4];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
        const float sp = 1.0f / (1.0f + expf(-p[i])), sn = 1.0f / (1.0f + expf(-n[i]));
        acc -= (double)logf(sp + 1e-15f);
        acc -= (double)logf(1.0f - sn + 1e-15f);
    }
    acc = ppk_wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sl[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sl[0] + sl[1]) + (sl[2] + sl[3]);
}
__global__ __launch_bounds__(64) void k_dgi_final(const double* __restrict__ part, int nparts, long long N, float* __restrict__ loss)
{
    double a = 0.0;
    for (int q = threadIdx.x; q < nparts; q += 64) a += part[q];
    a = ppk_wave_sum_d(a);
    if (threadIdx.x == 0) loss[0] = (float)(a / (double)N);
}
__global__ __launch_bounds__(256) void k_dgi_bwd(const float* __restrict__ p, const float* __restrict__ n, long long N,
                                                 const float* __restrict__ dloss, float* __restrict__ dp, float* __restrict__ dn)
{
    const float g = dloss[0] / (float)N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
        const float sp = 1.0f / (1.0f + expf(-p[i])), sn = 1.0f / (1.0f + expf(-n[i]));
        dp[i] = -g * sp * (1.0f - sp) / (sp + 1e-15f);
        dn[i] = g * sn * (1.0f - sn) / (1.0f - sn + 1e-15f);
    }
}

}  // namespace

extern "C" {

int ppk_gather_rows(const float* src, int64_t ld_src, int64_t src_rows, const int64_t* idx, int64_t n_rows, int64_t width,
                    float* dst, int64_t ld_dst, hipStream_t stream)
{
    if (n_rows == 0 || width == 0) return PPK_OK;
    if (!src || !idx || !dst) PPK_FAIL(PPK_ERR_NULL, "gather_rows: null pointer");
    if (n_rows < 0 || width < 0 || ld_src < width || ld_dst < width) PPK_FAIL(PPK_ERR_SHAPE, "gather_rows: bad sizes");
    k_rows<false><<<dim3(grid_for(n_rows * width)), dim3(256), 0, stream>>>(src, ld_src, idx, n_rows, (int)width, src_rows, dst, ld_dst, 0);
    PPK_CHECK_LAUNCH("k_rows<gather>");
    return PPK_OK;
}

int ppk_scatter_rows(const float* src, int64_t ld_src, const int64_t* idx, int64_t n_rows, int64_t width,
                     float* dst, int64_t ld_dst, int64_t dst_rows, hipStream_t stream)
{
    if (n_rows == 0 || width == 0) return PPK_OK;
    if (!src || !idx || !dst) PPK_FAIL(PPK_ERR_NULL, "scatter_rows: null pointer");
    if (n_rows < 0 || width < 0 || ld_src < width || ld_dst < width) PPK_FAIL(PPK_ERR_SHAPE, "scatter_rows: bad sizes");
    k_rows<true><<<dim3(grid_for(n_rows * width)), dim3(256), 0, stream>>>(src, ld_src, idx, n_rows, (int)width, 0, dst, ld_dst, dst_rows);
    PPK_CHECK_LAUNCH("k_rows<scatter>");
    return PPK_OK;
}

size_t ppk_colsum_workspace_bytes(int64_t rows, int64_t cols)
{
    if (rows <= 0 || cols <= 0) return 0;
    return ppk_align_up(sizeof(float) * (size_t)((rows + CS_CHUNK - 1) / CS_CHUNK) * (size_t)cols, 256);
}

// out[c] = beta * out[c] + sum_r A[r, c]
int ppk_colsum(const float* A, int64_t lda, int64_t rows, int64_t cols, float beta, float* out,
               void* workspace, size_t ws_bytes, hipStream_t stream)
{
    if (cols == 0) return PPK_OK;
    if (!A || !out || !workspace) PPK_FAIL(PPK_ERR_NULL, "colsum: null pointer");
    if (rows < 1 || cols < 1 || lda < cols) PPK_FAIL(PPK_ERR_SHAPE, "colsum: bad sizes");
    if (ws_bytes < ppk_colsum_workspace_bytes(rows, cols)) PPK_FAIL(PPK_ERR_WS, "colsum: workspace too small");
    const int nparts = (int)((rows + CS_CHUNK - 1) / CS_CHUNK);
    if (nparts > 65535) PPK_FAIL(PPK_ERR_UNSUP, "colsum: too many rows");
    dim3 grid((unsigned)((cols + 255) / 256), (unsigned)nparts);
    k_colsum_part<<<grid, dim3(256), 0, stream>>>(A, lda, rows, (int)cols, (float*)workspace);
    PPK_CHECK_LAUNCH("k_colsum_part");
    k_colsum_final<<<dim3((unsigned)((cols + 31) / 32)), dim3(256), 0, stream>>>((const float*)workspace, nparts, (int)cols, beta, out);
    PPK_CHECK_LAUNCH("k_colsum_final");
    return PPK_OK;
}

// ppk_colsum for callers that cannot allocate (HIP graph capture, serving loops): the partials live in the
// library's own file-scope device array, reached through the address hipGetSymbolAddress returns for the CURRENT
// device (looked up once per device).  No hipMalloc / hipFree / allocating call on this path.  All calls on one GPU
// share that GPU's array: calls that overlap on one GPU (two streams, two threads) race.  Nothing has to start at
// zero — pass 1 overwrites every partial pass 2 reads — so a call does not depend on what the last one left.
int ppk_colsum_noalloc(const float* A, int64_t lda, int64_t rows, int64_t cols, float beta, float* out, hipStream_t stream)
{
    if (cols == 0) return PPK_OK;
    if (!A || !out) PPK_FAIL(PPK_ERR_NULL, "colsum_noalloc: null pointer");
    if (rows < 1 || cols < 1 || lda < cols) PPK_FAIL(PPK_ERR_SHAPE, "colsum_noalloc: bad sizes");
    const int64_t nparts = (rows + CS_CHUNK - 1) / CS_CHUNK;
    if (nparts > 65535 || (uint64_t)nparts * (uint64_t)cols > NOALLOC_FLOATS)
        PPK_FAIL(PPK_ERR_UNSUP, "colsum_noalloc: ceil(rows / %d) * cols = %lld exceeds the built-in scratch of %zu floats (use ppk_colsum with a workspace)",
                 CS_CHUNK, (long long)(nparts * cols), NOALLOC_FLOATS);
    int dev = -1;
    PPK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) PPK_FAIL(PPK_ERR_UNSUP, "colsum_noalloc: device index %d not supported", dev);
    float* part = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[dev]) {
            void* sym = nullptr;
            PPK_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)));        // this device's copy of the array
            if (!sym) PPK_FAIL(PPK_ERR_HIP, "colsum_noalloc: hipGetSymbolAddress returned null");
            g_scratch_ptr[dev] = (float*)sym;
        }
        part = g_scratch_ptr[dev];
    }
    dim3 grid((unsigned)((cols + 255) / 256), (unsigned)nparts);
    k_colsum_part<<<grid, dim3(256), 0, stream>>>(A, lda, rows, (int)cols, part);
    PPK_CHECK_LAUNCH("k_colsum_part");
    k_colsum_final<<<dim3((unsigned)((cols + 31) / 32)), dim3(256), 0, stream>>>(part, (int)nparts, (int)cols, beta, out);
    PPK_CHECK_LAUNCH("k_colsum_final");
    return PPK_OK;
}

size_t ppk_rowsum_by_key_workspace_bytes(int64_t n_rows, int64_t n_keys, int64_t F)
{
    if (n_rows < 1 || n_keys < 1 || F < 1) return 0;
    const size_t nblk = (size_t)((n_rows + RBK_BLOCK - 1) / RBK_BLOCK), chunks = (size_t)(n_rows / RBK_CHUNK + n_keys);
    return ppk_align_up(sizeof(int) * nblk * (size_t)n_keys, 256) + ppk_align_up(sizeof(int) * (size_t)n_keys, 256)
           + ppk_align_up(sizeof(long long) * (size_t)(n_keys + 1), 256) + ppk_align_up(sizeof(int) * (size_t)(n_keys + 1), 256)
           + ppk_align_up(sizeof(int) * (size_t)n_rows, 256) + ppk_align_up(sizeof(float) * chunks * (size_t)F, 256) + 256;
}

int ppk_rowsum_by_key(const int64_t* keys, int64_t n_rows, int64_t n_keys, const float* x, int64_t ldx, int64_t F,
                      float* out, int64_t ldo, int* status, void* workspace, size_t ws_bytes, hipStream_t stream)
{
    if (!keys || !x || !out || !status || !workspace) PPK_FAIL(PPK_ERR_NULL, "rowsum_by_key: null pointer");
    if (n_rows < 1 || n_keys < 1 || F < 1 || ldx < F || ldo < F) PPK_FAIL(PPK_ERR_SHAPE, "rowsum_by_key: bad sizes");
    if (n_keys > RBK_MAX_KEYS || n_rows > 0x7fffffff) PPK_FAIL(PPK_ERR_UNSUP, "rowsum_by_key: more than %d keys or 2^31 rows", RBK_MAX_KEYS);
    if ((F & 3) || (ldx & 3) || (ldo & 3) || ((uintptr_t)x & 15) || ((uintptr_t)out & 15)) PPK_FAIL(PPK_ERR_ALIGN, "rowsum_by_key: F, ldx, ldo must be multiples of 4 and the tables 16-byte aligned");
    if (ws_bytes < ppk_rowsum_by_key_workspace_bytes(n_rows, n_keys, F)) PPK_FAIL(PPK_ERR_WS, "rowsum_by_key: workspace too small");
    const int nblk = (int)((n_rows + RBK_BLOCK - 1) / RBK_BLOCK), nk = (int)n_keys;
    const long long chunks = n_rows / RBK_CHUNK + n_keys;
    char* cur = (char*)workspace;
    int* hist = (int*)ppk_carve(cur, sizeof(int) * (size_t)nblk * (size_t)nk);
    int* cnt = (int*)ppk_carve(cur, sizeof(int) * (size_t)nk);
    long long* rowptr = (long long*)ppk_carve(cur, sizeof(long long) * (size_t)(nk + 1));
    int* chunkptr = (int*)ppk_carve(cur, sizeof(int) * (size_t)(nk + 1));
    int* list = (int*)ppk_carve(cur, sizeof(int) * (size_t)n_rows);
    float* part = (float*)ppk_carve(cur, sizeof(float) * (size_t)chunks * (size_t)F);
    int* bad = (int*)ppk_carve(cur, 256);
    PPK_HIP(hipMemsetAsync(bad, 0, 256, stream));
    k_rbk_hist<<<dim3(nblk), dim3(256), 0, stream>>>(keys, n_rows, nk, hist, bad);
    PPK_CHECK_LAUNCH("k_rbk_hist");
    k_rbk_prefix<<<dim3((nk + 255) / 256), dim3(256), 0, stream>>>(hist, nblk, nk, cnt);
    PPK_CHECK_LAUNCH("k_rbk_prefix");
    k_rbk_scan<<<dim3(1), dim3(256), 0, stream>>>(cnt, nk, rowptr, chunkptr);
    PPK_CHECK_LAUNCH("k_rbk_scan");
    k_rbk_fill<<<dim3(nblk), dim3(256), 0, stream>>>(keys, n_rows, nk, hist, rowptr, list);
    PPK_CHECK_LAUNCH("k_rbk_fill");
    const unsigned cb = (unsigned)((F + 1023) / 1024);
    k_rbk_partial<<<dim3((unsigned)chunks, cb), dim3(256), 0, stream>>>(rowptr, chunkptr, nk, list, x, ldx, (int)F, part);
    PPK_CHECK_LAUNCH("k_rbk_partial");
    k_rbk_final<<<dim3((unsigned)nk, cb), dim3(256), 0, stream>>>(chunkptr, part, (int)F, out, ldo);
    PPK_CHECK_LAUNCH("k_rbk_final");
    PPK_HIP(hipMemcpyAsync(status, bad, sizeof(int), hipMemcpyDeviceToDevice, stream));
    return PPK_OK;
}

int ppk_dropout_factors(uint64_t seed, float p, int64_t layers, int64_t NT, int64_t H, int64_t H_valid, float* out, hipStream_t stream)
{
    if (!out) PPK_FAIL(PPK_ERR_NULL, "dropout_factors: null pointer");
    if (layers < 1 || NT < 1 || H < 4 || (H & 3) || H_valid < 1 || H_valid > H || !(p >= 0.f && p < 1.f)) PPK_FAIL(PPK_ERR_SHAPE, "dropout_factors: bad sizes or p outside [0, 1)");
    if ((uintptr_t)out & 15) PPK_FAIL(PPK_ERR_ALIGN, "dropout_factors: out must be 16-byte aligned");
    k_dropout_factors<<<dim3(grid_for(layers * NT * (H / 4), 8192)), dim3(256), 0, stream>>>((unsigned long long)seed, p, layers, NT, (int)H, (int)H_valid, out);
    PPK_CHECK_LAUNCH("k_dropout_factors");
    return PPK_OK;
}

static int ce_blocks(int64_t rows) { long long b = (rows + 3) / 4; return (int)(b < 2048 ? (b < 1 ? 1 : b) : 2048); }

size_t ppk_ce_workspace_bytes(int64_t rows)
{
    if (rows <= 0) return 0;
    return ppk_align_up(sizeof(float) * (size_t)rows, 256) + ppk_align_up(sizeof(double) * 2048, 256) + ppk_align_up(sizeof(unsigned) * 2048, 256) + 256;
}

// loss[0] = mean over rows with target != ignore_index of (logsumexp(logits[row]) - logits[row, target])
// The workspace keeps lse / count for ppk_ce_bwd.
int ppk_ce_fwd(const float* logits, int64_t ldl, const int64_t* targets, int64_t rows, int64_t V, int64_t ignore_index,
               float* loss, void* workspace, size_t ws_bytes, hipStream_t stream)
{
    if (!logits || !targets || !loss || !workspace) PPK_FAIL(PPK_ERR_NULL, "ce_fwd: null pointer");
    if (rows < 1 || V < 1 || ldl < V) PPK_FAIL(PPK_ERR_SHAPE, "ce_fwd: bad sizes");
    if (ws_bytes < ppk_ce_workspace_bytes(rows)) PPK_FAIL(PPK_ERR_WS, "ce_fwd: workspace too small");
    char* cur = (char*)workspace;
    float* lse = (float*)ppk_carve(cur, sizeof(float) * (size_t)rows);
    double* pl = (double*)ppk_carve(cur, sizeof(double) * 2048);
    unsigned* pc = (unsigned*)ppk_carve(cur, sizeof(unsigned) * 2048);
    float* count = (float*)ppk_carve(cur, 256);
    const int nb = ce_blocks(rows);
    k_ce_rows<<<dim3(nb), dim3(256), 0, stream>>>(logits, ldl, targets, rows, (int)V, ignore_index, lse, pl, pc);
    PPK_CHECK_LAUNCH("k_ce_rows");
    k_ce_final<<<dim3(1), dim3(256), 0, stream>>>(pl, pc, nb, loss, count);
    PPK_CHECK_LAUNCH("k_ce_final");
    return PPK_OK;
}

int ppk_ce_bwd(const float* logits, int64_t ldl, const int64_t* targets, int64_t rows, int64_t V, int64_t ignore_index,
               const float* dloss, float* dlogits, int64_t ldd, const void* workspace, size_t ws_bytes, hipStream_t stream)
{
    if (!logits || !targets || !dloss || !dlogits || !workspace) PPK_FAIL(PPK_ERR_NULL, "ce_bwd: null pointer");
    if (rows < 1 || V < 1 || ldl < V || ldd < V) PPK_FAIL(PPK_ERR_SHAPE, "ce_bwd: bad sizes");
    if (ws_bytes < ppk_ce_workspace_bytes(rows)) PPK_FAIL(PPK_ERR_WS, "ce_bwd: workspace too small");
    char* cur = (char*)workspace;
    const float* lse = (const float*)ppk_carve(cur, sizeof(float) * (size_t)rows);
    (void)ppk_carve(cur, sizeof(double) * 2048);
    (void)ppk_carve(cur, sizeof(unsigned) * 2048);
    const float* count = (const float*)ppk_carve(cur, 256);
    k_ce_bwd<<<dim3(grid_for(rows * V)), dim3(256), 0, stream>>>(logits, ldl, targets, rows, (int)V, ignore_index, lse, count, dloss, dlogits, ldd);
    PPK_CHECK_LAUNCH("k_ce_bwd");
    return PPK_OK;
}

size_t ppk_reparam_workspace_bytes(int64_t n)
{
    (void)n;
    return ppk_align_up(sizeof(double) * 1024, 256);
}

// z = eps * exp(0.5 logvar) + mu ;  kl[0] = -0.5 sum(1 + logvar - mu^2 - exp(logvar))   (kl may be NULL)
int ppk_reparam_fwd(const float* mu, const float* logvar, const float* eps, int64_t n, float* z, float* kl,
                    void* workspace, size_t ws_bytes, hipStream_t stream)
{
    if (!mu || !logvar || !eps || !z || !workspace) PPK_FAIL(PPK_ERR_NULL, "reparam_fwd: null pointer");
    if (n < 1) PPK_FAIL(PPK_ERR_SHAPE, "reparam_fwd: n < 1");
    if (ws_bytes < ppk_reparam_workspace_bytes(n)) PPK_FAIL(PPK_ERR_WS, "reparam_fwd: workspace too small");
    const unsigned nb = grid_for(n, 1024);
    k_reparam_fwd<<<dim3(nb), dim3(256), 0, stream>>>(mu, logvar, eps, n, z, (double*)workspace);
    PPK_CHECK_LAUNCH("k_reparam_fwd");
    if (kl) {
        k_kl_final<<<dim3(1), dim3(64), 0, stream>>>((const double*)workspace, (int)nb, kl);
        PPK_CHECK_LAUNCH("k_kl_final");
    }
    return PPK_OK;
}

// dmu = dz + dkl * mu ; dlogvar = dz * eps * 0.5 exp(0.5 logvar) + dkl * 0.5 (exp(logvar) - 1)   (dz or dkl may be NULL)
int ppk_reparam_bwd(const float* mu, const float* logvar, const float* eps, int64_t n, const float* dz, const float* dkl,
                    float* dmu, float* dlogvar, hipStream_t stream)
{
    if (!mu || !logvar || !eps || !dmu || !dlogvar) PPK_FAIL(PPK_ERR_NULL, "reparam_bwd: null pointer");
    if (n < 1) PPK_FAIL(PPK_ERR_SHAPE, "reparam_bwd: n < 1");
    k_reparam_bwd<<<dim3(grid_for(n)), dim3(256), 0, stream>>>(mu, logvar, eps, n, dz, dkl, dmu, dlogvar);
    PPK_CHECK_LAUNCH("k_reparam_bwd");
    return PPK_OK;
}

// One Adam step over flat fp32 buffers (torch.optim.Adam defaults semantics; step is 1-based).
int ppk_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, hipStream_t stream)
{
    if (n == 0) return PPK_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) PPK_FAIL(PPK_ERR_NULL, "adam_step: null pointer");
    if (n < 0 || step < 1) PPK_FAIL(PPK_ERR_SHAPE, "adam_step: bad n / step");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    k_adam<<<dim3(grid_for(n)), dim3(256), 0, stream>>>(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                                                        (float)bc1, (float)sqrt(bc2));
    PPK_CHECK_LAUNCH("k_adam");
    return PPK_OK;
}

int ppk_prelu_fwd(const float* x, const float* alpha, int64_t rows, int64_t C, float* y, hipStream_t stream)
{
    if (!x || !alpha || !y) PPK_FAIL(PPK_ERR_NULL, "prelu_fwd: null pointer");
    if (rows < 1 || C < 1) PPK_FAIL(PPK_ERR_SHAPE, "prelu_fwd: bad sizes");
    k_prelu_fwd<<<dim3(grid_for(rows * C)), dim3(256), 0, stream>>>(x, alpha, rows, (int)C, y);
    PPK_CHECK_LAUNCH("k_prelu_fwd");
    return PPK_OK;
}

// dx [rows, C]; t [rows, C] scratch whose column sums (ppk_colsum) are dalpha
int ppk_prelu_bwd(const float* x, const float* alpha, const float* dy, int64_t rows, int64_t C, float* dx, float* t,
                  hipStream_t stream)
{
    if (!x || !alpha || !dy || !dx || !t) PPK_FAIL(PPK_ERR_NULL, "prelu_bwd: null pointer");
    if (rows < 1 || C < 1) PPK_FAIL(PPK_ERR_SHAPE, "prelu_bwd: bad sizes");
    k_prelu_bwd<<<dim3(grid_for(rows * C)), dim3(256), 0, stream>>>(x, alpha, dy, rows, (int)C, dx, t);
    PPK_CHECK_LAUNCH("k_prelu_bwd");
    return PPK_OK;
}

size_t ppk_dgi_loss_workspace_bytes(int64_t N)
{
    (void)N;
    return ppk_align_up(sizeof(double) * 1024, 256);
}

int ppk_dgi_loss_fwd(const float* pos_logit, const float* neg_logit, int64_t N, float* loss, void* workspace, size_t ws_bytes,
                     hipStream_t stream)
{
    if (!pos_logit || !neg_logit || !loss || !workspace) PPK_FAIL(PPK_ERR_NULL, "dgi_loss_fwd: null pointer");
    if (N < 1) PPK_FAIL(PPK_ERR_SHAPE, "dgi_loss_fwd: N < 1");
    if (ws_bytes < ppk_dgi_loss_workspace_bytes(N)) PPK_FAIL(PPK_ERR_WS, "dgi_loss_fwd: workspace too small");
    const unsigned nb = grid_for(N, 1024);
    k_dgi_part<<<dim3(nb), dim3(256), 0, stream>>>(pos_logit, neg_logit, N, (double*)workspace);
    PPK_CHECK_LAUNCH("k_dgi_part");
    k_dgi_final<<<dim3(1), dim3(64), 0, stream>>>((const double*)workspace, (int)nb, N, loss);
    PPK_CHECK_LAUNCH("k_dgi_final");
    return PPK_OK;
}

int ppk_dgi_loss_bwd(const float* pos_logit, const float* neg_logit, int64_t N, const float* dloss, float* dpos, float* dneg,
                     hipStream_t stream)
{
    if (!pos_logit || !neg_logit || !dloss || !dpos || !dneg) PPK_FAIL(PPK_ERR_NULL, "dgi_loss_bwd: null pointer");
    if (N < 1) PPK_FAIL(PPK_ERR_SHAPE, "dgi_loss_bwd: N < 1");
    k_dgi_bwd<<<dim3(grid_for(N)), dim3(256), 0, stream>>>(pos_logit, neg_logit, N, dloss, dpos, dneg);
    PPK_CHECK_LAUNCH("k_dgi_bwd");
    return PPK_OK;
}

}  // extern "C"
