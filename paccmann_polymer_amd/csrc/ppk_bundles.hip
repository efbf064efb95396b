// Device-side bundle sampler (the part of batch construction that was still host Python).
//
// Replaces, per root node of the relation graph, the statements of
//   experiments/chem_reactions/utils.py:155-196 (load_data: the 2-hop "bundle" of a node)
//     close = np.random.permutation(list(G.neighbors(root)))[:max_close_neighbors]
//     total_far = sum(len(list(G.neighbors(x))) for x in close)
//     shortlist = [[root, x] for x in close]
//     while len(shortlist) < max_total_bundle and counter < total_far:
//         cur = close[counter % len(close)]; far = np.random.choice(list(G.neighbors(cur)))
//         shortlist.append([cur, far]); counter += 1
//     nx.from_edgelist(shortlist) -> convert_node_labels_to_integers -> to_pandas_edgelist
// The reference draws from numpy's global Mersenne Twister, one stream through all roots in order — inherently serial
// and not reproducible on a device.  Here every root has its own counter-based stream (Philox4x32-10, key = seed,
// counter = (draw index, 0, root lo, root hi)), so all roots are sampled at once and a root's bundle does not depend on
// the others.  Same distribution per root (uniform ordered max_close-subset of the neighbours = the head of a uniform
// permutation; uniform choice of a far neighbour), different numbers than numpy: the oracle
// (oracle/bundle_sampler_ref.py) replays the same Philox draws through the host restatement of the reference's
// statements (data_prep.sample_bundle / relabel_bundle) and the outputs are compared bit for bit.
//
// One thread per root: at most max_total (<= 32) draws and a relabelling of at most 33 nodes — integer work of a few
// hundred operations; the kernel is bound by the random reads of the adjacency lists.
#include "ppk_common.h"

namespace {

constexpr int MAXT = 32;            // max_total_bundle supported (the reference uses 20)
constexpr int MAXN = MAXT + 1;

__device__ __forceinline__ void philox(unsigned (&c)[4], unsigned k0, unsigned k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
// draw number `idx` of root `root`: an integer in [0, n), n < 2^31  (multiply-shift on the first Philox word)
__device__ __forceinline__ long long draw(unsigned long long seed, long long root, unsigned idx, long long n)
{
    unsigned c[4] = {idx, 0u, (unsigned)root, (unsigned)((unsigned long long)root >> 32)};
    philox(c, (unsigned)seed, (unsigned)(seed >> 32));
    return (long long)(((unsigned long long)c[0] * (unsigned long long)n) >> 32);
}

__global__ __launch_bounds__(64) void k_sample_bundles(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col, long long N,
                                                       const int64_t* __restrict__ roots, long long R, int max_close, int max_total,
                                                       unsigned long long seed, int64_t* __restrict__ shortlist, int* __restrict__ n_short,
                                                       int64_t* __restrict__ labels, int* __restrict__ n_nodes,
                                                       int64_t* __restrict__ edges, int* __restrict__ n_edges, int* __restrict__ status)
{
    const long long r = (long long)blockIdx.x * 64 + threadIdx.x;
    if (r >= R) return;
    const long long root = roots ? roots[r] : r;
    int64_t* sl = shortlist + r * max_total * 2;
    if (root < 0 || root >= N) { atomicOr(status, 1); n_short[r] = 0; n_nodes[r] = 0; n_edges[r] = 0; return; }
    // ---- close neighbours: the first max_close entries of a uniform permutation = max_close distinct positions,
    // draw i picks the d-th position not taken yet
    const long long r0 = rowptr[root], deg = rowptr[root + 1] - r0;
    const int k = (int)(deg < max_close ? deg : max_close);
    long long taken[MAXT], close[MAXT];
    unsigned nd = 0;
    long long total_far = 0;
    for (int i = 0; i < k; ++i) {
        long long idx = draw(seed, root, nd++, deg - i);
        // taken[0 .. i) is kept ascending: skip over the positions already used
        int at = 0;
        for (; at < i && taken[at] <= idx; ++at) ++idx;
        for (int m = i; m > at; --m) taken[m] = taken[m - 1];
        taken[at] = idx;
        close[i] = col[r0 + idx];
        if (close[i] < 0 || close[i] >= N) {          // a column outside the graph: nothing is read through it
            atomicOr(status, 2);
            n_short[r] = 0; n_nodes[r] = 0; n_edges[r] = 0;
            return;
        }
        total_far += rowptr[close[i] + 1] - rowptr[close[i]];
        sl[2 * i] = root; sl[2 * i + 1] = close[i];
    }
    int ns = k;
    for (long long counter = 0; ns < max_total && counter < total_far; ++counter) {
        const long long cur = close[counter % k];
        const long long c0 = rowptr[cur], cd = rowptr[cur + 1] - c0;
        const long long far = col[c0 + draw(seed, root, nd++, cd)];
        sl[2 * ns] = cur; sl[2 * ns + 1] = far;
        ++ns;
    }
    n_short[r] = ns;
    // ---- nx.from_edgelist(shortlist): nodes by first appearance, adjacency lists in insertion order, duplicates merged;
    // convert_node_labels_to_integers + to_pandas_edgelist: every undirected edge once, by node then adjacency order
    long long order[MAXN];
    unsigned char adj[MAXN][MAXN], adjn[MAXN];
    unsigned long long seen[MAXN];
    int nn = 0;
    auto pos_of = [&](long long x) {
        for (int p = 0; p < nn; ++p) if (order[p] == x) return p;
        order[nn] = x; adjn[nn] = 0; seen[nn] = 0ull;
        return nn++;
    };
    for (int e = 0; e < ns; ++e) {
        const int pu = pos_of(sl[2 * e]), pv = pos_of(sl[2 * e + 1]);
        bool have = false;
        for (int m = 0; m < adjn[pu]; ++m) have = have || adj[pu][m] == pv;
        if (!have) {
            adj[pu][adjn[pu]++] = (unsigned char)pv;
            if (pu != pv) adj[pv][adjn[pv]++] = (unsigned char)pu;
        }
    }
    int64_t* lab = labels + r * (max_total + 1);
    for (int p = 0; p < nn; ++p) lab[p] = order[p];
    n_nodes[r] = nn;
    int64_t* ed = edges + r * max_total * 2;
    int ne = 0;
    for (int pu = 0; pu < nn; ++pu)
        for (int m = 0; m < adjn[pu]; ++m) {
            const int pv = adj[pu][m], lo = pu < pv ? pu : pv, hi = pu < pv ? pv : pu;
            if (seen[lo] >> hi & 1ull) continue;
            seen[lo] |= 1ull << hi;
            ed[2 * ne] = pu; ed[2 * ne + 1] = pv;
            ++ne;
        }
    n_edges[r] = ne;
}

}  // namespace

extern "C" int ppk_sample_bundles(const int64_t* rowptr, const int64_t* col, int64_t N, const int64_t* roots, int64_t R,
                                  int max_close, int max_total, uint64_t seed, int64_t* shortlist, int* n_short,
                                  int64_t* labels, int* n_nodes, int64_t* edges, int* n_edges, int* status, hipStream_t stream)
{
    if (!rowptr || !col || !shortlist || !n_short || !labels || !n_nodes || !edges || !n_edges || !status)
        PPK_FAIL(PPK_ERR_NULL, "sample_bundles: null pointer");
    if (N < 1 || R < 1 || max_close < 1) PPK_FAIL(PPK_ERR_SHAPE, "sample_bundles: bad sizes (N=%lld R=%lld)", (long long)N, (long long)R);
    if (max_total < 1 || max_total > MAXT || max_close > max_total)
        PPK_FAIL(PPK_ERR_UNSUP, "sample_bundles: max_total_bundle must be in [1, %d] and >= max_close_neighbors", MAXT);
    k_sample_bundles<<<dim3((unsigned)((R + 63) / 64)), dim3(64), 0, stream>>>(rowptr, col, N, roots, R, max_close, max_total, seed,
                                                                               shortlist, n_short, labels, n_nodes, edges, n_edges, status);
    PPK_CHECK_LAUNCH("k_sample_bundles");
    return PPK_OK;
}
