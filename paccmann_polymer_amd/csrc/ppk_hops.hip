// Batched hop-distance matrices of small bundle graphs.
//
// Replaces, per bundle, the host statements of
//   experiments/chem_reactions/reactions_vae.py:79-83   (same in experiments/cora/utils_vae.py:32-35)
//     graph = nx.from_edgelist(edge_index.T.tolist()); graph.add_nodes_from(range(n))
//     dists = nx.floyd_warshall_numpy(graph); dists[isinf] = 0
// i.e. undirected all-pairs hop counts as float64 with 0 for unreachable pairs and the diagonal, rows
// and columns in networkx NODE INSERTION ORDER (Quirk C: first appearance in the edge list, u then v per
// edge, then the remaining range(n)) or, optionally, in index order.
//
// One wave per bundle, n <= 64 nodes: adjacency as one 64-bit mask per node in LDS, the insertion order
// built by a single lane (it is inherently sequential), one BFS per lane over bit masks.  Integer
// arithmetic only: results are exact.
#include "ppk_common.h"

namespace {

__global__ __launch_bounds__(64) void k_hops(const int64_t* __restrict__ esrc, const int64_t* __restrict__ edst,
                                             const int64_t* __restrict__ edge_ptr, const int64_t* __restrict__ seg_ptr,
                                             const int64_t* __restrict__ mat_ptr, int networkx_order,
                                             double* __restrict__ G, int64_t* __restrict__ order_out, int* __restrict__ status)
{
    __shared__ unsigned long long adj[64];
    __shared__ int pos[64];
    __shared__ int order[64];
    const int k = blockIdx.x, lane = threadIdx.x;
    const long long e0 = edge_ptr[k], e1 = edge_ptr[k + 1];
    const long long node0 = seg_ptr[k];
    const int n = (int)(seg_ptr[k + 1] - node0);
    const long long mat0 = mat_ptr[k];
    adj[lane] = 0ull;
    pos[lane] = -1;
    __syncthreads();
    if (lane == 0) {
        int cnt = 0;
        if (networkx_order) {
            for (long long e = e0; e < e1; ++e) {
                const long long uv[2] = {esrc[e], edst[e]};
                for (int q = 0; q < 2; ++q) {
                    const long long x = uv[q];
                    if (x < 0 || x >= n) { atomicOr(status, 1); continue; }
                    if (pos[x] < 0) { pos[x] = cnt; order[cnt++] = (int)x; }
                }
            }
        }
        for (int x = 0; x < n; ++x)
            if (pos[x] < 0) { pos[x] = cnt; order[cnt++] = x; }
    }
    __syncthreads();
    for (long long e = e0 + lane; e < e1; e += 64) {
        const long long u = esrc[e], v = edst[e];
        if (u < 0 || u >= n || v < 0 || v >= n) { atomicOr(status, 1); continue; }
        if (u == v) continue;
        atomicOr(&adj[pos[u]], 1ull << pos[v]);
        atomicOr(&adj[pos[v]], 1ull << pos[u]);
    }
    __syncthreads();
    if (lane < n) {
        if (order_out) order_out[node0 + lane] = order[lane];
        double* row = G + mat0 + (long long)lane * n;
        for (int j = 0; j < n; ++j) row[j] = 0.0;
        unsigned long long visited = 1ull << lane, frontier = visited;
        int level = 0;
        while (frontier) {
            ++level;
            unsigned long long next = 0ull;
            for (unsigned long long f = frontier; f; f &= f - 1) next |= adj[__ffsll((long long)f) - 1];
            next &= ~visited;
            for (unsigned long long f = next; f; f &= f - 1) row[__ffsll((long long)f) - 1] = (double)level;
            visited |= next;
            frontier = next;
        }
    }
}

}  // namespace

extern "C" {

// esrc / edst [E_total]: LOCAL node labels (0 .. n_k - 1) of the concatenated edge lists; edge_ptr / seg_ptr / mat_ptr
// [n_seg + 1] device prefix arrays (edges, nodes, n_k^2).  G [mat_ptr[n_seg]] fp64 out; order_out [seg_ptr[n_seg]]
// (may be null): original label of each row; status [1] device int, OR-ed with 1 when a label is out of range
// (such edges are skipped).  networkx_order != 0: rows in insertion order (the reference), else index order.
int ppk_hop_distances(const int64_t* esrc, const int64_t* edst, const int64_t* edge_ptr, const int64_t* seg_ptr,
                      const int64_t* mat_ptr, int64_t n_seg, int64_t max_n, int networkx_order,
                      double* G, int64_t* order_out, int* status, hipStream_t stream)
{
    if (n_seg == 0) return PPK_OK;
    if (!edge_ptr || !seg_ptr || !mat_ptr || !G || !status) PPK_FAIL(PPK_ERR_NULL, "hop_distances: null pointer");
    if (n_seg < 0 || n_seg > 0x7fffffff) PPK_FAIL(PPK_ERR_SHAPE, "hop_distances: bad n_seg");
    if (max_n < 1 || max_n > 64) PPK_FAIL(PPK_ERR_UNSUP, "hop_distances: bundles of 1..64 nodes (got max_n = %lld); larger graphs: host BFS", (long long)max_n);
    PPK_HIP(hipMemsetAsync(status, 0, sizeof(int), stream));
    k_hops<<<dim3((unsigned)n_seg), dim3(64), 0, stream>>>(esrc, edst, edge_ptr, seg_ptr, mat_ptr, networkx_order, G, order_out, status);
    PPK_CHECK_LAUNCH("k_hops");
    return PPK_OK;
}

}  // extern "C"
