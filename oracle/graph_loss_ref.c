/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Loop-for-loop CPU restatement of the reference graph-regularisation loss,
 *   /root/reference/paccmann_polymer/topologically_regularized_models/graph_constrained_loss.py
 *     graph_regularizer_function :4-10
 *     compute_graph_dists        :13-22
 *     graph_loss                 :25-67
 * plus an analytic gradient (the reference relies on autograd; the formula
 * is SURVEY.md Appendix A.3 and is pinned against autograd of the real
 * reference by tests/golden/make_golden.py -> tests/golden/graphreg_*.npz).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.  Nothing in paccmann_polymer_amd/ does.
 *
 * Build: see oracle/Makefile  (gcc -O2 -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* graph_constrained_loss.py:4-10 — value of f and d f / d di (d f / d dj = -that). */
static inline float reg_f(int sgn, float di, float dj, float *dfd_di)
{
    if (sgn == 0) {               /* :5-6  (di - dj)^2 */
        float t = di - dj;
        *dfd_di = 2.0f * t;
        return t * t;
    } else if (sgn > 0) {         /* :7-8  relu(dj - di) */
        float t = dj - di;
        if (t > 0.0f) { *dfd_di = -1.0f; return t; }
        *dfd_di = 0.0f; return 0.0f;
    } else {                      /* :9-10 relu(di - dj) */
        float t = di - dj;
        if (t > 0.0f) { *dfd_di = 1.0f; return t; }
        *dfd_di = 0.0f; return 0.0f;
    }
}

static inline int sgn_d(double x) { return (x > 0.0) - (x < 0.0); }

/* compute_graph_dists (:13-22): p=2 pairwise distances, fp32 result.
 * Accumulated in double then rounded once (the closest-to-exact fp32 matrix;
 * torch's own two cdist code paths differ from it by <=1e-6 rel, SURVEY landmine 5). */
void oracle_cdist(const float *z, int64_t n, int64_t zdim, float *D)
{
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) {
            double s = 0.0;
            for (int64_t k = 0; k < zdim; ++k) {
                double d = (double)z[i * zdim + k] - (double)z[j * zdim + k];
                s += d * d;
            }
            D[i * n + j] = (float)sqrt(s);
        }
}

/*
 * graph_loss (:25-67) on a precomputed fp32 distance matrix D and an fp64
 * graph-distance matrix G (both n x n, row-major).
 *   acc_fp32 != 0 : accumulate `loss += f / max` sequentially in fp32, exactly
 *                   in the reference's loop order (:44-66).
 *   acc_fp32 == 0 : accumulate in double (used as the high-precision arbiter).
 * gD (n x n, may be NULL) receives d loss / d D_ab per ENTRY, as autograd
 * would deliver it to `z_dists`, including the gradient through max (Quirk B,
 * split evenly over ties as torch >= 1.7 does).
 * Returns the loss.
 */
double oracle_graph_loss_from_dists(const float *D, const double *G, int64_t n,
                                    int acc_fp32, double *gD)
{
    float maxd = D[0];
    for (int64_t a = 0; a < n * n; ++a) if (D[a] > maxd) maxd = D[a];   /* :40 */

    float  loss32 = 0.0f;
    double loss64 = 0.0;
    double S = 0.0;                                  /* un-normalised sum, for d/dmax */
    if (gD) memset(gD, 0, sizeof(double) * (size_t)(n * n));

    for (int64_t i = 0; i < n; ++i) {                /* :45 */
        for (int64_t j = i + 1; j < n; ++j) {        /* :46 */
            float  d_ij = D[i * n + j];              /* :47 */
            double g_ij = G[i * n + j];              /* :48 */
            for (int64_t r = i; r < n; ++r) {        /* :49-57 column raster */
                if (r == j) continue;                /* :51 */
                float  d_k = D[r * n + j];
                double g_k = G[r * n + j];
                float dfa;
                float f = reg_f(sgn_d(g_ij - g_k), d_ij, d_k, &dfa);
                loss32 += f / maxd;                  /* :54-56 */
                loss64 += (double)f / (double)maxd;
                S += (double)f;
                if (gD) { gD[i * n + j] += dfa; gD[r * n + j] -= dfa; }
            }
            for (int64_t c = j; c < n; ++c) {        /* :58-66 row raster */
                /* :60  i != j + k always holds since c >= j > i */
                float  d_k = D[i * n + c];           /* :61 */
                double g_k = (double)D[i * n + c];   /* :62  Quirk A: z_dists, not g_dists */
                float dfa;
                float f = reg_f(sgn_d(g_ij - g_k), d_ij, d_k, &dfa);
                loss32 += f / maxd;
                loss64 += (double)f / (double)maxd;
                S += (double)f;
                if (gD) { gD[i * n + j] += dfa; gD[i * n + c] -= dfa; }
            }
        }
    }
    if (gD) {
        /* d/dD of (sum f)/max : (1/max) * dsum  -  (S/max^2) * dmax/dD, ties share evenly */
        int64_t ties = 0;
        for (int64_t a = 0; a < n * n; ++a) if (D[a] == maxd) ++ties;
        double inv = 1.0 / (double)maxd;
        double gmax = -S * inv * inv / (double)ties;
        for (int64_t a = 0; a < n * n; ++a) {
            gD[a] *= inv;
            if (D[a] == maxd) gD[a] += gmax;
        }
    }
    return acc_fp32 ? (double)loss32 : loss64;
}

/*
 * Full path: z [n, zdim] fp32, G [n, n] fp64 -> loss, dz [n, zdim] (double).
 * cdist backward: dz_a = sum_b (gD_ab + gD_ba) (z_a - z_b) / D_ab, 0 where D_ab == 0.
 */
double oracle_graph_loss(const float *z, const double *G, int64_t n, int64_t zdim,
                         int acc_fp32, double *dz)
{
    float  *D  = (float *)malloc(sizeof(float) * (size_t)(n * n));
    double *gD = dz ? (double *)malloc(sizeof(double) * (size_t)(n * n)) : NULL;
    oracle_cdist(z, n, zdim, D);
    double loss = oracle_graph_loss_from_dists(D, G, n, acc_fp32, gD);
    if (dz) {
        memset(dz, 0, sizeof(double) * (size_t)(n * zdim));
        for (int64_t a = 0; a < n; ++a)
            for (int64_t b = 0; b < n; ++b) {
                if (a == b) continue;
                double dab = (double)D[a * n + b];
                if (dab == 0.0) continue;
                double w = (gD[a * n + b] + gD[b * n + a]) / dab;
                if (w == 0.0) continue;
                for (int64_t k = 0; k < zdim; ++k)
                    dz[a * zdim + k] += w * ((double)z[a * zdim + k] - (double)z[b * zdim + k]);
            }
    }
    free(D);
    free(gD);
    return loss;
}

/* Smallest |G_ij - D_ic| over the row-raster comparisons: tests use it to
 * reject seeds whose delta sign is ill-conditioned (SURVEY §7 hard parts). */
double oracle_graph_loss_min_margin(const float *z, const double *G, int64_t n, int64_t zdim)
{
    float *D = (float *)malloc(sizeof(float) * (size_t)(n * n));
    oracle_cdist(z, n, zdim, D);
    double m = INFINITY;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = i + 1; j < n; ++j)
            for (int64_t c = j; c < n; ++c) {
                double t = fabs(G[i * n + j] - (double)D[i * n + c]);
                if (t < m) m = t;
            }
    free(D);
    return m;
}

/*
 * BUILD-DEFINED sparse generalisation (NOT in the reference):
 * the dense loops above with every compared pair restricted to members of an
 * undirected edge set (adj[a*n+b] != 0, symmetric, zero diagonal) and the
 * normaliser taken over edges only.  With a complete edge set it IS the
 * reference loss (pinned by tests against the golden vectors).
 * G holds the edge attribute at [a*n+b] for edges.
 */
double oracle_graph_loss_sparse(const float *z, const double *G, const unsigned char *adj,
                                int64_t n, int64_t zdim, double *dz)
{
    float *D = (float *)malloc(sizeof(float) * (size_t)(n * n));
    double *gD = (double *)calloc((size_t)(n * n), sizeof(double));
    oracle_cdist(z, n, zdim, D);
    float maxd = 0.0f;
    for (int64_t a = 0; a < n; ++a)
        for (int64_t b = 0; b < n; ++b)
            if (adj[a * n + b] && D[a * n + b] > maxd) maxd = D[a * n + b];
    double S = 0.0;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = i + 1; j < n; ++j) {
            if (!adj[i * n + j]) continue;
            float d_ij = D[i * n + j];
            double g_ij = G[i * n + j];
            for (int64_t r = i; r < n; ++r) {
                if (r == j || (r != i && !adj[r * n + j])) continue;
                float dfa;
                float f = reg_f(sgn_d(g_ij - G[r * n + j]), d_ij, D[r * n + j], &dfa);
                S += (double)f;
                gD[i * n + j] += dfa; gD[r * n + j] -= dfa;
            }
            for (int64_t c = j; c < n; ++c) {
                if (c != j && !adj[i * n + c]) continue;
                float dfa;
                float f = reg_f(sgn_d(g_ij - (double)D[i * n + c]), d_ij, D[i * n + c], &dfa);
                S += (double)f;
                gD[i * n + j] += dfa; gD[i * n + c] -= dfa;
            }
        }
    double inv = 1.0 / (double)maxd;
    if (dz) {
        /* ties of the max among edges: both orientations of each tied undirected edge share evenly */
        int64_t ties = 0;
        for (int64_t a = 0; a < n * n; ++a) if (adj[a] && D[a] == maxd) ++ties;
        memset(dz, 0, sizeof(double) * (size_t)(n * zdim));
        for (int64_t a = 0; a < n; ++a)
            for (int64_t b = 0; b < n; ++b) {
                if (a == b || !adj[a * n + b]) continue;
                double dab = (double)D[a * n + b];
                if (dab == 0.0) continue;
                double w = (gD[a * n + b] + gD[b * n + a]) * inv;
                if (D[a * n + b] == maxd) w += 2.0 * (-S * inv * inv / (double)ties);
                w /= dab;
                for (int64_t k = 0; k < zdim; ++k)
                    dz[a * zdim + k] += w * ((double)z[a * zdim + k] - (double)z[b * zdim + k]);
            }
    }
    free(D);
    free(gD);
    return S * inv;
}
