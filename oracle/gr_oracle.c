/*
 * gr_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see gr_oracle.h).
 *
 * Plain C99 restatement of the reference's host algorithms.  Each function cites the
 * reference file:line it follows.  Nothing here is linked into libgunrock.so.
 */
#define _GNU_SOURCE
#include "gr_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void gro_csr_free(gro_csr *g)
{
    if (!g) return;
    free(g->row_offsets);
    free(g->col_indices);
    free(g->edge_values);
    memset(g, 0, sizeof(*g));
}

/* ------------------------------------------------------------------------------------------
 * G1  graphio::ReadMarketStream   (gunrock/graphio/market.cuh:56-215)
 * ---------------------------------------------------------------------------------------- */
int gro_read_market(const char *path, int undirected, int reversed,
                    gro_tuple **coo_out, int32_t *nodes_out, int32_t *tuples_out)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;

    char line[1024];
    long long expected = 0, got = -1; /* got == -1: header not seen yet (market.cuh:65,92) */
    int32_t nodes = 0;
    gro_tuple *coo = NULL;
    int rc = 0;

    /* The reference scans "%[^\n]\n": one non-empty line, then ALL following white space
     * (blank lines, indentation of the next line).  An empty first line ends the parse
     * (market.cuh:81-83).  Lines are bounded to the reference's 1024-byte buffer. */
    while (fscanf(f, "%1023[^\n]\n", line) > 0) {
        if (line[0] == '%') continue;                       /* comment (market.cuh:85-87) */
        if (got == -1) {                                    /* problem line (market.cuh:89-118) */
            long long nx, ny, ne;
            if (sscanf(line, "%lld %lld %lld", &nx, &ny, &ne) != 3 || nx != ny) { rc = -1; break; }
            nodes = (int32_t)nx;
            expected = undirected ? ne * 2 : ne;            /* market.cuh:108 */
            coo = (gro_tuple *)malloc(sizeof(gro_tuple) * (size_t)(expected > 0 ? expected : 1));
            got = 0;
            continue;
        }
        if (got >= expected) { rc = -1; break; }            /* market.cuh:126-133 */
        long long first, second, value;
        int k = sscanf(line, "%lld %lld %lld", &first, &second, &value);
        if (k < 2) { rc = -1; break; }                      /* market.cuh:138-146 */
        if (k == 2) value = 1;                              /* pattern entry (market.cuh:146-148) */
        /* first number is the COLUMN, second the ROW (market.cuh:139-141) */
        long long ll_col = first, ll_row = second;
        coo[got].val = value;
        if (reversed && !undirected) {                      /* market.cuh:161-164 */
            coo[got].col = (int32_t)(ll_row - 1);
            coo[got].row = (int32_t)(ll_col - 1);
        } else {                                            /* market.cuh:165-169 */
            coo[got].row = (int32_t)(ll_row - 1);
            coo[got].col = (int32_t)(ll_col - 1);
        }
        got++;
        if (undirected) {                                   /* market.cuh:173-184 */
            coo[got].row = (int32_t)(ll_col - 1);
            coo[got].col = (int32_t)(ll_row - 1);
            coo[got].val = value;
            got++;
        }
    }
    fclose(f);
    if (rc == 0 && (coo == NULL || got != expected)) rc = -1;   /* market.cuh:187-198 */
    if (rc != 0) { free(coo); return rc; }
    *coo_out = coo;
    *nodes_out = nodes;
    *tuples_out = (int32_t)expected;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * G2  Csr::FromCoo (gunrock/csr.cuh:247-340), comparator coo.cuh:71-85
 * ---------------------------------------------------------------------------------------- */
static int tuple_less(const gro_tuple *x, const gro_tuple *y)
{
    if (x->row < y->row) return 1;
    if (x->row == y->row && x->col < y->col) return 1;
    return 0;
}

/* bottom-up merge sort: stable, like std::stable_sort (csr.cuh:266-268) */
static void stable_sort_tuples(gro_tuple *a, int64_t n)
{
    if (n < 2) return;
    gro_tuple *tmp = (gro_tuple *)malloc(sizeof(gro_tuple) * (size_t)n);
    gro_tuple *src = a, *dst = tmp;
    for (int64_t width = 1; width < n; width *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * width) {
            int64_t mid = lo + width < n ? lo + width : n;
            int64_t hi = lo + 2 * width < n ? lo + 2 * width : n;
            int64_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                /* take from the right run only when strictly smaller => stability */
                if (tuple_less(&src[j], &src[i])) dst[k++] = src[j++];
                else dst[k++] = src[i++];
            }
            while (i < mid) dst[k++] = src[i++];
            while (j < hi) dst[k++] = src[j++];
        }
        gro_tuple *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, sizeof(gro_tuple) * (size_t)n);
    free(tmp);
}

int gro_csr_from_coo(gro_tuple *coo, int32_t nodes, int32_t tuples, gro_csr *out)
{
    memset(out, 0, sizeof(*out));
    out->nodes = nodes;
    out->row_offsets = (int32_t *)calloc((size_t)nodes + 1, sizeof(int32_t));
    out->col_indices = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tuples > 0 ? tuples : 1));
    out->edge_values = (int32_t *)malloc(sizeof(int32_t) * (size_t)(tuples > 0 ? tuples : 1));
    if (tuples <= 0) return 0;

    stable_sort_tuples(coo, tuples);

    /* keep the first tuple unless it is a self loop (csr.cuh:272-277); keep tuple i+1 iff it
     * differs from tuple i and is not a self loop (csr.cuh:278-288) */
    int32_t kept = 0;
    int32_t prev_row = -1;
    for (int32_t i = 0; i < tuples; ++i) {
        int keep;
        if (i == 0) keep = (coo[0].col != coo[0].row);
        else keep = ((coo[i].col != coo[i - 1].col) || (coo[i].row != coo[i - 1].row)) &&
                    (coo[i].col != coo[i].row);
        if (!keep) continue;
        int32_t r = coo[i].row;
        for (int32_t row = prev_row + 1; row <= r; ++row) out->row_offsets[row] = kept; /* csr.cuh:296-299 */
        prev_row = r;
        out->col_indices[kept] = coo[i].col;
        out->edge_values[kept] = (int32_t)coo[i].val;       /* Coo::Val -> Value=int (coo.cuh:32-58) */
        kept++;
    }
    for (int32_t row = prev_row + 1; row <= nodes; ++row) out->row_offsets[row] = kept; /* csr.cuh:308-310 */
    out->edges = kept;                                       /* csr.cuh:311 */
    return 0;
}

int gro_build_market(const char *path, int undirected, int reversed, gro_csr *out)
{
    gro_tuple *coo = NULL;
    int32_t nodes = 0, tuples = 0;
    int rc = gro_read_market(path, undirected, reversed, &coo, &nodes, &tuples);
    if (rc) return rc;
    rc = gro_csr_from_coo(coo, nodes, tuples, out);
    free(coo);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * G3  graphio::BuildRmatGraph (gunrock/graphio/rmat.cuh:27-91) + utils.cuh:47-130
 * ---------------------------------------------------------------------------------------- */
void gro_srand(unsigned seed) { srand(seed); }

static double ref_sprng(void) { return (double)rand() / RAND_MAX; }      /* utils.cuh:47-50 */
static int ref_flip(void) { return rand() >= RAND_MAX / 2; }             /* utils.cuh:52-55 */

/* shared by both generators: utils.cuh:58-82 (strict inequalities kept) */
static void choose_partition(int32_t *u, int32_t *v, int32_t step,
                             double a, double b, double c, double d, double p)
{
    if (p < a) {
    } else if ((a < p) && (p < a + b)) {
        *v += step;
    } else if ((a + b < p) && (p < a + b + c)) {
        *u += step;
    } else if ((a + b + c < p) && (p < a + b + c + d)) {
        *u += step;
        *v += step;
    }
}

int gro_rmat_reference(int32_t nodes, int32_t edges, int undirected,
                       double a0, double b0, double c0, double d0, gro_csr *out)
{
    if (nodes < 0 || edges < 0) return -1;
    int32_t directed = undirected ? edges * 2 : edges;       /* rmat.cuh:46 */
    gro_tuple *coo = (gro_tuple *)malloc(sizeof(gro_tuple) * (size_t)(directed > 0 ? directed : 1));
    for (int32_t i = 0; i < edges; ++i) {
        double a = a0, b = b0, c = c0, d = d0;
        int32_t u = 1, v = 1, step = nodes / 2;              /* rmat.cuh:57-59 */
        while (step >= 1) {
            choose_partition(&u, &v, step, a, b, c, d, ref_sprng());
            step /= 2;
            /* VaryParams (utils.cuh:84-130): Flip() then Sprng() per parameter, 5 % noise */
            const double var = 0.05;
            if (ref_flip()) a += a * var * ref_sprng(); else a -= a * var * ref_sprng();
            if (ref_flip()) b += b * var * ref_sprng(); else b -= b * var * ref_sprng();
            if (ref_flip()) c += c * var * ref_sprng(); else c -= c * var * ref_sprng();
            if (ref_flip()) d += d * var * ref_sprng(); else d -= d * var * ref_sprng();
            double S = a + b + c + d;
            a = a / S; b = b / S; c = c / S; d = d / S;
        }
        coo[i].row = u - 1; coo[i].col = v - 1; coo[i].val = 1;   /* rmat.cuh:68-70 */
        if (undirected) {                                          /* rmat.cuh:72-78 */
            coo[edges + i].row = coo[i].col;
            coo[edges + i].col = coo[i].row;
            coo[edges + i].val = 1;
        }
    }
    int rc = gro_csr_from_coo(coo, nodes, directed, out);
    free(coo);
    return rc;
}

/* ---- own seeded generator; the HIP kernel rmat_seeded_kernel follows the same spec ---- */
static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static inline uint64_t rmat_draw(uint64_t seed, uint64_t edge, unsigned level, unsigned k)
{
    uint64_t ctr = (edge << 10) | ((uint64_t)level << 4) | k;
    return splitmix64(seed ^ splitmix64(ctr));
}
static inline double u01(uint64_t r) { return (double)(r >> 11) * (1.0 / 9007199254740992.0); }

void gro_rmat_seeded_coo(int scale, int64_t pairs, uint64_t seed, int undirected,
                         double a0, double b0, double c0, double d0,
                         int64_t first, int64_t count, int32_t *rows, int32_t *cols)
{
    (void)pairs; (void)undirected;
    for (int64_t t = 0; t < count; ++t) {
        uint64_t e = (uint64_t)(first + t);
        double a = a0, b = b0, c = c0, d = d0;
        int32_t u = 0, v = 0;
        for (int level = 0; level < scale; ++level) {
            int32_t step = (int32_t)1 << (scale - 1 - level);
            choose_partition(&u, &v, step, a, b, c, d, u01(rmat_draw(seed, e, level, 0)));
            uint64_t flips = rmat_draw(seed, e, level, 1);
            const double var = 0.05;
            double sa = u01(rmat_draw(seed, e, level, 2));
            double sb = u01(rmat_draw(seed, e, level, 3));
            double sc = u01(rmat_draw(seed, e, level, 4));
            double sd = u01(rmat_draw(seed, e, level, 5));
            double ta = (a * var) * sa, tb = (b * var) * sb, tc = (c * var) * sc, td = (d * var) * sd;
            a = (flips & 1) ? a + ta : a - ta;
            b = (flips & 2) ? b + tb : b - tb;
            c = (flips & 4) ? c + tc : c - tc;
            d = (flips & 8) ? d + td : d - td;
            double S = ((a + b) + c) + d;
            a = a / S; b = b / S; c = c / S; d = d / S;
        }
        rows[t] = u;
        cols[t] = v;
    }
}

int gro_rmat_seeded(int scale, int64_t pairs, uint64_t seed, int undirected,
                    double a, double b, double c, double d, gro_csr *out)
{
    int64_t directed = undirected ? 2 * pairs : pairs;
    if (directed > INT32_MAX) return -1;
    int32_t *rows = (int32_t *)malloc(sizeof(int32_t) * (size_t)(pairs > 0 ? pairs : 1));
    int32_t *cols = (int32_t *)malloc(sizeof(int32_t) * (size_t)(pairs > 0 ? pairs : 1));
    gro_rmat_seeded_coo(scale, pairs, seed, undirected, a, b, c, d, 0, pairs, rows, cols);
    gro_tuple *coo = (gro_tuple *)malloc(sizeof(gro_tuple) * (size_t)(directed > 0 ? directed : 1));
    for (int64_t i = 0; i < pairs; ++i) {
        coo[i].row = rows[i]; coo[i].col = cols[i]; coo[i].val = 1;
        if (undirected) { coo[pairs + i].row = cols[i]; coo[pairs + i].col = rows[i]; coo[pairs + i].val = 1; }
    }
    free(rows); free(cols);
    int rc = gro_csr_from_coo(coo, (int32_t)1 << scale, (int32_t)directed, out);
    free(coo);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * G4  csr.cuh:442-455, 475-485
 * ---------------------------------------------------------------------------------------- */
int32_t gro_highest_degree_node(const int32_t *ro, int32_t nodes, int32_t *max_degree)
{
    int32_t degree = 0, src = 0;
    for (int32_t v = 0; v < nodes; ++v)
        if (ro[v + 1] - ro[v] > degree) { degree = ro[v + 1] - ro[v]; src = v; }  /* strict > : first max */
    if (max_degree) *max_degree = degree;
    return src;
}

int32_t gro_average_degree(const int32_t *ro, int32_t nodes)
{
    double mean = 0, count = 0;                                   /* running mean, csr.cuh:477-482 */
    for (int32_t v = 0; v < nodes; ++v) {
        count += 1;
        mean += (ro[v + 1] - ro[v] - mean) / count;
    }
    return (int32_t)mean;
}

/* ------------------------------------------------------------------------------------------
 * X1  SimpleReferenceBfs (tests/bfs/test_bfs.cu:258-322)
 * ---------------------------------------------------------------------------------------- */
int32_t gro_bfs(const int32_t *ro, const int32_t *ci, int32_t nodes,
                int32_t src, int32_t *labels, int32_t *preds)
{
    for (int32_t i = 0; i < nodes; ++i) { labels[i] = -1; if (preds) preds[i] = -1; }
    if (nodes <= 0) return 1;
    labels[src] = 0;
    int32_t depth = 0;
    /* std::deque used as FIFO (test_bfs.cu:279-280): each vertex enters once => array of n slots */
    int32_t *fifo = (int32_t *)malloc(sizeof(int32_t) * (size_t)nodes);
    int64_t head = 0, tail = 0;
    fifo[tail++] = src;
    while (head < tail) {
        int32_t u = fifo[head++];
        int32_t nd = labels[u] + 1;
        for (int32_t e = ro[u]; e < ro[u + 1]; ++e) {
            int32_t w = ci[e];
            if (labels[w] == -1) {
                labels[w] = nd;
                if (preds) preds[w] = u;
                if (depth < nd) depth = nd;
                fifo[tail++] = w;
            }
        }
    }
    if (preds) preds[src] = -1;
    free(fifo);
    return depth + 1;                                             /* test_bfs.cu:318 */
}

/* Level-synchronous top-down BFS on all host cores (OpenMP): NOT a restatement of anything in the reference (its CPU BFS
 * is the serial loop above) -- the stronger CPU baseline SURVEY 8(d) asks to time next to it, with the thread count
 * reported.  Labels are BFS depths, hence identical to gro_bfs; each vertex is claimed with one compare-and-swap. */
int32_t gro_bfs_parallel(const int32_t *ro, const int32_t *ci, int32_t nodes, int32_t src, int32_t *labels, int32_t threads)
{
    if (nodes <= 0) return 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    int used = 1;
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
    }
#else
    const int used = 1;
    (void)threads;
#endif
    int32_t *cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)nodes);
    int32_t *next = (int32_t *)malloc(sizeof(int32_t) * (size_t)nodes);
    int64_t *counts = (int64_t *)calloc((size_t)used + 1, sizeof(int64_t));
    int32_t **local = (int32_t **)calloc((size_t)used, sizeof(int32_t *));
    int64_t *local_cap = (int64_t *)calloc((size_t)used, sizeof(int64_t));
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < nodes; ++i) labels[i] = -1;
    labels[src] = 0;
    cur[0] = src;
    int64_t cur_len = 1;
    int32_t depth = 0;
    while (cur_len > 0) {
        const int32_t nd = depth + 1;
#pragma omp parallel
        {
#ifdef _OPENMP
            const int t = omp_get_thread_num();
#else
            const int t = 0;
#endif
            int64_t n = 0;
#pragma omp for schedule(dynamic, 64)
            for (int64_t i = 0; i < cur_len; ++i) {
                const int32_t u = cur[i];
                for (int32_t e = ro[u]; e < ro[u + 1]; ++e) {
                    const int32_t w = ci[e];
                    if (labels[w] == -1 && __sync_bool_compare_and_swap(&labels[w], -1, nd)) {
                        if (n == local_cap[t]) {
                            local_cap[t] = local_cap[t] ? 2 * local_cap[t] : 4096;
                            local[t] = (int32_t *)realloc(local[t], sizeof(int32_t) * (size_t)local_cap[t]);
                        }
                        local[t][n++] = w;
                    }
                }
            }
            counts[t + 1] = n;
#pragma omp barrier
#pragma omp single
            for (int k = 0; k < used; ++k) counts[k + 1] += counts[k];
            memcpy(next + counts[t], local[t], sizeof(int32_t) * (size_t)n);
        }
        cur_len = counts[used];
        counts[0] = 0;
        int32_t *tmp = cur; cur = next; next = tmp;
        if (cur_len > 0) depth = nd;
    }
    for (int k = 0; k < used; ++k) free(local[k]);
    free(local); free(local_cap); free(counts); free(cur); free(next);
    return used;
}

/* ------------------------------------------------------------------------------------------
 * X2  Dijkstra over uint32 (tests/sssp/test_sssp.cu:242-343 semantics)
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint32_t key; int32_t v; } heap_item;

static void heap_push(heap_item *h, int64_t *n, heap_item x)
{
    int64_t i = (*n)++;
    while (i > 0) {
        int64_t p = (i - 1) / 2;
        if (h[p].key <= x.key) break;
        h[i] = h[p]; i = p;
    }
    h[i] = x;
}
static heap_item heap_pop(heap_item *h, int64_t *n)
{
    heap_item top = h[0];
    heap_item x = h[--(*n)];
    int64_t i = 0;
    for (;;) {
        int64_t c = 2 * i + 1;
        if (c >= *n) break;
        if (c + 1 < *n && h[c + 1].key < h[c].key) c++;
        if (x.key <= h[c].key) break;
        h[i] = h[c]; i = c;
    }
    h[i] = x;
    return top;
}

void gro_sssp(const int32_t *ro, const int32_t *ci, const uint32_t *w,
              int32_t nodes, int32_t src, uint32_t *dist, int32_t *preds)
{
    for (int32_t i = 0; i < nodes; ++i) { dist[i] = UINT32_MAX; if (preds) preds[i] = i; }
    if (nodes <= 0) return;
    int64_t cap = (int64_t)ro[nodes] + 2, n = 0;
    heap_item *h = (heap_item *)malloc(sizeof(heap_item) * (size_t)cap);
    dist[src] = 0;
    heap_item s = {0u, src};
    heap_push(h, &n, s);
    while (n > 0) {
        heap_item it = heap_pop(h, &n);
        if (it.key != dist[it.v]) continue;                       /* stale entry */
        for (int32_t e = ro[it.v]; e < ro[it.v + 1]; ++e) {
            /* closed_plus<unsigned>: saturate at "infinity" so an overflowing path never wins */
            uint64_t cand = (uint64_t)it.key + (uint64_t)w[e];
            if (cand >= UINT32_MAX) continue;
            int32_t t = ci[e];
            if ((uint32_t)cand < dist[t]) {
                dist[t] = (uint32_t)cand;
                if (preds) preds[t] = it.v;
                heap_item nx = {(uint32_t)cand, t};
                heap_push(h, &n, nx);
            }
        }
    }
    free(h);
}

/* ------------------------------------------------------------------------------------------
 * X3  connected components: union-find with min-id representative
 * ---------------------------------------------------------------------------------------- */
static int32_t uf_find(int32_t *p, int32_t x)
{
    int32_t r = x;
    while (p[r] != r) r = p[r];
    while (p[x] != r) { int32_t nx = p[x]; p[x] = r; x = nx; }
    return r;
}

int32_t gro_cc(const int32_t *ro, const int32_t *ci, int32_t nodes, int32_t *comp)
{
    for (int32_t v = 0; v < nodes; ++v) comp[v] = v;
    for (int32_t v = 0; v < nodes; ++v)
        for (int32_t e = ro[v]; e < ro[v + 1]; ++e) {
            int32_t a = uf_find(comp, v), b = uf_find(comp, ci[e]);
            if (a == b) continue;
            if (a < b) comp[b] = a; else comp[a] = b;             /* smaller id is the root */
        }
    int32_t count = 0;
    for (int32_t v = 0; v < nodes; ++v) { comp[v] = uf_find(comp, v); }
    for (int32_t v = 0; v < nodes; ++v) if (comp[v] == v) count++;   /* cc_problem.cuh:164-170 */
    return count;
}

/* cc_enactor.cuh:165-873 driven sequentially; functors cc_functor.cuh:18-410 */
int32_t gro_cc_reference_schedule(const int32_t *ro, const int32_t *ci, int32_t nodes,
                                  int32_t *comp, int32_t *edge_sweeps, int32_t *vertex_sweeps)
{
    int32_t m = nodes > 0 ? ro[nodes] : 0;
    int32_t *from = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m > 0 ? m : 1));
    int32_t *mask = (int32_t *)calloc((size_t)(nodes > 0 ? nodes : 1), sizeof(int32_t));
    unsigned char *mark = (unsigned char *)calloc((size_t)(m > 0 ? m : 1), 1);
    int32_t ih = 0, ij = 0;
    for (int32_t v = 0; v < nodes; ++v) {                         /* cc_problem.cuh:262-272 */
        comp[v] = v;
        for (int32_t e = ro[v]; e < ro[v + 1]; ++e) from[e] = v;
    }
    /* HookInit over all edges (cc_functor.cuh:91-104; cc_enactor.cuh:407-424) */
    for (int32_t e = 0; e < m; ++e) {
        int32_t f = from[e], t = ci[e];
        int32_t mx = f > t ? f : t, mn = f + t - mx;
        comp[mx] = mn;
    }
    if (m > 0) ih++;
    /* PtrJump until no change (cc_functor.cuh:230-262; cc_enactor.cuh:442-493) */
    for (int flag = 0; !flag;) {
        flag = 1;
        for (int32_t v = 0; v < nodes; ++v) {
            int32_t p = comp[v], g = comp[p];
            if (p != g) { flag = 0; comp[v] = g; }
        }
        ij++;
    }
    /* UpdateMask (cc_functor.cuh:30-47; cc_enactor.cuh:501-517) */
    for (int32_t v = 0; v < nodes; ++v) mask[v] = (comp[v] == v) ? 0 : 1;
    ij++;
    for (;;) {                                                    /* cc_enactor.cuh:524-862 */
        int edge_flag = 1;
        for (int32_t e = 0; e < m; ++e) {                         /* HookMax, cc_functor.cuh:172-216 */
            if (mark[e]) continue;
            int32_t pf = comp[from[e]], pt = comp[ci[e]];
            int32_t mx = pf > pt ? pf : pt, mn = pf + pt - mx;
            if (mx == mn) mark[e] = 1;
            else { comp[mx] = mn; edge_flag = 0; }
        }
        ih++;
        if (edge_flag) break;                                     /* cc_enactor.cuh:757-761 */
        for (int flag = 0; !flag;) {                              /* PtrJumpMask, cc_functor.cuh:276-313 */
            flag = 1;
            for (int32_t v = 0; v < nodes; ++v) {
                if (mask[v] != 0) continue;
                int32_t p = comp[v], g = comp[p];
                if (p != g) { flag = 0; comp[v] = g; }
                else mask[v] = -1;
            }
            ij++;
        }
        for (int32_t v = 0; v < nodes; ++v)                       /* PtrJumpUnmask, cc_functor.cuh:327-352 */
            if (mask[v] == 1) comp[v] = comp[comp[v]];
        ij++;
        for (int32_t v = 0; v < nodes; ++v) mask[v] = (comp[v] == v) ? 0 : 1;   /* UpdateMask */
        ij++;
    }
    int32_t count = 0;
    for (int32_t v = 0; v < nodes; ++v) if (comp[v] == v) count++;
    if (edge_sweeps) *edge_sweeps = ih;
    if (vertex_sweeps) *vertex_sweeps = ij;
    free(from); free(mask); free(mark);
    return count;
}

/* ------------------------------------------------------------------------------------------
 * DisplayStats (tests/bfs/test_bfs.cu:184-216)
 * ---------------------------------------------------------------------------------------- */
void gro_bfs_stats(const int32_t *ro, int32_t nodes, const int32_t *labels,
                   int64_t *nodes_visited, int64_t *edges_visited)
{
    int64_t nv = 0, ev = 0;
    for (int32_t v = 0; v < nodes; ++v)
        if (labels[v] > -1) { nv++; ev += ro[v + 1] - ro[v]; }
    *nodes_visited = nv;
    *edges_visited = ev;
}

/* ------------------------------------------------------------------------------------------
 * validators for the non-unique outputs (preds are "valid parent", not bit-equal)
 * ---------------------------------------------------------------------------------------- */
int64_t gro_check_bfs_preds(const int32_t *ro, const int32_t *ci, int32_t nodes, int32_t src,
                            const int32_t *labels, const int32_t *preds)
{
    int64_t bad = 0;
    for (int32_t v = 0; v < nodes; ++v) {
        if (v == src) { if (preds[v] != -1) bad++; continue; }
        if (labels[v] < 0) { if (preds[v] != -2 && preds[v] != -1) bad++; continue; }
        int32_t p = preds[v];
        if (p < 0 || p >= nodes || labels[p] != labels[v] - 1) { bad++; continue; }
        int found = 0;
        for (int32_t e = ro[p]; e < ro[p + 1] && !found; ++e) found = (ci[e] == v);
        if (!found) bad++;
    }
    return bad;
}

int64_t gro_check_sssp_preds(const int32_t *ro, const int32_t *ci, const uint32_t *w, int32_t nodes,
                             int32_t src, const uint32_t *dist, const int32_t *preds)
{
    int64_t bad = 0;
    for (int32_t v = 0; v < nodes; ++v) {
        if (v == src || dist[v] == UINT32_MAX) { if (preds[v] != v) bad++; continue; }
        int32_t p = preds[v];
        if (p < 0 || p >= nodes || dist[p] == UINT32_MAX) { bad++; continue; }
        int found = 0;
        for (int32_t e = ro[p]; e < ro[p + 1] && !found; ++e)
            found = (ci[e] == v) && ((uint64_t)dist[p] + w[e] == dist[v]);
        if (!found) bad++;
    }
    return bad;
}

/* ------------------------------------------------------------------------------------------
 * X5: betweenness centrality.  The reference validates against Boost's brandes_betweenness_centrality on an
 * undirectedS graph for "all sources" (tests/bc/test_bc.cu:144-213) and against its own serial Brandes pass for one
 * source (:224-300); its GPU drivers halve the accumulated values (bc_app.cu:112-113, test_bc.cu:443-444), which is
 * what Boost does for undirected graphs.  Restated: Brandes' algorithm per source (BFS order, sigma counts, reverse
 * accumulation delta[v] += sigma[v] / sigma[w] * (1 + delta[w]) over edges to the next level), bc[v] += delta[v]
 * for v != source, everything in double, halved at the end.  src = -1: every vertex in turn.
 * ---------------------------------------------------------------------------------------- */
int gro_bc(const int32_t *ro, const int32_t *ci, int32_t nodes, int32_t src, double *bc_out, double *sigma_out)
{
    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nodes > 0 ? nodes : 1));
    int32_t *label = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nodes > 0 ? nodes : 1));
    double *sigma = (double *)malloc(sizeof(double) * (size_t)(nodes > 0 ? nodes : 1));
    double *delta = (double *)malloc(sizeof(double) * (size_t)(nodes > 0 ? nodes : 1));
    if (!order || !label || !sigma || !delta) { free(order); free(label); free(sigma); free(delta); return 1; }
    for (int32_t v = 0; v < nodes; ++v) bc_out[v] = 0.0;
    const int32_t first = (src == -1) ? 0 : src, last = (src == -1) ? nodes : src + 1;
    for (int32_t s = first; s < last; ++s) {
        for (int32_t v = 0; v < nodes; ++v) { label[v] = -1; sigma[v] = 0.0; delta[v] = 0.0; }
        int32_t head = 0, tail = 0;
        label[s] = 0; sigma[s] = 1.0; order[tail++] = s;
        while (head < tail) {
            const int32_t v = order[head++];
            for (int32_t e = ro[v]; e < ro[v + 1]; ++e) {
                const int32_t w = ci[e];
                if (label[w] < 0) { label[w] = label[v] + 1; order[tail++] = w; }
                if (label[w] == label[v] + 1) sigma[w] += sigma[v];
            }
        }
        for (int32_t i = tail - 1; i >= 0; --i) {
            const int32_t v = order[i];
            for (int32_t e = ro[v]; e < ro[v + 1]; ++e) {
                const int32_t w = ci[e];
                if (label[w] == label[v] + 1) delta[v] += sigma[v] / sigma[w] * (1.0 + delta[w]);
            }
            if (v != s) bc_out[v] += delta[v];
        }
        if (sigma_out) for (int32_t v = 0; v < nodes; ++v) sigma_out[v] = sigma[v];
    }
    for (int32_t v = 0; v < nodes; ++v) bc_out[v] *= 0.5;
    free(order); free(label); free(sigma); free(delta);
    return 0;
}

/* ---- PageRank: CPU restatement of the reference's GPU schedule (there is no CPU PageRank in its C-ABI path; its driver
 *      compares against Boost's page_rank, tests/pr/test_pr.cu, which is not installable here).
 *      gunrock/app/pr/pr_enactor.cuh:220-300  peeling of vertices without out-edges, round by round;
 *      pr_functor.cuh:52-71   an edge s->d with both ends alive moves rank[s] / degree[s] to d;
 *      pr_functor.cuh:84-93   rank = delta * sum + (1 - delta) * [vertex is the source, or source == -1]; a vertex is active
 *                             while |new - old| > threshold;
 *      pr_enactor.cuh:478-498 ranks of ALL vertices are replaced by the sums' array (peeled vertices end at 0); stop when no
 *                             vertex is active or after max_iter iterations (at least one iteration runs);
 *      pr_problem.cuh:423     initial rank 1 - delta.
 *      Accumulation here is double and in vertex order; the GPU sums floats in another order: compare with a tolerance.
 *      PARITY UNPINNED: the reference's ctest answer for this path ("Node ID 2: Page Rank 0.402378", CMakeLists.txt:231-233)
 *      is not what the code in the tree computes for shared_lib_tests/test_pr.c (node 2 converges to 0.398 and never passes
 *      through 0.402378); tests/test_oracle.py records the search. ---- */
int gro_pagerank(const int32_t *ro, const int32_t *ci, int32_t nodes, int32_t src, double delta, double threshold, int32_t max_iter,
                 double *rank_out, int32_t *degrees_out, int32_t *iterations_out)
{
    int32_t *deg = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nodes > 0 ? nodes : 1));
    int32_t *pong = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nodes > 0 ? nodes : 1));
    double *cur = (double *)malloc(sizeof(double) * (size_t)(nodes > 0 ? nodes : 1));
    double *nxt = (double *)malloc(sizeof(double) * (size_t)(nodes > 0 ? nodes : 1));
    if (!deg || !pong || !cur || !nxt) return -1;
    for (int32_t v = 0; v < nodes; ++v) deg[v] = ro[v + 1] - ro[v];
    for (;;) {                                       /* peeling rounds */
        int64_t removed = 0;
        for (int32_t v = 0; v < nodes; ++v) pong[v] = deg[v] == 0 ? -1 : deg[v];
        for (int32_t s = 0; s < nodes; ++s) {
            if (deg[s] <= 0) continue;               /* only queued vertices expand; a vertex at 0 is retired this round */
            for (int32_t e = ro[s]; e < ro[s + 1]; ++e)
                if (deg[ci[e]] == 0) pong[s] -= 1;
        }
        for (int32_t v = 0; v < nodes; ++v) if (deg[v] == 0) removed++;
        memcpy(deg, pong, sizeof(int32_t) * (size_t)nodes);
        if (!removed) break;
    }
    for (int32_t v = 0; v < nodes; ++v) { cur[v] = 1.0 - delta; nxt[v] = 0.0; }
    int32_t it = 0;
    int64_t alive = 0;
    for (int32_t v = 0; v < nodes; ++v) alive += deg[v] > 0;
    for (;;) {                                       /* `while (done[0] < 0)`, pr_enactor.cuh:341: the first pass always runs, even
                                                        over an empty queue (every vertex peeled: all ranks end at 0, iteration 1) */
        for (int32_t s = 0; s < nodes; ++s) {
            if (deg[s] <= 0) continue;
            const double c = (double)(float)((float)cur[s] / (float)deg[s]);   /* the GPU divides in float */
            for (int32_t e = ro[s]; e < ro[s + 1]; ++e)
                if (deg[ci[e]] > 0) nxt[ci[e]] += c;
        }
        int64_t active = 0;
        for (int32_t v = 0; v < nodes; ++v) {
            if (deg[v] <= 0) continue;
            nxt[v] = delta * nxt[v] + (1.0 - delta) * ((src == v || src == -1) ? 1.0 : 0.0);
            if (fabs(nxt[v] - cur[v]) > threshold) active++;
        }
        for (int32_t v = 0; v < nodes; ++v) { cur[v] = nxt[v]; nxt[v] = 0.0; }
        ++it;
        if (active == 0 || alive == 0 || it >= max_iter) break;
    }
    for (int32_t v = 0; v < nodes; ++v) rank_out[v] = cur[v];
    if (degrees_out) memcpy(degrees_out, deg, sizeof(int32_t) * (size_t)nodes);
    if (iterations_out) *iterations_out = it;
    free(deg); free(pong); free(cur); free(nxt);
    return 0;
}

/* ---- TopK degree centrality (gunrock/app/topk/topk_enactor.cuh:236-272): vertices by descending in + out degree; the
 *      reference's stable pair sort leaves ties in ascending vertex order. ---- */
void gro_topk(const int32_t *ro, const int32_t *co, int32_t nodes, int32_t k, int32_t *ids, int32_t *in_deg, int32_t *out_deg)
{
    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nodes > 0 ? nodes : 1));
    for (int32_t v = 0; v < nodes; ++v) order[v] = v;
    /* insertion into a sorted prefix of length k: O(n k), fine for a checker */
    for (int32_t i = 0; i < nodes; ++i) {
        const int32_t v = order[i];
        const int64_t tv = (int64_t)(ro[v + 1] - ro[v]) + (co ? co[v + 1] - co[v] : 0);
        int32_t j = i < k ? i : k;
        if (i >= k) {
            const int32_t w = order[k - 1];
            const int64_t tw = (int64_t)(ro[w + 1] - ro[w]) + (co ? co[w + 1] - co[w] : 0);
            if (tv <= tw) continue;
            j = k - 1;
        }
        while (j > 0) {
            const int32_t w = order[j - 1];
            const int64_t tw = (int64_t)(ro[w + 1] - ro[w]) + (co ? co[w + 1] - co[w] : 0);
            if (tw >= tv) break;
            order[j] = w;
            --j;
        }
        order[j] = v;
    }
    for (int32_t i = 0; i < k && i < nodes; ++i) {
        const int32_t v = order[i];
        ids[i] = v;
        out_deg[i] = ro[v + 1] - ro[v];
        in_deg[i] = co ? co[v + 1] - co[v] : 0;
    }
    free(order);
}
