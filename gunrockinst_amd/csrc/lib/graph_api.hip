// lib/graph_api.hip -- C ABI for host graphs and builders (include/gunrock/gunrock_mi355x.h).
#include <gunrock/gunrock_mi355x.h>

#include <cstdint>
#include <cstring>
#include <new>

#include <gunrock/csr.hpp>
#include <gunrock/graphio/market.hpp>
#include <gunrock/graphio/rmat.hpp>
#include <gunrock/graphio/device_csr.hpp>
#include <gunrock/graphio/rmat_device.hpp>
#include <gunrock/graphio/utils.hpp>

using gunrock::Coo;
using gunrock::Csr;

struct grx_graph {
    Csr<int, int, int> csr;
    bool has_values = false;
};

extern "C" {

int grx_graph_from_market(const char *path, int undirected, int reversed, grx_graph **out)
{
    if (!path || !out) return -1;
    grx_graph *g = new (std::nothrow) grx_graph();
    if (!g) return -2;
    if (gunrock::graphio::BuildMarketGraph<true>(const_cast<char *>(path), g->csr, undirected != 0, reversed != 0)) {
        delete g;
        return -3;
    }
    g->has_values = true;
    *out = g;
    return 0;
}

int grx_graph_from_market_cached(const char *path, int undirected, int reversed, int *cache_hit, grx_graph **out)
{
    if (!path || !out) return -1;
    grx_graph *g = new (std::nothrow) grx_graph();
    if (!g) return -2;
    if (gunrock::graphio::BuildMarketGraphCached<true>(const_cast<char *>(path), g->csr, undirected != 0, reversed != 0, cache_hit)) {
        delete g;
        return -3;
    }
    g->has_values = true;
    *out = g;
    return 0;
}

int grx_graph_rmat_libc(int nodes, int edges, int undirected, double a, double b, double c, double d, grx_graph **out)
{
    if (!out) return -1;
    grx_graph *g = new (std::nothrow) grx_graph();
    if (!g) return -2;
    if (gunrock::graphio::BuildRmatGraph<true>(nodes, edges, g->csr, undirected != 0, a, b, c, d)) {
        delete g;
        return -3;
    }
    g->has_values = true;
    *out = g;
    return 0;
}

int grx_graph_rmat_seeded(int scale, long long pairs, uint64_t seed, int undirected, double a, double b, double c,
                          double d, grx_graph **out)
{
    if (!out) return -1;
    grx_graph *g = new (std::nothrow) grx_graph();
    if (!g) return -2;
    if (gunrock::graphio::BuildSeededRmatGraph<true>(scale, pairs, seed, g->csr, undirected != 0, a, b, c, d)) {
        delete g;
        return -3;
    }
    g->has_values = true;
    *out = g;
    return 0;
}

int grx_graph_from_coo(int nodes, long long tuples, const int *rows, const int *cols, const int *vals, grx_graph **out)
{
    if (!out || nodes < 0 || tuples < 0 || tuples > 0x7fffffffLL || (tuples > 0 && (!rows || !cols))) return -1;
    typedef Coo<int, int> Tuple;
    Tuple *coo = static_cast<Tuple *>(std::malloc(sizeof(Tuple) * static_cast<size_t>(tuples > 0 ? tuples : 1)));
    if (!coo) return -2;
    for (long long i = 0; i < tuples; ++i) coo[i] = Tuple(rows[i], cols[i], vals ? vals[i] : 1);
    grx_graph *g = new (std::nothrow) grx_graph();
    if (!g) { std::free(coo); return -2; }
    g->csr.FromCoo<true>(nullptr, coo, nodes, static_cast<int>(tuples));
    g->has_values = true;
    std::free(coo);
    *out = g;
    return 0;
}

int grx_graph_from_csr(int nodes, int edges, const int *row_offsets, const int *col_indices, const int *edge_values,
                       grx_graph **out)
{
    if (!out || nodes < 0 || edges < 0 || !row_offsets || (edges > 0 && !col_indices)) return -1;
    grx_graph *g = new (std::nothrow) grx_graph();
    if (!g) return -2;
    if (edge_values) g->csr.FromScratch<true, false>(nodes, edges);
    else g->csr.FromScratch<false, false>(nodes, edges);
    std::memcpy(g->csr.row_offsets, row_offsets, sizeof(int) * (static_cast<size_t>(nodes) + 1));
    if (edges > 0) std::memcpy(g->csr.column_indices, col_indices, sizeof(int) * static_cast<size_t>(edges));
    if (edge_values && edges > 0) std::memcpy(g->csr.edge_values, edge_values, sizeof(int) * static_cast<size_t>(edges));
    g->has_values = edge_values != nullptr;
    *out = g;
    return 0;
}

int grx_graph_nodes(const grx_graph *g) { return g ? g->csr.nodes : -1; }
int grx_graph_edges(const grx_graph *g) { return g ? g->csr.edges : -1; }
const int *grx_graph_row_offsets(const grx_graph *g) { return g ? g->csr.row_offsets : nullptr; }
const int *grx_graph_col_indices(const grx_graph *g) { return g ? g->csr.column_indices : nullptr; }
const int *grx_graph_edge_values(const grx_graph *g) { return (g && g->has_values) ? g->csr.edge_values : nullptr; }

int grx_graph_highest_degree_node(grx_graph *g, int *max_degree)
{
    if (!g) return -1;
    int md = 0;
    int v = g->csr.GetNodeWithHighestDegree(md);
    if (max_degree) *max_degree = md;
    return v;
}

int grx_graph_average_degree(grx_graph *g) { return g ? g->csr.GetAverageDegree() : -1; }

int grx_random_node(int num_nodes) { return num_nodes > 0 ? gunrock::graphio::RandomNode(num_nodes) : -1; }

void grx_graph_free(grx_graph *g) { delete g; }

int grx_rmat_seeded_device(int scale, long long first, long long count, uint64_t seed, double a, double b, double c,
                           double d, int *d_rows, int *d_cols, void *stream)
{
    if (scale < 1 || scale > 30 || count < 0 || (count > 0 && (!d_rows || !d_cols))) return -1;
    return static_cast<int>(gunrock::graphio::SeededRmatDevice(scale, first, count, seed, a, b, c, d, d_rows, d_cols,
                                                                static_cast<hipStream_t>(stream)));
}

struct grx_coo2csr {
    gunrock::graphio::DeviceCooToCsr state;
};

int grx_coo_to_csr_sort(grx_coo2csr **handle, int rows, int nodes, long long pairs, const int *d_rows, const int *d_cols,
                        int undirected, int parts, int rank, long long *edges_out, void *stream)
{
    if (!handle || rows < 0 || nodes < 1 || pairs < 0 || (pairs > 0 && (!d_rows || !d_cols)) || parts < 1 || rank < 0 ||
        rank >= parts)
        return -1;
    const long long tuples = undirected ? 2 * pairs : pairs;
    if (tuples >= (1ll << 31)) return -2;  // SIZET_INT contract of the C ABI (gunrock.h:33-37)
    grx_coo2csr *h = new grx_coo2csr();
    hipError_t rc = h->state.Sort(rows, nodes, pairs, d_rows, d_cols, undirected != 0, parts, rank, static_cast<hipStream_t>(stream));
    if (rc != hipSuccess) {
        h->state.Release();
        delete h;
        return static_cast<int>(rc);
    }
    if (edges_out) *edges_out = h->state.edges;
    *handle = h;
    return 0;
}

int grx_coo_to_csr_emit(grx_coo2csr *h, int *d_row_offsets, int *d_col_indices, void *stream)
{
    if (!h || !d_row_offsets || (h->state.edges > 0 && !d_col_indices)) return -1;
    return static_cast<int>(h->state.Emit(d_row_offsets, d_col_indices, static_cast<hipStream_t>(stream)));
}

void grx_coo_to_csr_free(grx_coo2csr *h)
{
    if (!h) return;
    h->state.Release();
    delete h;
}

void grx_bfs_count_visited(int nodes, const int *row_offsets, const int *labels, long long *nodes_visited,
                           long long *edges_visited)
{
    long long nv = 0, ev = 0;
    for (int v = 0; v < nodes; ++v) {
        if (labels[v] > -1) {
            ++nv;
            ev += row_offsets[v + 1] - row_offsets[v];
        }
    }
    if (nodes_visited) *nodes_visited = nv;
    if (edges_visited) *edges_visited = ev;
}

const char *grx_version(void) { return "gunrock-mi355x 0.1 (gfx950, wave64, HIP)"; }

}  // extern "C"
