/*
 * gr_oracle.h -- CPU oracle for the BFS / CC / SSSP (+ BC) frontier path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * host-side algorithms (graph ingest, R-MAT generator, CPU reference BFS,
 * Dijkstra, connected components, Brandes betweenness centrality), plus an
 * OpenMP BFS that exists only as the all-cores CPU baseline of bench.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product library
 * (gunrockinst_amd/csrc -> libgunrock.so) never links or calls it.
 *
 * Pinning (see DESIGN.md "Oracle"): checked against the reference's own
 * fixtures -- shared_lib_tests/test_bfs.c:32-33 (CSR of dataset/small/test_bc.mtx),
 * CMakeLists.txt:215-229 (known-answer regexes, BC's 0.500000 included), dataset/small/test_cc.mtx,
 * simple_example/bips98_606.mtx -- and against the golden values the survey
 * captured from the reference's host code, committed
 * under tests/golden/reference_goldens.json.
 *
 * All citations are relative to the reference tree.
 */
#ifndef GR_ORACLE_H
#define GR_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* COO tuple, value kept as long long exactly like the %lld parse (market.cuh:136-148). */
typedef struct {
    int32_t row;
    int32_t col;
    int64_t val;
} gro_tuple;

typedef struct {
    int32_t  nodes;
    int32_t  edges;          /* after dedup / self-loop removal (csr.cuh:311) */
    int32_t *row_offsets;    /* nodes + 1 */
    int32_t *col_indices;    /* edges */
    int32_t *edge_values;    /* edges (int Value; SSSP reinterprets as unsigned) */
} gro_csr;

void gro_csr_free(gro_csr *g);

/* G1: graphio::ReadMarketStream (market.cuh:56-215).  Returns 0 on success.
 * Entries are read "col row [val]" -> edge row-1 -> col-1 with row = SECOND number
 * (market.cuh:139-141,166-167), swapped when reversed && !undirected (:161-164);
 * undirected appends the mirrored tuple right after each entry (:173-184). */
int gro_read_market(const char *path, int undirected, int reversed,
                    gro_tuple **coo_out, int32_t *nodes_out, int32_t *tuples_out);

/* G2: Csr::FromCoo (csr.cuh:247-340) + RowFirstTupleCompare (coo.cuh:71-85).
 * Stable sort by (row, col), drop self loops and consecutive duplicates. `coo` is sorted in place. */
int gro_csr_from_coo(gro_tuple *coo, int32_t nodes, int32_t tuples, gro_csr *out);

/* Convenience: G1 + G2 (graphio::BuildMarketGraph, market.cuh:296-339; values always loaded :317,324,331). */
int gro_build_market(const char *path, int undirected, int reversed, gro_csr *out);

/* G3: graphio::BuildRmatGraph (rmat.cuh:27-91) on libc rand(), which the reference never
 * seeds (tests/bfs/test_bfs.cu:719-720).  Caller controls srand (call gro_srand(1) for the
 * "fresh process" stream). */
void gro_srand(unsigned seed);
int gro_rmat_reference(int32_t nodes, int32_t edges, int undirected,
                       double a, double b, double c, double d, gro_csr *out);

/* Own seeded counter-based R-MAT: same partition/noise rules as G3
 * (utils.cuh:58-130) but draws come from splitmix64(seed, edge, level, k).  NOT the libc stream.
 * Writes 2*pairs tuples when undirected (tuple i and pairs+i, rmat.cuh:71-81 layout). */
void gro_rmat_seeded_coo(int scale, int64_t pairs, uint64_t seed, int undirected,
                         double a, double b, double c, double d,
                         int64_t first, int64_t count, int32_t *rows, int32_t *cols);
int gro_rmat_seeded(int scale, int64_t pairs, uint64_t seed, int undirected,
                    double a, double b, double c, double d, gro_csr *out);

/* G4: Csr::GetNodeWithHighestDegree (csr.cuh:442-455), GetAverageDegree (csr.cuh:475-485). */
int32_t gro_highest_degree_node(const int32_t *row_offsets, int32_t nodes, int32_t *max_degree);
int32_t gro_average_degree(const int32_t *row_offsets, int32_t nodes);

/* X1: SimpleReferenceBfs (tests/bfs/test_bfs.cu:258-322).  labels = depth, -1 unreachable;
 * preds (may be NULL): -1 for source and unreached.  Returns the printed search depth (max label + 1). */
int32_t gro_bfs(const int32_t *row_offsets, const int32_t *col_indices, int32_t nodes,
                int32_t src, int32_t *labels, int32_t *preds);

/* X2: SimpleReferenceSssp (tests/sssp/test_sssp.cu:242-343) = Boost dijkstra_shortest_paths over
 * unsigned weights; Boost is absent here, restated as binary-heap Dijkstra.  dist = UINT_MAX when
 * unreachable (sssp_problem.cuh:325); preds (may be NULL) = own id for src/unreached (:363-372 iota init). */
void gro_sssp(const int32_t *row_offsets, const int32_t *col_indices, const uint32_t *weights,
              int32_t nodes, int32_t src, uint32_t *dist, int32_t *preds);

/* X3: RefCPUCC (tests/cc/test_cc.cu:183-202) = Boost connected_components (count only); restated as
 * union-find over edges treated as undirected; comp[v] = minimum vertex id of v's component
 * (SURVEY 8(a) C3 argument).  Returns the component count (#{v: comp[v]==v}, cc_problem.cuh:164-170). */
int32_t gro_cc(const int32_t *row_offsets, const int32_t *col_indices, int32_t nodes, int32_t *comp);

/* Sequential simulation of the reference CC schedule (cc_enactor.cuh:165-873 with the functors of
 * cc_functor.cuh:18-410): HookInit, PtrJump*, UpdateMask, {HookMax, PtrJumpMask*, PtrJumpUnmask, UpdateMask}*.
 * Counts edge sweeps (I_h) and vertex sweeps (I_j) for the roofline figure of SURVEY 8(d). */
int32_t gro_cc_reference_schedule(const int32_t *row_offsets, const int32_t *col_indices,
                                  int32_t nodes, int32_t *comp,
                                  int32_t *edge_sweeps, int32_t *vertex_sweeps);

/* DisplayStats (tests/bfs/test_bfs.cu:184-216): edges_visited = sum of out-degree over label > -1. */
void gro_bfs_stats(const int32_t *row_offsets, int32_t nodes, const int32_t *labels,
                   int64_t *nodes_visited, int64_t *edges_visited);

/* Validators (tests only): pred is a valid BFS parent / SSSP parent. Return number of violations. */
int64_t gro_check_bfs_preds(const int32_t *row_offsets, const int32_t *col_indices, int32_t nodes,
                            int32_t src, const int32_t *labels, const int32_t *preds);
int64_t gro_check_sssp_preds(const int32_t *row_offsets, const int32_t *col_indices,
                             const uint32_t *weights, int32_t nodes, int32_t src,
                             const uint32_t *dist, const int32_t *preds);

/* Level-synchronous BFS on `threads` host cores (0 = all), same labels as gro_bfs; returns the thread count used.
 * A stronger CPU baseline than the reference's serial loop, reported separately. */
int32_t gro_bfs_parallel(const int32_t *row_offsets, const int32_t *col_indices, int32_t nodes, int32_t src,
                         int32_t *labels, int32_t threads);

/* X5: Brandes betweenness centrality, doubles, halved like the reference's GPU drivers and Boost's undirected form
 * (tests/bc/test_bc.cu:144-300, bc_app.cu:112-113).  src = -1: all sources.  sigma_out (optional): path counts of the
 * last source. */
int gro_bc(const int32_t *row_offsets, const int32_t *col_indices, int32_t nodes, int32_t src,
           double *bc_out, double *sigma_out);

#ifdef __cplusplus
}
#endif
/* PageRank with the reference's GPU schedule (pr_enactor.cuh / pr_functor.cuh), doubles; degrees_out = out-degrees after
 * peeling (-1 peeled), iterations_out = iterations run.  PARITY UNPINNED (see gr_oracle.c). */
int gro_pagerank(const int32_t *row_offsets, const int32_t *col_indices, int32_t nodes, int32_t src, double delta, double threshold,
                 int32_t max_iter, double *rank_out, int32_t *degrees_out, int32_t *iterations_out);
/* TopK degree centrality (topk_enactor.cuh:236-272); col_offsets may be NULL (in-degrees 0) */
void gro_topk(const int32_t *row_offsets, const int32_t *col_offsets, int32_t nodes, int32_t k, int32_t *ids, int32_t *in_degrees,
              int32_t *out_degrees);

#endif
