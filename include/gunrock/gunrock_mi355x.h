/*
 * gunrock/gunrock_mi355x.h -- handle-based C ABI over the engine's Problem / Enactor classes.
 *
 * gunrock.h's one-shot calls (upload, run, download, free) hide the phases the reference's own
 * drivers time separately (tests/bfs/test_bfs.cu:385-445: Init once, then Reset + timed Enact per
 * run, Extract, validate).  This header exposes those phases -- and the host graph builders the
 * drivers use -- as plain C entry points with plain pointers and sizes, so a foreign-language host
 * (ctypes / cgo / JNI) can keep a graph resident in HBM across runs.  Every function cites the
 * reference C++ interface it stands for.
 *
 * All functions return 0 on success, a positive hipError_t value on a HIP failure (already printed
 * to stderr in the reference's GRError format) or a negative value for a host-side error.
 * "d_" pointers are device (HBM) addresses; everything else is host memory.
 */
#ifndef GUNROCK_GUNROCK_MI355X_H_
#define GUNROCK_GUNROCK_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * Host graphs: gunrock::Csr<int,int,int> (reference gunrock/csr.cuh:38-80)
 * ---------------------------------------------------------------------------------------------- */
typedef struct grx_graph grx_graph;

/* graphio::BuildMarketGraph<true>(file, csr, undirected, reversed)  (reference graphio/market.cuh:296-339) */
int grx_graph_from_market(const char *path, int undirected, int reversed, grx_graph **out);
/* The same with the reference's CSR cache rule (market.cuh:296-339, csr.cuh:140-232: "<dir>/.<name>_{undirected,reversed,
 * nonreversed}_csr" written after the first parse and preferred afterwards), made safe: the cache is BINARY ("....bin"), and
 * is used only when it was written for a source file of exactly this size and modification time; *cache_hit (may be NULL)
 * tells which way the graph came.  An unwritable directory is not an error. */
int grx_graph_from_market_cached(const char *path, int undirected, int reversed, int *cache_hit, grx_graph **out);
/* graphio::BuildRmatGraph<true>(nodes, edges, csr, undirected, a,b,c,d) on libc rand() (reference graphio/rmat.cuh:27-91) */
int grx_graph_rmat_libc(int nodes, int edges, int undirected, double a, double b, double c, double d, grx_graph **out);
/* seeded counter-based R-MAT: 2^scale vertices, `pairs` generated edges (mirrored when undirected) */
int grx_graph_rmat_seeded(int scale, long long pairs, uint64_t seed, int undirected,
                          double a, double b, double c, double d, grx_graph **out);
/* Csr::FromCoo<true>(coo, nodes, tuples)  (reference csr.cuh:247-340): stable sort, drop self loops + repeats */
int grx_graph_from_coo(int nodes, long long tuples, const int *rows, const int *cols, const int *vals, grx_graph **out);
/* wrap existing CSR arrays (copied) -- what bfs_app.cu:256-260 does with the caller's pointers */
int grx_graph_from_csr(int nodes, int edges, const int *row_offsets, const int *col_indices, const int *edge_values,
                       grx_graph **out);
int grx_graph_nodes(const grx_graph *g);
int grx_graph_edges(const grx_graph *g);
const int *grx_graph_row_offsets(const grx_graph *g);
const int *grx_graph_col_indices(const grx_graph *g);
const int *grx_graph_edge_values(const grx_graph *g);   /* NULL when the graph carries no values */
/* Csr::GetNodeWithHighestDegree (reference csr.cuh:442-455) / GetAverageDegree (csr.cuh:475-485) */
int grx_graph_highest_degree_node(grx_graph *g, int *max_degree);
int grx_graph_average_degree(grx_graph *g);
/* graphio::RandomNode (reference graphio/utils.cuh:38-45) */
int grx_random_node(int num_nodes);
void grx_graph_free(grx_graph *g);

/* Device-side seeded R-MAT tuple generation (same stream as grx_graph_rmat_seeded): writes `count` tuples
 * starting at generated-edge index `first` into d_rows / d_cols.  `stream` is a hipStream_t or NULL. */
int grx_rmat_seeded_device(int scale, long long first, long long count, uint64_t seed,
                           double a, double b, double c, double d, int *d_rows, int *d_cols, void *stream);

/* Device-side COO -> CSR with Csr::FromCoo's graph semantics (reference csr.cuh:247-340: stable sort by (row, col), self
 * loops and duplicates dropped, trailing empty rows kept; `undirected` mirrors every tuple first, as the reference's
 * loaders do, market.cuh:172-183).  Hand-written LSD radix sort + device-wide scan, all in HBM.
 *   sort : d_rows / d_cols = `pairs` tuples on the device; rows = CSR rows to produce, nodes = vertex id space.
 *          parts > 1 builds one rank's slice of a vertex-cut partition: only tuples whose source v has v mod parts == rank
 *          are kept and stored under local row v div parts (ownership rule of reference problem_base.cuh:185-210).
 *          Returns the number of CSR edges in *edges_out so the caller can allocate.
 *   emit : writes row_offsets[rows + 1] and col_indices[edges] into caller-owned device arrays.
 * Tuples must stay below 2^31 (SIZET_INT).  `stream` is a hipStream_t or NULL. */
typedef struct grx_coo2csr grx_coo2csr;
int grx_coo_to_csr_sort(grx_coo2csr **handle, int rows, int nodes, long long pairs, const int *d_rows, const int *d_cols,
                        int undirected, int parts, int rank, long long *edges_out, void *stream);
int grx_coo_to_csr_emit(grx_coo2csr *handle, int *d_row_offsets, int *d_col_indices, void *stream);
void grx_coo_to_csr_free(grx_coo2csr *handle);

/* ------------------------------------------------------------------------------------------------
 * BFS: BFSProblem + BFSEnactor (reference gunrock/app/bfs/bfs_problem.cuh:41-364, bfs_enactor.cuh:40-708)
 * ---------------------------------------------------------------------------------------------- */
typedef struct grx_bfs grx_bfs;

/* picks the <MARK_PREDECESSORS, ENABLE_IDEMPOTENCE> instantiation like dispatch_bfs (reference bfs_app.cu:299-348);
 * `instrument` selects BFSEnactor<true>: per-launch HIP-event timing of the operator kernels */
int grx_bfs_create(grx_bfs **out, int mark_pred, int idempotence, int instrument, int device);
/* BFSProblem::Init(false, csr, 1) (reference bfs_problem.cuh:188-261): uploads the CSR */
int grx_bfs_init(grx_bfs *p, int nodes, int edges, const int *row_offsets, const int *col_indices);
/* same, for a CSR that already lives in HBM (pointers are borrowed for the life of the handle) */
int grx_bfs_init_device(grx_bfs *p, int nodes, int edges, int *d_row_offsets, int *d_col_indices);
/* Enable direction-optimizing traversal (reference app/dobfs: DOBFSProblem::Init takes the graph AND its inverse,
 * dobfs_problem.cuh; alpha/beta as in tests/dobfs/test_dobfs.cu:530-534).  d_inv_* is the in-neighbour CSR in HBM
 * (borrowed); pass NULL, NULL when the graph is undirected/symmetric to reuse the forward arrays.  alpha/beta <= 0
 * keep the defaults.  Takes effect with traversal_mode = 2 in grx_bfs_enact.  Call after init. */
int grx_bfs_set_inverse_graph(grx_bfs *p, const int *d_inv_row_offsets, const int *d_inv_col_indices,
                              float alpha, float beta);
/* What gunrock_bfs_func does before its search: when the CSR is its own inverse (every edge mirrored; checked on the device)
 * the graph serves as its own in-neighbour lists; otherwise, with build_if_directed != 0, the transpose is built on the device and
 * owned by the handle (the reference's DOBFS driver builds the inverse graph on the host, tests/dobfs/test_dobfs.cu; its enactor
 * takes both, dobfs_enactor.cuh:397,569).  *enabled: traversal_mode 2 is available; *built: a transpose was built; *build_ms:
 * what that cost.  Any out pointer may be NULL.  Call after init. */
int grx_bfs_auto_inverse(grx_bfs *p, int build_if_directed, int *enabled, int *built, float *build_ms);
/* Enactor tuning knobs (the reference exposes alpha/beta on its DOBFS command line, tests/dobfs/test_dobfs.cu:530-534):
 * alpha, beta as above; lite_factor: a top-down level runs without queue output when frontier_edges*alpha*lite_factor
 * exceeds the unexplored edges (0 disables); tail_edge_limit: levels with at most this many edges run inside the
 * single-workgroup multi-level kernel (0 disables).  Non-positive alpha/beta and negative other values keep the current
 * setting.  Results do not depend on any of these. */
int grx_bfs_set_tuning(grx_bfs *p, float alpha, float beta, float lite_factor, int tail_edge_limit);
/* Top-down levels with more than tail_edge_limit and at most `edge_limit` edges run inside the persistent multi-workgroup
 * kernel (one resident workgroup per CU, grid barrier between levels; 0 disables).  This is what the reference's
 * traversal_mode 1 (TWC advance for low-degree, high-diameter graphs, tests/bfs/test_bfs.cu:563-566) is for. */
int grx_bfs_set_persistent_limit(grx_bfs *p, int edge_limit);
/* traversal_mode 1 / graphs of average degree <= 8: a top-down frontier of at most 4096 vertices and `edge_limit` edges (default
 * 8192; 0 = never) runs in the TWC workgroup -- thread / wave / workgroup tiers by neighbour-list length (reference
 * edge_map_forward/cta.cuh:224-545), frontier kept in LDS, level after level inside one launch -- until a level outgrows it. */
int grx_bfs_set_twc_limit(grx_bfs *p, int edge_limit);
/* Launch that kernel with hipLaunchCooperativeKernel (the runtime then refuses a grid that exceeds the occupancy query at
 * launch time; costs ~15-19 us of host time per launch).  Off by default: a plain launch of the same grid has the same
 * residency, and the grid barrier's timeout word reports a lost co-residency at run time either way. */
int grx_bfs_set_cooperative_launch(grx_bfs *p, int on);
/* Top-down levels with at least `min_edges` frontier edges run as a destination-binned advance: expand + status screen,
 * claims on the destination's owner XCD without atomics, then a vertex-ordered closing sweep that labels and enqueues
 * (0 = never; default 2^23).  This replaces the per-edge atomicCAS of the reference's functor (bfs_functor.cuh:56-58) on the
 * levels where it dominates.  Results do not depend on it. */
int grx_bfs_set_binned_min_edges(grx_bfs *p, long long min_edges);
/* Deferred labels of a direction-optimizing search (DESIGN.md 3.3 i): sweeps that find vertices in vertex order keep their
 * output bitmap and ONE pass at the end of Enact writes every label.  enabled: 1 / 0, -1 = leave; mask_limit: frontier
 * bitmaps a search may hold before it flushes the kept ones into the labels (4..12, 0 = leave; tests shrink it).  Takes
 * effect at the next grx_bfs_reset.  Labels are identical either way. */
int grx_bfs_set_label_deferral(grx_bfs *p, int enabled, int mask_limit);
/* Enactor tuning by name (returns 1 for an unknown name): "emit_queue_factor" (a compacting bottom-up sweep also writes its
 * finds as the next top-down queue when its input frontier is within this factor of the switch-back threshold),
 * "sparse_sweep_div", "speculative_emit" (1/0: the label pass is queued behind the closing top-down launch without waiting
 * for its read-back), "chain_sweeps" (bottom-up sweeps queued per host round trip).  Results never depend on them. */
int grx_bfs_set_option(grx_bfs *p, const char *name, double value);
/* Direction-optimizing only: a level that would run count-only or bottom-up and has between `min_edges` and `max_edges`
 * frontier edges starts with a bottom-up pass that probes only the adjacency heads (the highest-degree in-neighbours); the
 * count-only top-down advance then handles what is left.  -1 = automatic bounds (edges/30 .. edges/7.8), min 0 = never,
 * max 0 = no upper bound.  Results do not depend on it. */
int grx_bfs_set_head_pass(grx_bfs *p, int min_edges, int max_edges);
/* BFSProblem::Reset(src, frontier_type, queue_sizing) (reference bfs_problem.cuh:272-360) */
int grx_bfs_reset(grx_bfs *p, int src, double queue_sizing);
/* BFSEnactor::Enact(context, problem, src, max_grid_size, traversal_mode) (reference bfs_enactor.cuh:573-579);
 * traversal_mode 0 = load-balanced top-down, 1 = low-degree/high-diameter choice of the reference driver (its TWC
 * advance; here LB advance + persistent mid-size levels kernel), 2 = direction-optimizing;
 * bracketed by HIP events on the problem's stream like the reference's GpuTimer (test_bfs.cu:408-438) */
int grx_bfs_enact(grx_bfs *p, int src, int max_grid_size, int traversal_mode, float *elapsed_ms);
/* BFSEnactor::GetStatistics (reference bfs_enactor.cuh:173-186) plus, when instrumented, operator-kernel
 * launch count and summed kernel time of the last Enact */
int grx_bfs_stats(grx_bfs *p, long long *total_queued, long long *search_depth, double *avg_duty,
                  long long *kernel_launches, double *kernel_ms);
/* instrumented enactors only: per-BSP-iteration record of the last Enact (input frontier length, its edge count,
 * operator kernel milliseconds, operator kind: 0 = top-down advance, 1 = bottom-up advance).  Fills up to
 * max_levels entries and returns the number of iterations recorded.  The `--v` per-iteration printout of the
 * reference driver (bfs_enactor.cuh:333-338) in data form. */
int grx_bfs_level_trace(grx_bfs *p, int max_levels, long long *frontier, long long *edges, double *ms, int *kind);
/* BFSProblem::Extract(h_labels, h_preds) (reference bfs_problem.cuh:144-177); h_preds may be NULL */
int grx_bfs_extract(grx_bfs *p, int *h_labels, int *h_preds);
/* device result arrays (valid until destroy / next init) */
int grx_bfs_device_results(grx_bfs *p, int **d_labels, int **d_preds);
void grx_bfs_destroy(grx_bfs *p);
/* The compacting filter operator by itself: reference filter::Kernel (gunrock/oprtr/filter/kernel.cuh:211-383) with the BFS
 * functor, whose CondFilter keeps valid vertex ids (bfs_functor.cuh:100-105).  d_in[n] in HBM, -1 = culled entry.
 * d_row_offsets != NULL: the output is a complete vertex frontier for the load-balanced advance -- id, first edge and the
 * exclusive prefix of the degrees in OUTPUT order; vertices without out-edges are dropped -- else ids only (row_start / scan
 * unused).  out_len entries were written (order unspecified), *out_edges = sum of their degrees.  Returns a HIP error code
 * ("Frontier queue overflow" when capacity is too small, filter/cta.cuh:526-529). */
int grx_filter_queue(int n, const int *d_in, const int *d_row_offsets, int capacity, int *d_out_v, int *d_out_row_start, int *d_out_scan,
                     int *out_len, long long *out_edges, int max_grid_size);

/* DisplayStats' counters (reference tests/bfs/test_bfs.cu:184-196): visited vertices and the sum of their
 * out-degrees -- the numerator of MTEPS = edges_visited / (elapsed_ms * 1000) */
void grx_bfs_count_visited(int nodes, const int *row_offsets, const int *labels,
                           long long *nodes_visited, long long *edges_visited);

/* ------------------------------------------------------------------------------------------------
 * CC: CCProblem + CCEnactor (reference gunrock/app/cc/cc_problem.cuh:36-440, cc_enactor.cuh:36-919)
 * ---------------------------------------------------------------------------------------------- */
typedef struct grx_cc grx_cc;

int grx_cc_create(grx_cc **out, int instrument, int device);
/* CCProblem::Init(false, csr, 1) (reference cc_problem.cuh:221-345) */
int grx_cc_init(grx_cc *p, int nodes, int edges, const int *row_offsets, const int *col_indices);
int grx_cc_init_device(grx_cc *p, int nodes, int edges, int *d_row_offsets, int *d_col_indices);
/* CCProblem::Reset(frontier_type) (reference cc_problem.cuh:361-440) */
int grx_cc_reset(grx_cc *p);
/* CCEnactor::Enact(problem, max_grid_size) (reference cc_enactor.cuh:889-919), HIP-event timed */
int grx_cc_enact(grx_cc *p, int max_grid_size, float *elapsed_ms);
/* sweeps of the last Enact: edge sweeps (hooks) and vertex sweeps (jumps / mask updates) -- I_h and I_j of the
 * roofline figure -- plus, when instrumented, kernel launches and their summed time */
int grx_cc_stats(grx_cc *p, long long *edge_sweeps, long long *vertex_sweeps, long long *kernel_launches, double *kernel_ms);
/* CCProblem::Extract(h_component_ids) (reference cc_problem.cuh:144-175); num_components = #{v: id[v] == v} */
/* *mirrored = 1 when Init found every edge (f, t), f < t, mirrored by (t, f): the hooking sweeps then park that orientation on first
 * sight (its mirror performs the identical root comparison), i.e. from the second edge sweep on half of the edges are skipped */
int grx_cc_mirrored(grx_cc *p, int *mirrored);
/* edges the hooking sweeps run over: all of them, or -- mirrored input -- only the from > to orientation of every edge, which the
 * problem materialises once at init (the other orientation performs the identical hooks) */
int grx_cc_sweep_edges(grx_cc *p, long long *edges);
int grx_cc_extract(grx_cc *p, int *h_component_ids, unsigned *num_components);
int grx_cc_device_results(grx_cc *p, int **d_component_ids);
void grx_cc_destroy(grx_cc *p);

/* ------------------------------------------------------------------------------------------------
 * SSSP: SSSPProblem + SSSPEnactor (reference gunrock/app/sssp/sssp_problem.cuh:35-387, sssp_enactor.cuh:36-563)
 * ---------------------------------------------------------------------------------------------- */
typedef struct grx_sssp grx_sssp;

/* mark_pred selects SSSPProblem<..., MARK_PATHS = true> (reference tests/sssp/test_sssp.cu:588-633) */
int grx_sssp_create(grx_sssp **out, int mark_pred, int instrument, int device);
/* SSSPProblem::Init(false, csr, 1, delta_factor) (reference sssp_problem.cuh:185-288); weights are unsigned 32-bit */
int grx_sssp_init(grx_sssp *p, int nodes, int edges, const int *row_offsets, const int *col_indices,
                  const unsigned *edge_weights, int delta_factor);
/* CSR + weights already in HBM (borrowed); `delta` is the bucket width to use (0 = one bucket per distance) */
int grx_sssp_init_device(grx_sssp *p, int nodes, int edges, int *d_row_offsets, int *d_col_indices,
                         const unsigned *d_edge_weights, float delta);
/* Enable PULL relaxation of dense levels: a level whose frontier has more than `pull_min_edges` out-edges (-1: 3/4 of the
 * edges, 0: never) takes, for every vertex, the minimum over its IN-edges of (neighbour's distance + weight) with the reducing
 * advance -- no atomicMin per edge -- instead of pushing along the frontier's out-edges (sssp_functor.cuh:54-64).  Needs the
 * in-neighbour lists WITH the weight of every in-edge, in HBM (borrowed); pass NULLs to have the weighted transpose built on
 * the device.  Distances do not depend on it.  Call after init. */
int grx_sssp_set_inverse_graph(grx_sssp *p, const int *d_inv_row_offsets, const int *d_inv_col_indices, const unsigned *d_inv_weights,
                               long long pull_min_edges);
/* levels of the last Enact that were relaxed by pulling */
int grx_sssp_pull_levels(grx_sssp *p, long long *levels);
/* SSSPProblem::Reset(src, frontier_type, queue_sizing) (reference sssp_problem.cuh:299-377) */
int grx_sssp_reset(grx_sssp *p, int src, double queue_sizing);
/* SSSPEnactor::Enact(context, problem, src, queue_sizing, max_grid_size) (reference sssp_enactor.cuh:485-563) */
int grx_sssp_enact(grx_sssp *p, int src, int max_grid_size, float *elapsed_ms);
/* work done by the last Enact: vertices dequeued, edge slots relaxed, BSP iterations; instrumented: kernel launches
 * and summed kernel time; the bucket width in use */
int grx_sssp_stats(grx_sssp *p, long long *relaxed_vertices, long long *relaxed_edges, long long *iterations,
                   long long *kernel_launches, double *kernel_ms, float *delta);
/* SSSPProblem::Extract(h_labels, h_preds): unsigned distances (UINT_MAX unreachable); preds = own id for the source and
 * unreached vertices (reference iota initialisation, sssp_problem.cuh:363-372); h_preds may be NULL */
int grx_sssp_extract(grx_sssp *p, unsigned *h_distances, int *h_preds);
void grx_sssp_destroy(grx_sssp *p);

/* ------------------------------------------------------------------------------------------------
 * Vertex-partitioned multi-GPU BFS: the LOCAL (one GPU, one rank) steps.  The reference has no multi-GPU code
 * (gunrock/app/problem_base.cuh:336-338 is a TODO); ownership follows its striped rule owner = v mod parts,
 * local id = v div parts (problem_base.cuh:185-210).  Collectives are the caller's job (RCCL via torch.distributed in
 * gunrockinst_amd/multi_gpu.py); none of these calls communicates.
 * ---------------------------------------------------------------------------------------------- */
typedef struct grx_pbfs grx_pbfs;

int grx_pbfs_create(grx_pbfs **out, int device);
/* local CSR in HBM (borrowed): rows = owned vertices in local-id order, column ids GLOBAL */
int grx_pbfs_init_device(grx_pbfs *p, int n_global, int parts, int rank, int n_local, int m_local,
                         int *d_row_offsets, int *d_col_indices);
/* labels = -1, bitmaps = 0; the owner of `src` seeds its frontier (BFSProblem::Reset role, bfs_problem.cuh:272-360) */
int grx_pbfs_reset(grx_pbfs *p, int src);
/* current local frontier: vertices with out-edges and the sum of their degrees */
int grx_pbfs_frontier(grx_pbfs *p, unsigned *len, unsigned *edges);
/* top-down, before the exchange: advance over the local frontier; every destination not forwarded before by this rank
 * is bucketed by owner.  h_send_counts[parts] = ids per destination rank; *d_send_buffer = the ids as LOCAL ids of their
 * owner, segments in rank order (what all_to_all_single wants). */
int grx_pbfs_advance_local(grx_pbfs *p, unsigned *h_send_counts, int **d_send_buffer);
/* top-down, after the exchange: the filter operator claims + labels the received local ids (first arrival wins) and
 * builds the next local frontier; returns its length / edge count */
int grx_pbfs_filter_received(grx_pbfs *p, const int *d_recv, int n_recv, unsigned *next_len, unsigned *next_edges);
/* direction-optimizing: local queue -> local frontier bitmap (entering bottom-up) */
int grx_pbfs_queue_to_bitmap(grx_pbfs *p);
/* the local frontier bitmap to all-gather: `words` 32-bit words, identical on every rank */
int grx_pbfs_frontier_bitmap(grx_pbfs *p, unsigned **d_bitmap, int *words);
/* bottom-up level over the owned unvisited vertices against the all-gathered bitmaps (parts x words_per_rank words) */
int grx_pbfs_bottom_up(grx_pbfs *p, const unsigned *d_gathered, int words_per_rank, unsigned *found, unsigned *found_edges);
/* local frontier bitmap -> local queue (leaving bottom-up) */
int grx_pbfs_bitmap_to_queue(grx_pbfs *p, unsigned *len, unsigned *edges);
/* device pointer to the local labels (depth per owned vertex, local-id order, -1 unreached) */
int grx_pbfs_labels(grx_pbfs *p, int **d_labels);
/* device pointer to the local predecessors (GLOBAL id of a valid BFS parent per owned vertex; -1 source, -2 unreached);
 * top-down discoveries carry a parent only when mark_pred was set (grx_pbfs_set_options) */
int grx_pbfs_preds(grx_pbfs *p, int **d_preds);

/* ---- the whole level loop inside the library: one call per search, the halo exchange through RCCL over xGMI ----
 * The reference's enactor is a host loop around operator launches (bfs_enactor.cuh:208-553); this is that loop for the
 * vertex-partitioned problem, with the per-level exchange of SURVEY 8(e) issued from C++ on the engine's stream:
 *   top-down level : advance -> bucket by owner -> ncclAllGather of the P x P count matrix -> grouped ncclSend/ncclRecv of
 *                    the ids (and, with mark_pred, of their parents) -> filter (claim, label, next frontier)
 *                    -> ncclAllGather of the packed frontier tails (termination + direction rule);
 *   bottom-up level: ONE ncclAllGather of the per-rank frontier bitmaps whose trailing word carries each rank's frontier
 *                    size -> local sweep; the search stays bottom-up to the end once Beamer's edge rule fires.
 * grx_rccl_unique_id: rank 0 creates the 128-byte ncclUniqueId, the caller distributes it (any channel) and every rank calls
 * grx_pbfs_comm_init_rccl after grx_pbfs_init_device.  RCCL is dlopen'ed on first use (librccl.so.1).
 * grx_rccl_load only loads the library (0 = every symbol resolved): ranks agree on its outcome BEFORE any of them enters
 * ncclCommInitRank, which is collective and would block the ranks that did load while the others have already given up. */
int grx_rccl_load(void);
int grx_rccl_unique_id(char id[128]);
int grx_pbfs_comm_init_rccl(grx_pbfs *p, const char id[128]);
/* the same loop with the three exchanges performed by the caller, synchronously, on device pointers (tests run several
 * ranks on one GPU over gloo this way); counts and offsets are in 32-bit words, segments in rank order */
typedef int (*grx_all_gather_fn)(void *ctx, const void *d_send, void *d_recv, size_t words_per_rank);
typedef int (*grx_all_to_all_v_fn)(void *ctx, const void *d_send, const size_t *send_counts, const size_t *send_offsets,
                                   void *d_recv, const size_t *recv_counts, const size_t *recv_offsets);
int grx_pbfs_set_transport(grx_pbfs *p, void *ctx, grx_all_gather_fn all_gather, grx_all_to_all_v_fn all_to_all_v);
/* mark_pred: exchange (id, parent) pairs on top-down levels; alpha: direction rule factor (<= 0 keeps the default, 30) */
int grx_pbfs_set_options(grx_pbfs *p, int mark_pred, float alpha);
/* Reset + the whole search from `src` (a GLOBAL vertex id, the same on every rank); levels = BSP levels executed;
 * elapsed_ms = device time of this rank from the reset to the last level (HIP events on the engine's stream) */
int grx_pbfs_search(grx_pbfs *p, int src, int direction_optimizing, int *levels, float *elapsed_ms);
/* Tuning of the level loop by name (1 = unknown name): "lite_factor" (a top-down level runs count-only -- destinations marked with
 * byte stores, ONE all-to-all of per-owner bitmap slices, then bottom-up to the end -- when global frontier edges * alpha *
 * lite_factor > unexplored edges and no parents are wanted; 0 = never), "alpha", "sparse_sweep_div".  grx_pbfs_stat:
 * "marked_levels" = count-only levels run since the handle was created (-1 = unknown name). */
int grx_pbfs_set_option(grx_pbfs *p, const char *name, double value);
long long grx_pbfs_stat(grx_pbfs *p, const char *name);
void grx_pbfs_destroy(grx_pbfs *p);

/* library / build identification: returns a static string such as "gunrock-mi355x gfx950 ..." */
/* ------------------------------------------------------------------------------------------------
 * BC: BCProblem + BCEnactor (reference gunrock/app/bc/bc_problem.cuh:36-485, bc_enactor.cuh:36-634), the instantiation of
 * the reference's C entry point: <int, int, float>, MARK_PREDECESSORS (bc_app.cu:61-66).
 * ------------------------------------------------------------------------------------------------ */
typedef struct grx_bc grx_bc;
int grx_bc_create(grx_bc **out, int device);
/* BCProblem::Init (bc_problem.cuh:203-330): host CSR in / device CSR borrowed */
int grx_bc_init(grx_bc *p, int nodes, int edges, const int *row_offsets, const int *col_indices);
int grx_bc_init_device(grx_bc *p, int nodes, int edges, int *d_row_offsets, int *d_col_indices);
/* The driver loop of run_bc (bc_app.cu:86-113): bc_values = 0; for src (or, when src == -1, every vertex in turn)
 * Reset + Enact; bc_values *= 0.5.  elapsed_ms = device time of the whole loop. */
int grx_bc_run(grx_bc *p, int src, int max_grid_size, double queue_sizing, float *elapsed_ms);
/* BCProblem::Extract (bc_problem.cuh:140-192); sigmas are those of the LAST source; any pointer may be NULL */
int grx_bc_extract(grx_bc *p, float *h_sigmas, float *h_bc_values, float *h_ebc_values);
void grx_bc_destroy(grx_bc *p);

/* ------------------------------------------------------------------------------------------------
 * PageRank: PRProblem + PREnactor (reference gunrock/app/pr/pr_problem.cuh:36-467, pr_enactor.cuh:36-622), the <int, float, int>
 * instantiation of its C entry point (pr_app.cu:213-296).  Ranks are pulled over the in-neighbour lists by the reducing
 * advance (the reference's R_TYPE / R_OP advance + SegReduceCsr, advance/kernel.cuh:733-761), so the problem needs the
 * inverse graph.
 * ------------------------------------------------------------------------------------------------ */
typedef struct grx_pr grx_pr;
int grx_pr_create(grx_pr **out, int device);
/* PRProblem::Init (pr_problem.cuh:186-307): host CSR in / device CSR borrowed */
int grx_pr_init(grx_pr *p, int nodes, int edges, const int *row_offsets, const int *col_indices);
int grx_pr_init_device(grx_pr *p, int nodes, int edges, int *d_row_offsets, int *d_col_indices);
/* in-neighbour lists: device CSC arrays (borrowed); or NULL, NULL with build_if_null = 0 for a symmetric graph (its CSR is its
 * own inverse) or build_if_null != 0 to have the transpose built on the device.  Call after init. */
int grx_pr_set_inverse_graph(grx_pr *p, const int *d_inv_row_offsets, const int *d_inv_col_indices, int build_if_null);
/* PRProblem::Reset(src, delta, threshold, frontier_type) (pr_problem.cuh:316-457); src = -1: every vertex teleports */
int grx_pr_reset(grx_pr *p, int src, float delta, float threshold);
/* PREnactor::Enact(context, problem, max_iteration, traversal_mode, max_grid_size) (pr_enactor.cuh:536-618), HIP-event timed */
int grx_pr_enact(grx_pr *p, int max_iter, int max_grid_size, float *elapsed_ms);
/* iterations run, peeling rounds (vertices without out-edges are removed round by round first, pr_enactor.cuh:220-300) and the
 * number of vertices left after peeling */
int grx_pr_stats(grx_pr *p, long long *iterations, long long *peeling_rounds, long long *surviving_nodes);
/* PRProblem::Extract (pr_problem.cuh:139-175): the first `count` ranks in descending order with their vertex ids
 * (count < 0: all); either pointer may be NULL */
int grx_pr_extract(grx_pr *p, float *h_rank_sorted, int *h_node_ids, int count);
/* device arrays: ranks indexed by vertex, vertex ids by descending rank */
int grx_pr_device_results(grx_pr *p, float **d_rank_by_vertex, int **d_node_ids_by_rank);
void grx_pr_destroy(grx_pr *p);

const char *grx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GUNROCK_GUNROCK_MI355X_H_ */
