/*
 * gunrock/gunrock.h -- C ABI of the MI355X frontier engine (libgunrock.so).
 *
 * Drop-in boundary: the type names, enumerator order, struct field order/types and the six
 * entry-point signatures below are byte-compatible with the reference's public header
 * (reference: gunrock/gunrock.h:25-151), so a C program written against the reference
 * (e.g. reference shared_lib_tests/test_bfs.c:12-68) compiles and links unchanged.
 *
 * All six entry points run on the GPU (BFS is the hot path, CC / SSSP / BC /
 * PageRank / TopK are the "next" rows built on the same operators).
 *
 * Calling contract (reference: gunrock/app/bfs/bfs_app.cu:146-396, cc_app.cu:126-279):
 *   - graph_in->row_offsets / col_indices (/ edge_values) are HOST arrays owned by the caller;
 *     they are only read during the call.
 *   - only {VTXID_INT, SIZET_INT} with VALUE_INT (BFS, CC, TopK), VALUE_UINT (SSSP) and VALUE_FLOAT
 *     (BC, PageRank) are implemented, the combinations the reference dispatches; other combinations print "Not Yet Support This DataType Combination." and return.
 *   - graph_out->node_values receives a malloc()ed array of num_nodes 32-bit values
 *     (BFS depth, -1 unreachable / CC component id = smallest vertex id of the component /
 *     SSSP unsigned distance, UINT_MAX unreachable); the CALLER frees it.
 *   - functions return void; failures are printed to stderr as
 *     "[file, line] message (HIP error N: text)" and the call continues or returns early.
 *   - not re-entrant across threads; sequential calls are fine.  The call is synchronous.
 */
#ifndef GUNROCK_GUNROCK_H_
#define GUNROCK_GUNROCK_H_

#include <stdlib.h>
#include <stdbool.h>

/* ---- data-type selectors (reference gunrock.h:25-46) ---- */
enum VertexIdType { VTXID_INT };
enum SizeTType { SIZET_INT };
enum ValueType { VALUE_INT, VALUE_UINT, VALUE_FLOAT };

struct GunrockDataType {          /* reference gunrock.h:51-56 */
    enum VertexIdType VTXID_TYPE;
    enum SizeTType    SIZET_TYPE;
    enum ValueType    VALUE_TYPE;
};

/* ---- graph exchange struct (reference gunrock.h:61-71) ---- */
struct GunrockGraph {
    size_t num_nodes;
    size_t num_edges;
    void  *row_offsets;   /* CSR: int[num_nodes + 1]        */
    void  *col_indices;   /* CSR: int[num_edges]            */
    void  *col_offsets;   /* CSC (unused by BFS / CC / SSSP) */
    void  *row_indices;   /* CSC (unused by BFS / CC / SSSP) */
    void  *node_values;   /* per-vertex output              */
    void  *edge_values;   /* per-edge input (SSSP weights)  */
};

/* ---- how the traversal source is picked (reference gunrock.h:76-81) ---- */
enum SrcMode {
    manually,        /* use GunrockConfig.src_node                               */
    randomize,       /* libc-rand() vertex, as graphio::RandomNode does           */
    largest_degree   /* first vertex of maximal out-degree                        */
};

/* ---- per-call options (reference gunrock.h:86-99) ---- */
struct GunrockConfig {
    bool  mark_pred;     /* BFS/SSSP: also compute predecessors                   */
    bool  idempotence;   /* BFS: idempotent traversal (same labels either way)    */
    int   src_node;      /* source when src_mode == manually                      */
    int   device;        /* GPU ordinal                                           */
    int   max_iter;      /* (BC/PR only)                                          */
    int   top_nodes;     /* (TopK/PR only)                                        */
    int   delta_factor;  /* SSSP: near/far bucket width multiplier                */
    float delta;         /* (PR only)                                             */
    float error;         /* (PR only)                                             */
    float queue_size;    /* frontier queue sizing factor                          */
    enum SrcMode src_mode;
};

#ifdef __cplusplus
extern "C" {
#endif

/* reference gunrock.h:106-110; implementation reference gunrock/app/bfs/bfs_app.cu:383-396 */
void gunrock_bfs_func(struct GunrockGraph *graph_out, const struct GunrockGraph *graph_in,
                      struct GunrockConfig configs, struct GunrockDataType data_type);

/* reference gunrock.h:113-117; implementation reference gunrock/app/bc/bc_app.cu:86-125,330-343:
 * VALUE_FLOAT; src_node = -1 accumulates over every source; node_values = float[num_nodes] (halved),
 * edge_values = float[num_edges] of zeros as in the reference */
void gunrock_bc_func(struct GunrockGraph *graph_out, const struct GunrockGraph *graph_in,
                     struct GunrockConfig configs, struct GunrockDataType data_type);

/* reference gunrock.h:120-124; implementation reference gunrock/app/cc/cc_app.cu:271-279 */
void gunrock_cc_func(struct GunrockGraph *graph_out, const struct GunrockGraph *graph_in,
                     struct GunrockConfig configs, struct GunrockDataType data_type);

/* reference gunrock.h:127-132 (declared there, its sssp_app.cu is stale and not built:
 * gunrock/CMakeLists.txt:27); semantics per shared_lib_tests/test_sssp.c:46-57:
 * `predecessor` is a caller-allocated int[num_nodes]. */
void gunrock_sssp_func(struct GunrockGraph *graph_out, void *predecessor,
                       const struct GunrockGraph *graph_in, struct GunrockConfig configs,
                       struct GunrockDataType data_type);

/* reference gunrock.h:135-141; implementation reference gunrock/app/pr/pr_app.cu:189-346: VALUE_FLOAT;
 * node_ids int[] / page_rank float[] are caller-allocated and receive min(num_nodes, top_nodes) entries
 * (all when top_nodes <= 0) in descending rank order.  graph_in's CSC fields are not read. */
void gunrock_pr_func(struct GunrockGraph *graph_out, void *node_ids, void *page_rank,
                     const struct GunrockGraph *graph_in, struct GunrockConfig configs,
                     struct GunrockDataType data_type);

/* reference gunrock.h:144-151; implementation reference gunrock/app/topk/topk_app.cu: VALUE_INT;
 * graph_in carries CSR and CSC (col_offsets / row_indices, shared_lib_tests/test_topk.c:30-41);
 * node_ids / in_degrees / out_degrees are caller-allocated int[top_nodes] */
void gunrock_topk_func(struct GunrockGraph *graph_out, void *node_ids, void *in_degrees,
                       void *out_degrees, const struct GunrockGraph *graph_in,
                       struct GunrockConfig configs, struct GunrockDataType data_type);

#ifdef __cplusplus
}
#endif

#endif /* GUNROCK_GUNROCK_H_ */
