// lib/bfs_app.hip -- BFS entry points of libgunrock.so.
//
//  * gunrock_bfs_func: drop-in for the reference's C entry point (gunrock/app/bfs/bfs_app.cu:383-396
//    -> dispatch_bfs :241-366 -> run_bfs :146-230), same source-selection rules (:271-294), same
//    ownership rules (caller's CSR borrowed, labels malloc()ed for the caller, :165,211,256-260,350-351),
//    same stdout statistics block (:103-118).
//  * grx_bfs_*: the Problem / Enactor phases as separate C calls (include/gunrock/gunrock_mi355x.h).
#include <gunrock/gunrock.h>
#include <gunrock/gunrock_mi355x.h>

#include <cstdio>
#include <cstdlib>

#include <gunrock/app/bfs/bfs_enactor.hpp>
#include <gunrock/app/bfs/bfs_problem.hpp>
#include <gunrock/graphio/device_sort.hpp>
#include <gunrock/graphio/symmetry.hpp>
#include <gunrock/oprtr/filter/kernel.hpp>
#include <gunrock/csr.hpp>
#include <gunrock/graphio/utils.hpp>
#include <gunrock/util/context.hpp>

using namespace gunrock;
using namespace gunrock::app;
using namespace gunrock::app::bfs;

namespace {

// Type-erased holder so one C handle serves the four <MARK_PREDECESSORS, ENABLE_IDEMPOTENCE> instantiations
// (reference bfs_app.cu:299-348 picks them with nested ifs).
struct BfsRunner {
    virtual ~BfsRunner() {}
    virtual hipError_t Init(const Csr<int, int, int> &g) = 0;
    virtual hipError_t InitDevice(int nodes, int edges, int *d_ro, int *d_ci) = 0;
    virtual hipError_t SetInverse(const int *d_iro, const int *d_ici, float alpha, float beta) = 0;
    virtual hipError_t AutoInverse(bool &enabled, bool &built, float &build_ms, bool build_if_directed = true) = 0;
    virtual void SetTuning(float alpha, float beta, float lite_factor, int tail_edge_limit) = 0;
    virtual void SetPersistentLimit(int limit) = 0;
    virtual void SetTwcLimit(int limit) = 0;
    virtual void SetCooperativeLaunch(bool on) = 0;
    virtual void SetBinnedMinEdges(long long min_edges) = 0;
    virtual void SetLabelDeferral(int enabled, int mask_limit) = 0;
    virtual int SetOption(const char *name, double value) = 0;
    virtual void SetHeadPass(int min_edges, int max_edges) = 0;
    virtual hipError_t Reset(int src, double queue_sizing) = 0;
    virtual hipError_t Enact(int src, int max_grid_size, int traversal_mode, float *ms) = 0;
    virtual void Stats(long long &queued, long long &depth, double &duty, long long &launches, double &kernel_ms) = 0;
    virtual int Trace(int max_levels, long long *frontier, long long *edges, double *ms, int *kind) = 0;
    virtual hipError_t Extract(int *labels, int *preds) = 0;
    virtual void DeviceResults(int **labels, int **preds) = 0;
};

template <bool PRED, bool IDEMP, bool INSTR>
struct BfsRunnerT : BfsRunner {
    typedef BFSProblem<int, int, int, PRED, IDEMP, (PRED && IDEMP)> Problem;
    util::DeviceContext context;
    Problem problem;
    BFSEnactor<INSTR> enactor;
    hipEvent_t start = nullptr, stop = nullptr;

    explicit BfsRunnerT(int device) : context(device), enactor(false)
    {
        util::GRError(hipEventCreate(&start), "hipEventCreate failed", __FILE__, __LINE__);
        util::GRError(hipEventCreate(&stop), "hipEventCreate failed", __FILE__, __LINE__);
    }
    ~BfsRunnerT() override
    {
        if (start) hipEventDestroy(start);
        if (stop) hipEventDestroy(stop);
        FreeInverse();
    }
    hipError_t Init(const Csr<int, int, int> &g) override { return problem.Init(false, g, 1); }
    hipError_t InitDevice(int nodes, int edges, int *d_ro, int *d_ci) override
    {
        return problem.InitFromDevice(nodes, edges, d_ro, d_ci);
    }
    hipError_t SetInverse(const int *d_iro, const int *d_ici, float alpha, float beta) override
    {
        if (!problem.data_slices) return hipErrorNotInitialized;
        if (!d_iro || !d_ici) return problem.InverseIsSelf(alpha, beta);
        return problem.SetInverseGraph(d_iro, d_ici, alpha, beta);
    }
    // One-shot callers (gunrock_bfs_func) hand over host arrays only.  Direction-optimizing traversal needs the in-neighbour
    // lists: the CSR itself when every edge has its mirror (checked on the device, graphio/symmetry.hpp), else its transpose,
    // built on the device (graphio::DeviceTransposeCsr) and owned by this runner -- the reference's DOBFS takes the inverse graph
    // from its caller (dobfs_enactor.cuh:397,569; DOBFSProblem::Init), whose driver builds it on the host from the same .mtx.
    // The CSC fields of GunrockGraph are NOT read: the reference's BFS ignores them and its own test leaves them uninitialised
    // (shared_lib_tests/test_bfs.c:36-42), so a drop-in must not dereference them.
    hipError_t AutoInverse(bool &enabled, bool &built, float &build_ms, bool build_if_directed) override
    {
        hipError_t retval = hipSuccess;
        enabled = false;
        built = false;
        build_ms = 0.f;
        if (!problem.data_slices || problem.nodes <= 0 || problem.edges <= 0) return retval;
        GraphSlice<int, int, int> *gs = problem.graph_slices[0];
        bool symmetric = false;
        GR_CHECK(graphio::DeviceIsSymmetric(problem.nodes, problem.edges, gs->d_row_offsets, gs->d_column_indices, gs->stream, symmetric),
                 "BFS symmetry check failed");
        if (symmetric) {
            GR_CHECK(problem.InverseIsSelf(), "BFS InverseIsSelf failed");
            enabled = true;
            return retval;
        }
        if (!build_if_directed) return retval;
        GR_CHECK(hipEventRecord(start, gs->stream), "hipEventRecord failed");
        FreeInverse();
        GR_CHECK(hipMalloc(&d_inv_row_offsets, sizeof(int) * (static_cast<size_t>(problem.nodes) + 1)), "BFS hipMalloc inverse offsets failed");
        GR_CHECK(hipMalloc(&d_inv_column_indices, sizeof(int) * static_cast<size_t>(problem.edges)), "BFS hipMalloc inverse columns failed");
        GR_CHECK(graphio::DeviceTransposeCsr(problem.nodes, problem.edges, gs->d_row_offsets, gs->d_column_indices, d_inv_row_offsets,
                                             d_inv_column_indices, gs->stream),
                 "BFS transpose failed");
        GR_CHECK(problem.SetInverseGraph(d_inv_row_offsets, d_inv_column_indices), "BFS SetInverseGraph failed");
        GR_CHECK(hipEventRecord(stop, gs->stream), "hipEventRecord failed");
        GR_CHECK(hipEventSynchronize(stop), "hipEventSynchronize failed");
        GR_CHECK(hipEventElapsedTime(&build_ms, start, stop), "hipEventElapsedTime failed");
        enabled = true;
        built = true;
        return retval;
    }
    void FreeInverse()
    {
        if (d_inv_row_offsets) util::GRError(hipFree(d_inv_row_offsets), "BFS hipFree inverse offsets failed", __FILE__, __LINE__);
        if (d_inv_column_indices) util::GRError(hipFree(d_inv_column_indices), "BFS hipFree inverse columns failed", __FILE__, __LINE__);
        d_inv_row_offsets = nullptr;
        d_inv_column_indices = nullptr;
    }
    int *d_inv_row_offsets = nullptr;     // in-neighbour CSR built by AutoInverse for a directed input (owned)
    int *d_inv_column_indices = nullptr;
    void SetPersistentLimit(int limit) override { problem.persistent_edge_limit = limit; }
    void SetTwcLimit(int limit) override { problem.twc_edge_limit = limit; }
    void SetCooperativeLaunch(bool on) override { problem.cooperative_launch = on; }
    void SetBinnedMinEdges(long long min_edges) override { problem.binned_min_edges = min_edges; }
    int SetOption(const char *name, double value) override
    {
        const std::string key(name ? name : "");
        if (key == "emit_queue_factor") problem.emit_queue_factor = static_cast<float>(value);
        else if (key == "sparse_sweep_div") problem.sparse_sweep_div = static_cast<int>(value);
        else if (key == "speculative_emit") problem.speculative_emit = value != 0.0;
        else if (key == "chain_sweeps") problem.chain_sweeps = static_cast<int>(value);
        else if (key == "chain_closing") problem.chain_closing = value != 0.0;
        else return 1;
        return 0;
    }
    void SetLabelDeferral(int enabled, int mask_limit) override
    {
        if (enabled >= 0) problem.defer_labels = enabled != 0;
        if (mask_limit > 0) problem.level_mask_limit = mask_limit;
    }
    void SetHeadPass(int min_edges, int max_edges) override
    {
        problem.head_pass_min_edges = min_edges;
        problem.head_pass_max_edges = max_edges;
    }
    void SetTuning(float alpha, float beta, float lite_factor, int tail_edge_limit) override
    {
        if (alpha > 0) problem.alpha = alpha;
        if (beta > 0) problem.beta = beta;
        if (lite_factor >= 0) problem.lite_factor = lite_factor;
        if (tail_edge_limit >= 0) problem.tail_edge_limit = tail_edge_limit;
    }
    hipError_t Reset(int src, double queue_sizing) override
    {
        return problem.Reset(src, enactor.GetFrontierType(), queue_sizing);
    }
    hipError_t Enact(int src, int max_grid_size, int traversal_mode, float *ms) override
    {
        hipStream_t stream = problem.graph_slices[0]->stream;
        hipError_t retval = hipSuccess;
        GR_CHECK(hipEventRecord(start, stream), "hipEventRecord failed");
        hipError_t run = enactor.template Enact<Problem>(context, &problem, src, max_grid_size, traversal_mode);
        GR_CHECK(hipEventRecord(stop, stream), "hipEventRecord failed");
        GR_CHECK(hipEventSynchronize(stop), "hipEventSynchronize failed");
        float t = 0;
        GR_CHECK(hipEventElapsedTime(&t, start, stop), "hipEventElapsedTime failed");
        if (ms) *ms = t;
        return run;
    }
    void Stats(long long &queued, long long &depth, double &duty, long long &launches, double &kernel_ms) override
    {
        enactor.GetStatistics(queued, depth, duty);
        enactor.GetKernelStatistics(launches, kernel_ms);
    }
    int Trace(int max_levels, long long *frontier, long long *edges, double *ms, int *kind) override
    {
        const auto &t = enactor.GetLevelTrace();
        int n = static_cast<int>(t.size()) < max_levels ? static_cast<int>(t.size()) : max_levels;
        for (int i = 0; i < n; ++i) {
            if (frontier) frontier[i] = t[i].frontier;
            if (edges) edges[i] = t[i].edges;
            if (ms) ms[i] = t[i].ms;
            if (kind) kind[i] = t[i].kind;
        }
        return static_cast<int>(t.size());
    }
    hipError_t Extract(int *labels, int *preds) override { return problem.Extract(labels, preds); }
    void DeviceResults(int **labels, int **preds) override
    {
        if (labels) *labels = problem.data_slices ? problem.data_slices[0]->d_labels : nullptr;
        if (preds) *preds = problem.data_slices ? problem.data_slices[0]->d_preds : nullptr;
    }
};

BfsRunner *MakeRunner(bool pred, bool idemp, bool instr, int device)
{
    if (instr) {
        if (pred) return idemp ? static_cast<BfsRunner *>(new BfsRunnerT<true, true, true>(device))
                               : new BfsRunnerT<true, false, true>(device);
        return idemp ? static_cast<BfsRunner *>(new BfsRunnerT<false, true, true>(device))
                     : new BfsRunnerT<false, false, true>(device);
    }
    if (pred) return idemp ? static_cast<BfsRunner *>(new BfsRunnerT<true, true, false>(device))
                           : new BfsRunnerT<true, false, false>(device);
    return idemp ? static_cast<BfsRunner *>(new BfsRunnerT<false, true, false>(device))
                 : new BfsRunnerT<false, false, false>(device);
}

// stdout block of the reference's DisplayStats (bfs_app.cu:76-118)
void DisplayStats(const char *name, int src, const int *h_labels, const Csr<int, int, int> &graph, double elapsed,
                  long long search_depth, long long total_queued, double avg_duty)
{
    long long nodes_visited = 0, edges_visited = 0;
    grx_bfs_count_visited(graph.nodes, graph.row_offsets, h_labels, &nodes_visited, &edges_visited);
    double redundant_work = 0.0;
    if (total_queued > 0 && edges_visited > 0)
        redundant_work = 100.0 * (static_cast<double>(total_queued) - edges_visited) / edges_visited;
    std::printf("[%s] finished.", name);
    if (nodes_visited < 5) {
        std::printf("Fewer than 5 vertices visited.\n");
        return;
    }
    const double m_teps = static_cast<double>(edges_visited) / (elapsed * 1000.0);
    std::printf("\nelapsed: %.3f ms, rate: %.3f MiEdges/s", elapsed, m_teps);
    if (search_depth != 0) std::printf(", search_depth: %lld", search_depth);
    if (avg_duty != 0) std::printf("\n avg CTA duty: %.2f%%", avg_duty * 100);
    std::printf("\nsource_node: %lld, nodes_visited: %lld, edges visited: %lld", static_cast<long long>(src),
                nodes_visited, edges_visited);
    if (total_queued > 0) std::printf(", total queued: %lld", total_queued);
    if (redundant_work > 0) std::printf(", redundant work: %.2f%%", redundant_work);
    std::printf("\n");
}

}  // namespace

struct grx_bfs {
    BfsRunner *runner = nullptr;
};

extern "C" {

int grx_bfs_create(grx_bfs **out, int mark_pred, int idempotence, int instrument, int device)
{
    if (!out) return -1;
    grx_bfs *h = new grx_bfs();
    h->runner = MakeRunner(mark_pred != 0, idempotence != 0, instrument != 0, device);
    *out = h;
    return 0;
}

int grx_bfs_init(grx_bfs *p, int nodes, int edges, const int *row_offsets, const int *col_indices)
{
    if (!p || !row_offsets || nodes < 0 || edges < 0) return -1;
    Csr<int, int, int> wrap(false);  // borrow the caller's arrays (bfs_app.cu:256-260)
    wrap.nodes = nodes;
    wrap.edges = edges;
    wrap.row_offsets = const_cast<int *>(row_offsets);
    wrap.column_indices = const_cast<int *>(col_indices);
    hipError_t rc = p->runner->Init(wrap);
    wrap.row_offsets = nullptr;  // do not free what we do not own (bfs_app.cu:350-351)
    wrap.column_indices = nullptr;
    return static_cast<int>(rc);
}

int grx_bfs_init_device(grx_bfs *p, int nodes, int edges, int *d_row_offsets, int *d_col_indices)
{
    if (!p || !d_row_offsets || nodes < 0 || edges < 0) return -1;
    return static_cast<int>(p->runner->InitDevice(nodes, edges, d_row_offsets, d_col_indices));
}

int grx_bfs_set_inverse_graph(grx_bfs *p, const int *d_inv_row_offsets, const int *d_inv_col_indices, float alpha,
                              float beta)
{
    if (!p) return -1;
    return static_cast<int>(p->runner->SetInverse(d_inv_row_offsets, d_inv_col_indices, alpha, beta));
}

int grx_bfs_auto_inverse(grx_bfs *p, int build_if_directed, int *enabled, int *built, float *build_ms)
{
    if (!p || !p->runner) return -1;
    bool on = false, made = false;
    float ms = 0.f;
    const hipError_t rc = p->runner->AutoInverse(on, made, ms, build_if_directed != 0);
    if (enabled) *enabled = on ? 1 : 0;
    if (built) *built = made ? 1 : 0;
    if (build_ms) *build_ms = ms;
    return static_cast<int>(rc);
}

int grx_bfs_set_head_pass(grx_bfs *p, int min_edges, int max_edges)
{
    if (!p || !p->runner || min_edges < -1 || max_edges < -1) return 1;
    p->runner->SetHeadPass(min_edges, max_edges);
    return 0;
}

int grx_bfs_set_option(grx_bfs *p, const char *name, double value)
{
    if (!p || !p->runner) return -1;
    return p->runner->SetOption(name, value);
}

int grx_bfs_set_label_deferral(grx_bfs *p, int enabled, int mask_limit)
{
    if (!p) return -1;
    p->runner->SetLabelDeferral(enabled, mask_limit);
    return 0;
}

int grx_bfs_set_binned_min_edges(grx_bfs *p, long long min_edges)
{
    if (!p || !p->runner || min_edges < 0) return 1;
    p->runner->SetBinnedMinEdges(min_edges);
    return 0;
}

int grx_bfs_set_cooperative_launch(grx_bfs *p, int on)
{
    if (!p || !p->runner) return 1;
    p->runner->SetCooperativeLaunch(on != 0);
    return 0;
}

int grx_bfs_set_persistent_limit(grx_bfs *p, int edge_limit)
{
    if (!p || !p->runner || edge_limit < 0) return 1;
    p->runner->SetPersistentLimit(edge_limit);
    return 0;
}

int grx_bfs_set_twc_limit(grx_bfs *p, int edge_limit)
{
    if (!p || !p->runner || edge_limit < 0) return 1;
    p->runner->SetTwcLimit(edge_limit);
    return 0;
}

int grx_bfs_set_tuning(grx_bfs *p, float alpha, float beta, float lite_factor, int tail_edge_limit)
{
    if (!p) return -1;
    p->runner->SetTuning(alpha, beta, lite_factor, tail_edge_limit);
    return 0;
}

int grx_bfs_reset(grx_bfs *p, int src, double queue_sizing)
{
    if (!p) return -1;
    return static_cast<int>(p->runner->Reset(src, queue_sizing));
}

int grx_bfs_enact(grx_bfs *p, int src, int max_grid_size, int traversal_mode, float *elapsed_ms)
{
    if (!p) return -1;
    return static_cast<int>(p->runner->Enact(src, max_grid_size, traversal_mode, elapsed_ms));
}

int grx_bfs_stats(grx_bfs *p, long long *total_queued, long long *search_depth, double *avg_duty,
                  long long *kernel_launches, double *kernel_ms)
{
    if (!p) return -1;
    long long q = 0, d = 0, l = 0;
    double duty = 0, kms = 0;
    p->runner->Stats(q, d, duty, l, kms);
    if (total_queued) *total_queued = q;
    if (search_depth) *search_depth = d;
    if (avg_duty) *avg_duty = duty;
    if (kernel_launches) *kernel_launches = l;
    if (kernel_ms) *kernel_ms = kms;
    return 0;
}

int grx_bfs_level_trace(grx_bfs *p, int max_levels, long long *frontier, long long *edges, double *ms, int *kind)
{
    if (!p || max_levels < 0) return -1;
    return p->runner->Trace(max_levels, frontier, edges, ms, kind);
}

int grx_bfs_extract(grx_bfs *p, int *h_labels, int *h_preds)
{
    if (!p || !h_labels) return -1;
    return static_cast<int>(p->runner->Extract(h_labels, h_preds));
}

int grx_bfs_device_results(grx_bfs *p, int **d_labels, int **d_preds)
{
    if (!p) return -1;
    p->runner->DeviceResults(d_labels, d_preds);
    return 0;
}

// The compacting filter operator on its own (reference filter::Kernel with the BFS functor, filter/kernel.cuh:211-383: CondFilter
// = "is a valid vertex id", bfs_functor.cuh:100-105): `n` queue entries in HBM, -1 = culled.  d_row_offsets != NULL: the output
// is a complete vertex frontier (id, first edge, exclusive degree prefix; vertices without out-edges are dropped);
// NULL: ids only.  *out_len / *out_edges: entries written and the sum of their degrees.  Output order is not specified.
int grx_filter_queue(int n, const int *d_in, const int *d_row_offsets, int capacity, int *d_out_v, int *d_out_row_start, int *d_out_scan,
                     int *out_len, long long *out_edges, int max_grid_size)
{
    typedef BFSProblem<int, int, int, false, false, false> Problem;
    typedef BFSFunctor<int, int, int, Problem> Functor;
    typedef oprtr::filter::KernelPolicy<256, 4, 8> Policy;
    if (n < 0 || !d_out_v || !out_len || (d_row_offsets && (!d_out_row_start || !d_out_scan))) return -1;
    hipError_t retval = hipSuccess;
    unsigned long long *d_tail = nullptr;
    int *d_overflow = nullptr;
    GR_CHECK(hipMalloc(&d_tail, sizeof(unsigned long long)), "grx_filter_queue hipMalloc failed");
    GR_CHECK(hipMalloc(&d_overflow, sizeof(int)), "grx_filter_queue hipMalloc failed");
    GR_CHECK(hipMemset(d_tail, 0, sizeof(unsigned long long)), "grx_filter_queue memset failed");
    GR_CHECK(hipMemset(d_overflow, 0, sizeof(int)), "grx_filter_queue memset failed");
    GR_CHECK(hipDeviceSynchronize(), "grx_filter_queue sync failed");
    oprtr::filter::FilterArgs<int, int> f;
    f.d_in = d_in;
    f.num_elements = n;
    f.out.v = d_out_v;
    f.out.row_start = d_out_row_start;
    f.out.scan = d_out_scan;
    f.out.capacity = capacity;
    f.d_tail_out = d_tail;
    f.d_tail_clear = nullptr;
    f.d_overflow = d_overflow;
    f.d_row_offsets = d_row_offsets;
    Problem::DataSlice slice{};
    const int grid = max_grid_size > 0 ? max_grid_size : 2048;
    if (n > 0) {
        if (d_row_offsets) retval = oprtr::filter::LaunchKernel<Policy, Problem, Functor, true>(f, slice, grid, 0);
        else retval = oprtr::filter::LaunchKernel<Policy, Problem, Functor, false>(f, slice, grid, 0);
    }
    unsigned long long tail = 0;
    int overflow = 0;
    if (!retval) retval = util::GRError(hipMemcpy(&tail, d_tail, sizeof(tail), hipMemcpyDeviceToHost), "grx_filter_queue read failed", __FILE__, __LINE__);
    if (!retval) retval = util::GRError(hipMemcpy(&overflow, d_overflow, sizeof(int), hipMemcpyDeviceToHost), "grx_filter_queue read failed", __FILE__, __LINE__);
    hipFree(d_tail);
    hipFree(d_overflow);
    if (retval) return static_cast<int>(retval);
    if (overflow) return static_cast<int>(util::GRError(hipErrorInvalidConfiguration, "Frontier queue overflow. Please increase queue-sizing factor.", __FILE__, __LINE__));
    *out_len = static_cast<int>(util::TailCount(tail));
    if (out_edges) *out_edges = static_cast<long long>(util::TailEdges(tail));
    return 0;
}

void grx_bfs_destroy(grx_bfs *p)
{
    if (!p) return;
    delete p->runner;
    delete p;
}

void gunrock_bfs_func(struct GunrockGraph *graph_out, const struct GunrockGraph *graph_in, struct GunrockConfig configs,
                      struct GunrockDataType data_type)
{
    if (!graph_out || !graph_in) return;
    if (data_type.VTXID_TYPE != VTXID_INT || data_type.SIZET_TYPE != SIZET_INT) return;
    if (data_type.VALUE_TYPE != VALUE_INT) {
        std::printf("Not Yet Support This DataType Combination.\n");  // bfs_app.cu:354-365
        return;
    }
    Csr<int, int, int> csr(false);
    csr.nodes = static_cast<int>(graph_in->num_nodes);
    csr.edges = static_cast<int>(graph_in->num_edges);
    csr.row_offsets = static_cast<int *>(graph_in->row_offsets);
    csr.column_indices = static_cast<int *>(graph_in->col_indices);

    int src = 0;
    switch (configs.src_mode) {  // bfs_app.cu:271-294
        case randomize: src = graphio::RandomNode(csr.nodes); break;
        case largest_degree: { int md = 0; src = csr.GetNodeWithHighestDegree(md); break; }
        case manually: src = configs.src_node; break;
        default: src = 0; break;
    }
    const double queue_sizing = configs.queue_size > 0 ? configs.queue_size : 1.0;

    int *h_labels = static_cast<int *>(std::malloc(sizeof(int) * static_cast<size_t>(csr.nodes > 0 ? csr.nodes : 1)));
    BfsRunner *runner = MakeRunner(configs.mark_pred, configs.idempotence, false, configs.device);
    float elapsed = 0;
    hipError_t rc = runner->Init(csr);
    // The reference's entry point always runs its top-down enactor (bfs_app.cu:196-200).  Here the search runs direction-
    // optimizing (the reference's separate DOBFS primitive): on the graph itself when it is its own inverse, else on the
    // transpose built on the device -- same labels, a fraction of the edges looked at.  Like the reference's Init, the setup is
    // outside the Enact timer; its cost is printed.  Small graphs stay top-down: the in-neighbour tables would cost more than
    // the search.
    bool dobfs = false, built = false;
    float build_ms = 0.f;
    if (!rc && csr.edges >= (1 << 16))
        rc = util::GRError(runner->AutoInverse(dobfs, built, build_ms), "BFS inverse graph setup failed", __FILE__, __LINE__);
    if (!rc) rc = util::GRError(runner->Reset(src, queue_sizing), "BFS Problem Data Reset Failed", __FILE__, __LINE__);
    if (!rc) rc = util::GRError(runner->Enact(src, 0, dobfs ? 2 : 0, &elapsed), "BFS Problem Enact Failed", __FILE__, __LINE__);
    long long queued = 0, depth = 0, launches = 0;
    double duty = 0, kernel_ms = 0;
    runner->Stats(queued, depth, duty, launches, kernel_ms);
    if (!rc) rc = util::GRError(runner->Extract(h_labels, nullptr), "BFS Problem Data Extraction Failed", __FILE__, __LINE__);
    graph_out->node_values = h_labels;  // caller frees (bfs_app.cu:211)
    if (!rc && dobfs && !built) std::printf("[GPU Breadth-first search] symmetric input: direction-optimizing traversal.\n");
    if (!rc && dobfs && built)
        std::printf("[GPU Breadth-first search] directed input: direction-optimizing traversal on the device-built inverse graph (%.3f ms).\n",
                    build_ms);
    if (!rc) DisplayStats("GPU Breadth-first search", src, h_labels, csr, elapsed, depth, queued, duty);
    delete runner;
    csr.row_offsets = nullptr;
    csr.column_indices = nullptr;
    util::GRError(hipDeviceSynchronize(), "hipDeviceSynchronize failed", __FILE__, __LINE__);
}

}  // extern "C"
