// lib/sssp_app.hip -- SSSP entry points of libgunrock.so.
//  * gunrock_sssp_func: the C entry point declared by the reference (gunrock/gunrock.h:127-132).  The reference's own
//    implementation file is stale and excluded from its build (gunrock/app/sssp/sssp_app.cu:154-157,
//    gunrock/CMakeLists.txt:27), so semantics follow the declared signature, shared_lib_tests/test_sssp.c:12-73 and the
//    structure of the BFS/CC wrappers: unsigned 32-bit weights in graph_in->edge_values, unsigned distances malloc()ed into
//    graph_out->node_values (UINT_MAX = unreachable), caller-allocated int predecessor[num_nodes].
//  * grx_sssp_*: Problem / Enactor phases as separate C calls.
#include <gunrock/gunrock.h>
#include <gunrock/gunrock_mi355x.h>

#include <cstdio>
#include <cstdlib>

#include <gunrock/app/sssp/sssp_enactor.hpp>
#include <gunrock/app/sssp/sssp_problem.hpp>
#include <gunrock/csr.hpp>
#include <gunrock/graphio/utils.hpp>
#include <gunrock/util/context.hpp>

using namespace gunrock;
using namespace gunrock::app;
using namespace gunrock::app::sssp;

namespace {

struct SsspRunner {
    virtual ~SsspRunner() {}
    virtual hipError_t Init(Csr<int, int, int> &g, int delta_factor) = 0;
    virtual hipError_t InitDevice(int nodes, int edges, int *d_ro, int *d_ci, const unsigned *d_w, float delta) = 0;
    virtual hipError_t SetInverse(const int *d_iro, const int *d_ici, const unsigned *d_iw, long long pull_min_edges) = 0;
    virtual long long PullLevels() = 0;
    virtual hipError_t Reset(int src, double queue_sizing) = 0;
    virtual hipError_t Enact(int src, int max_grid_size, float *ms) = 0;
    virtual void Stats(long long &vertices, long long &edges, long long &iters, long long &launches, double &kernel_ms) = 0;
    virtual hipError_t Extract(unsigned *labels, int *preds) = 0;
    virtual float Delta() = 0;
    virtual unsigned *DeviceLabels() = 0;
};

template <bool PATHS, bool INSTR>
struct SsspRunnerT : SsspRunner {
    typedef SSSPProblem<int, int, int, PATHS> Problem;
    util::DeviceContext context;
    Problem problem;
    SSSPEnactor<INSTR> enactor;
    hipEvent_t start = nullptr, stop = nullptr;
    explicit SsspRunnerT(int device) : context(device), enactor(false)
    {
        util::GRError(hipEventCreate(&start), "hipEventCreate failed", __FILE__, __LINE__);
        util::GRError(hipEventCreate(&stop), "hipEventCreate failed", __FILE__, __LINE__);
    }
    ~SsspRunnerT() override
    {
        if (start) hipEventDestroy(start);
        if (stop) hipEventDestroy(stop);
    }
    hipError_t Init(Csr<int, int, int> &g, int delta_factor) override { return problem.Init(false, g, 1, delta_factor); }
    hipError_t InitDevice(int nodes, int edges, int *d_ro, int *d_ci, const unsigned *d_w, float delta) override
    {
        return problem.InitFromDevice(nodes, edges, d_ro, d_ci, d_w, delta);
    }
    hipError_t SetInverse(const int *d_iro, const int *d_ici, const unsigned *d_iw, long long pull_min_edges) override
    {
        if (!problem.data_slices) return hipErrorNotInitialized;
        problem.pull_min_edges = pull_min_edges;
        if (d_iro && d_ici && d_iw) return problem.SetInverseGraph(d_iro, d_ici, d_iw);
        return problem.BuildInverse();
    }
    long long PullLevels() override { return enactor.pull_levels; }
    hipError_t Reset(int src, double queue_sizing) override { return problem.Reset(src, enactor.GetFrontierType(), queue_sizing); }
    hipError_t Enact(int src, int max_grid_size, float *ms) override
    {
        hipStream_t stream = problem.graph_slices[0]->stream;
        hipError_t retval = hipSuccess;
        GR_CHECK(hipEventRecord(start, stream), "hipEventRecord failed");
        hipError_t run = enactor.template Enact<Problem>(context, &problem, src, 1.0, max_grid_size, 0);
        GR_CHECK(hipEventRecord(stop, stream), "hipEventRecord failed");
        GR_CHECK(hipEventSynchronize(stop), "hipEventSynchronize failed");
        float t = 0;
        GR_CHECK(hipEventElapsedTime(&t, start, stop), "hipEventElapsedTime failed");
        if (ms) *ms = t;
        return run;
    }
    void Stats(long long &vertices, long long &edges, long long &iters, long long &launches, double &kernel_ms) override
    {
        long long q;
        double duty;
        enactor.GetStatistics(q, iters, duty);
        vertices = enactor.relaxed_vertices;
        edges = enactor.relaxed_edges;
        enactor.GetKernelStatistics(launches, kernel_ms);
    }
    hipError_t Extract(unsigned *labels, int *preds) override { return problem.Extract(labels, preds); }
    float Delta() override { return problem.data_slices ? problem.data_slices[0]->delta : 0.0f; }
    unsigned *DeviceLabels() override { return problem.data_slices ? problem.data_slices[0]->d_labels : nullptr; }
};

SsspRunner *MakeRunner(bool paths, bool instr, int device)
{
    if (paths) return instr ? static_cast<SsspRunner *>(new SsspRunnerT<true, true>(device)) : new SsspRunnerT<true, false>(device);
    return instr ? static_cast<SsspRunner *>(new SsspRunnerT<false, true>(device)) : new SsspRunnerT<false, false>(device);
}

}  // namespace

struct grx_sssp {
    SsspRunner *runner = nullptr;
};

extern "C" {

int grx_sssp_create(grx_sssp **out, int mark_pred, int instrument, int device)
{
    if (!out) return -1;
    grx_sssp *h = new grx_sssp();
    h->runner = MakeRunner(mark_pred != 0, instrument != 0, device);
    *out = h;
    return 0;
}

int grx_sssp_init(grx_sssp *p, int nodes, int edges, const int *row_offsets, const int *col_indices,
                  const unsigned *edge_weights, int delta_factor)
{
    if (!p || !row_offsets || !edge_weights || nodes < 0 || edges < 0) return -1;
    Csr<int, int, int> wrap(false);
    wrap.nodes = nodes;
    wrap.edges = edges;
    wrap.row_offsets = const_cast<int *>(row_offsets);
    wrap.column_indices = const_cast<int *>(col_indices);
    wrap.edge_values = reinterpret_cast<int *>(const_cast<unsigned *>(edge_weights));
    hipError_t rc = p->runner->Init(wrap, delta_factor);
    wrap.row_offsets = nullptr;
    wrap.column_indices = nullptr;
    wrap.edge_values = nullptr;
    return static_cast<int>(rc);
}

int grx_sssp_init_device(grx_sssp *p, int nodes, int edges, int *d_row_offsets, int *d_col_indices,
                         const unsigned *d_edge_weights, float delta)
{
    if (!p || !d_row_offsets || !d_edge_weights || nodes < 0 || edges < 0) return -1;
    return static_cast<int>(p->runner->InitDevice(nodes, edges, d_row_offsets, d_col_indices, d_edge_weights, delta));
}

int grx_sssp_set_inverse_graph(grx_sssp *p, const int *d_inv_row_offsets, const int *d_inv_col_indices, const unsigned *d_inv_weights,
                               long long pull_min_edges)
{
    if (!p) return -1;
    return static_cast<int>(p->runner->SetInverse(d_inv_row_offsets, d_inv_col_indices, d_inv_weights, pull_min_edges));
}

int grx_sssp_pull_levels(grx_sssp *p, long long *levels)
{
    if (!p || !levels) return -1;
    *levels = p->runner->PullLevels();
    return 0;
}

int grx_sssp_reset(grx_sssp *p, int src, double queue_sizing) { return p ? static_cast<int>(p->runner->Reset(src, queue_sizing)) : -1; }

int grx_sssp_enact(grx_sssp *p, int src, int max_grid_size, float *elapsed_ms)
{
    return p ? static_cast<int>(p->runner->Enact(src, max_grid_size, elapsed_ms)) : -1;
}

int grx_sssp_stats(grx_sssp *p, long long *relaxed_vertices, long long *relaxed_edges, long long *iterations,
                   long long *kernel_launches, double *kernel_ms, float *delta)
{
    if (!p) return -1;
    long long v = 0, e = 0, it = 0, l = 0;
    double k = 0;
    p->runner->Stats(v, e, it, l, k);
    if (relaxed_vertices) *relaxed_vertices = v;
    if (relaxed_edges) *relaxed_edges = e;
    if (iterations) *iterations = it;
    if (kernel_launches) *kernel_launches = l;
    if (kernel_ms) *kernel_ms = k;
    if (delta) *delta = p->runner->Delta();
    return 0;
}

int grx_sssp_extract(grx_sssp *p, unsigned *h_distances, int *h_preds)
{
    if (!p || !h_distances) return -1;
    return static_cast<int>(p->runner->Extract(h_distances, h_preds));
}

void grx_sssp_destroy(grx_sssp *p)
{
    if (!p) return;
    delete p->runner;
    delete p;
}

void gunrock_sssp_func(struct GunrockGraph *graph_out, void *predecessor, const struct GunrockGraph *graph_in,
                       struct GunrockConfig configs, struct GunrockDataType data_type)
{
    if (!graph_out || !graph_in) return;
    if (data_type.VTXID_TYPE != VTXID_INT || data_type.SIZET_TYPE != SIZET_INT) return;
    if (data_type.VALUE_TYPE != VALUE_UINT) {
        std::printf("Not Yet Support This DataType Combination.\n");
        return;
    }
    Csr<int, int, int> csr(false);
    csr.nodes = static_cast<int>(graph_in->num_nodes);
    csr.edges = static_cast<int>(graph_in->num_edges);
    csr.row_offsets = static_cast<int *>(graph_in->row_offsets);
    csr.column_indices = static_cast<int *>(graph_in->col_indices);
    csr.edge_values = static_cast<int *>(graph_in->edge_values);

    int src = 0;
    switch (configs.src_mode) {  // same rules as BFS (bfs_app.cu:271-294)
        case randomize: src = graphio::RandomNode(csr.nodes); break;
        case largest_degree: { int md = 0; src = csr.GetNodeWithHighestDegree(md); break; }
        case manually: src = configs.src_node; break;
        default: src = 0; break;
    }
    const double queue_sizing = configs.queue_size > 0 ? configs.queue_size : 1.0;
    const bool mark_pred = configs.mark_pred && predecessor != nullptr;

    unsigned *h_dist = static_cast<unsigned *>(std::malloc(sizeof(unsigned) * static_cast<size_t>(csr.nodes > 0 ? csr.nodes : 1)));
    SsspRunner *runner = MakeRunner(mark_pred, false, configs.device);
    float elapsed = 0;
    hipError_t rc = util::GRError(runner->Init(csr, configs.delta_factor), "SSSP Problem Initialization Failed", __FILE__, __LINE__);
    if (!rc) rc = util::GRError(runner->Reset(src, queue_sizing), "SSSP Problem Data Reset Failed", __FILE__, __LINE__);
    if (!rc) rc = util::GRError(runner->Enact(src, 0, &elapsed), "SSSP Problem Enact Failed", __FILE__, __LINE__);
    if (!rc) rc = util::GRError(runner->Extract(h_dist, mark_pred ? static_cast<int *>(predecessor) : nullptr),
                                "SSSP Problem Data Extraction Failed", __FILE__, __LINE__);
    graph_out->node_values = h_dist;  // caller frees
    if (!rc) std::printf("GPU Single-Source Shortest Path finished in %lf msec. source: %d\n", elapsed, src);
    delete runner;
    csr.row_offsets = nullptr;
    csr.column_indices = nullptr;
    csr.edge_values = nullptr;
    util::GRError(hipDeviceSynchronize(), "hipDeviceSynchronize failed", __FILE__, __LINE__);
}

}  // extern "C"
