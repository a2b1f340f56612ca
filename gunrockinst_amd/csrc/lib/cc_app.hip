// lib/cc_app.hip -- connected-components entry points of libgunrock.so.
//  * gunrock_cc_func: drop-in for the reference's C entry point (gunrock/app/cc/cc_app.cu:271-279 ->
//    dispatch_cc :193-262 -> run_cc :126-191): borrows the caller's CSR, returns malloc()ed component ids in
//    graph_out->node_values, prints "GPU Connected Component finished in ... msec." (:179).
//  * grx_cc_*: Problem / Enactor phases as separate C calls.
#include <gunrock/gunrock.h>
#include <gunrock/gunrock_mi355x.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include <gunrock/app/cc/cc_enactor.hpp>
#include <gunrock/app/cc/cc_problem.hpp>
#include <gunrock/csr.hpp>

using namespace gunrock;
using namespace gunrock::app;
using namespace gunrock::app::cc;

namespace {

struct CcRunner {
    virtual ~CcRunner() {}
    virtual hipError_t Init(const Csr<int, int, int> &g) = 0;
    virtual hipError_t InitDevice(int nodes, int edges, int *d_ro, int *d_ci) = 0;
    virtual hipError_t Reset() = 0;
    virtual hipError_t Enact(int max_grid_size, float *ms) = 0;
    virtual void Stats(long long &edge_sweeps, long long &vertex_sweeps, long long &launches, double &kernel_ms) = 0;
    virtual hipError_t Extract(int *ids, unsigned *num_components) = 0;
    virtual int *DeviceIds() = 0;
    virtual int Mirrored() = 0;
    virtual long long SweepEdges() = 0;
};

template <bool INSTR>
struct CcRunnerT : CcRunner {
    typedef CCProblem<int, int, int, true> Problem;
    Problem problem;
    CCEnactor<INSTR> enactor;
    hipEvent_t start = nullptr, stop = nullptr;
    explicit CcRunnerT(int device) : enactor(false)
    {
        util::GRError(hipSetDevice(device), "hipSetDevice failed", __FILE__, __LINE__);
        util::GRError(hipEventCreate(&start), "hipEventCreate failed", __FILE__, __LINE__);
        util::GRError(hipEventCreate(&stop), "hipEventCreate failed", __FILE__, __LINE__);
    }
    ~CcRunnerT() override
    {
        if (start) hipEventDestroy(start);
        if (stop) hipEventDestroy(stop);
    }
    hipError_t Init(const Csr<int, int, int> &g) override { return problem.Init(false, g, 1); }
    hipError_t InitDevice(int nodes, int edges, int *d_ro, int *d_ci) override
    {
        return problem.InitFromDevice(nodes, edges, d_ro, d_ci);
    }
    hipError_t Reset() override { return problem.Reset(enactor.GetFrontierType()); }
    hipError_t Enact(int max_grid_size, float *ms) override
    {
        hipStream_t stream = problem.graph_slices[0]->stream;
        hipError_t retval = hipSuccess;
        GR_CHECK(hipEventRecord(start, stream), "hipEventRecord failed");
        hipError_t run = enactor.template Enact<Problem>(&problem, max_grid_size);
        GR_CHECK(hipEventRecord(stop, stream), "hipEventRecord failed");
        GR_CHECK(hipEventSynchronize(stop), "hipEventSynchronize failed");
        float t = 0;
        GR_CHECK(hipEventElapsedTime(&t, start, stop), "hipEventElapsedTime failed");
        if (ms) *ms = t;
        return run;
    }
    void Stats(long long &es, long long &vs, long long &launches, double &kernel_ms) override
    {
        es = enactor.edge_sweeps;
        vs = enactor.vertex_sweeps;
        enactor.GetKernelStatistics(launches, kernel_ms);
    }
    hipError_t Extract(int *ids, unsigned *num_components) override
    {
        hipError_t rc = problem.Extract(ids);
        if (num_components) *num_components = problem.num_components;
        return rc;
    }
    int *DeviceIds() override { return problem.data_slices ? problem.data_slices[0]->d_component_ids : nullptr; }
    int Mirrored() override { return problem.data_slices ? problem.data_slices[0]->symmetric : 0; }
    long long SweepEdges() override { return problem.sweep_edges; }
};

}  // namespace

struct grx_cc {
    CcRunner *runner = nullptr;
};

extern "C" {

int grx_cc_create(grx_cc **out, int instrument, int device)
{
    if (!out) return -1;
    grx_cc *h = new grx_cc();
    h->runner = instrument ? static_cast<CcRunner *>(new CcRunnerT<true>(device)) : new CcRunnerT<false>(device);
    *out = h;
    return 0;
}

int grx_cc_init(grx_cc *p, int nodes, int edges, const int *row_offsets, const int *col_indices)
{
    if (!p || !row_offsets || nodes < 0 || edges < 0) return -1;
    Csr<int, int, int> wrap(false);
    wrap.nodes = nodes;
    wrap.edges = edges;
    wrap.row_offsets = const_cast<int *>(row_offsets);
    wrap.column_indices = const_cast<int *>(col_indices);
    hipError_t rc = p->runner->Init(wrap);
    wrap.row_offsets = nullptr;
    wrap.column_indices = nullptr;
    return static_cast<int>(rc);
}

int grx_cc_init_device(grx_cc *p, int nodes, int edges, int *d_row_offsets, int *d_col_indices)
{
    if (!p || !d_row_offsets || nodes < 0 || edges < 0) return -1;
    return static_cast<int>(p->runner->InitDevice(nodes, edges, d_row_offsets, d_col_indices));
}

int grx_cc_reset(grx_cc *p) { return p ? static_cast<int>(p->runner->Reset()) : -1; }

int grx_cc_enact(grx_cc *p, int max_grid_size, float *elapsed_ms)
{
    return p ? static_cast<int>(p->runner->Enact(max_grid_size, elapsed_ms)) : -1;
}

int grx_cc_stats(grx_cc *p, long long *edge_sweeps, long long *vertex_sweeps, long long *kernel_launches, double *kernel_ms)
{
    if (!p) return -1;
    long long es = 0, vs = 0, l = 0;
    double k = 0;
    p->runner->Stats(es, vs, l, k);
    if (edge_sweeps) *edge_sweeps = es;
    if (vertex_sweeps) *vertex_sweeps = vs;
    if (kernel_launches) *kernel_launches = l;
    if (kernel_ms) *kernel_ms = k;
    return 0;
}

int grx_cc_mirrored(grx_cc *p, int *mirrored)
{
    if (!p || !mirrored) return -1;
    *mirrored = p->runner->Mirrored();
    return 0;
}

int grx_cc_sweep_edges(grx_cc *p, long long *edges)
{
    if (!p || !edges) return -1;
    *edges = p->runner->SweepEdges();
    return 0;
}

int grx_cc_extract(grx_cc *p, int *h_component_ids, unsigned *num_components)
{
    if (!p || !h_component_ids) return -1;
    return static_cast<int>(p->runner->Extract(h_component_ids, num_components));
}

int grx_cc_device_results(grx_cc *p, int **d_component_ids)
{
    if (!p || !d_component_ids) return -1;
    *d_component_ids = p->runner->DeviceIds();
    return 0;
}

void grx_cc_destroy(grx_cc *p)
{
    if (!p) return;
    delete p->runner;
    delete p;
}

void gunrock_cc_func(struct GunrockGraph *graph_out, const struct GunrockGraph *graph_in, struct GunrockConfig configs,
                     struct GunrockDataType data_type)
{
    if (!graph_out || !graph_in) return;
    if (data_type.VTXID_TYPE != VTXID_INT || data_type.SIZET_TYPE != SIZET_INT) return;
    if (data_type.VALUE_TYPE != VALUE_INT) {
        std::printf("Not Yet Support This DataType Combination.\n");  // cc_app.cu:243-252
        return;
    }
    Csr<int, int, int> csr(false);
    csr.nodes = static_cast<int>(graph_in->num_nodes);
    csr.edges = static_cast<int>(graph_in->num_edges);
    csr.row_offsets = static_cast<int *>(graph_in->row_offsets);
    csr.column_indices = static_cast<int *>(graph_in->col_indices);

    int *h_ids = static_cast<int *>(std::malloc(sizeof(int) * static_cast<size_t>(csr.nodes > 0 ? csr.nodes : 1)));
    // the reference test leaves configs.device uninitialised only for fields it does not use; device IS set (test_cc.c:22)
    CcRunnerT<false> runner(configs.device);
    float elapsed = 0;
    unsigned components = 0;
    hipError_t rc = util::GRError(runner.Init(csr), "CC Problem Initialization Failed", __FILE__, __LINE__);
    if (!rc) rc = util::GRError(runner.Reset(), "CC Problem Data Reset Failed", __FILE__, __LINE__);
    if (!rc) rc = util::GRError(runner.Enact(0, &elapsed), "CC Problem Enact Failed", __FILE__, __LINE__);
    if (!rc) rc = util::GRError(runner.Extract(h_ids, &components), "CC Problem Data Extraction Failed", __FILE__, __LINE__);
    if (!rc && components > 0) {
        std::vector<int> roots(components);
        std::vector<unsigned> histogram(components);
        runner.problem.ComputeCCHistogram(h_ids, roots.data(), histogram.data());
    }
    graph_out->node_values = h_ids;  // caller frees (cc_app.cu:177)
    std::printf("GPU Connected Component finished in %lf msec.\n", elapsed);
    csr.row_offsets = nullptr;
    csr.column_indices = nullptr;
    util::GRError(hipDeviceSynchronize(), "hipDeviceSynchronize failed", __FILE__, __LINE__);
}

}  // extern "C"
