// lib/pr_app.hip -- PageRank and TopK (degree centrality) entry points of libgunrock.so.
//  * gunrock_pr_func: drop-in for the reference's C entry point (gunrock/app/pr/pr_app.cu:321-346 -> dispatch_page_rank
//    :189-306 -> run_page_rank :102-176): <int, float, int> only, source from src_mode, delta / error / max_iter from the
//    config, prints "[GPU PageRank] finished.  elapsed: ... ms" (:92-98).  Output: vertex ids and ranks in descending rank
//    order.  The reference copies ALL `nodes` entries into the caller's arrays (PRProblem::Extract, pr_problem.cuh:139-175)
//    although its own test allocates top_nodes of them (shared_lib_tests/test_pr.c:46-47): here min(nodes, top_nodes)
//    entries are written when top_nodes > 0, all of them otherwise.
//  * gunrock_topk_func: gunrock/app/topk/topk_app.cu: vertices by descending in-degree + out-degree (ties by ascending id,
//    the order of the reference's stable pair sort, topk_enactor.cuh:262-272), with both degrees; in-degrees from the
//    caller's CSC (col_offsets) as in shared_lib_tests/test_topk.c:30-31.
//  * grx_pr_*: Problem / Enactor phases as separate C calls on a graph that may already live in HBM.
#include <gunrock/gunrock.h>
#include <gunrock/gunrock_mi355x.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <gunrock/app/pr/pr_enactor.hpp>
#include <gunrock/app/pr/pr_problem.hpp>
#include <gunrock/csr.hpp>
#include <gunrock/graphio/symmetry.hpp>
#include <gunrock/graphio/utils.hpp>

using namespace gunrock;
using namespace gunrock::app;
using namespace gunrock::app::pr;

namespace {

struct PrRunner {
    typedef PRProblem<int, int, float> Problem;
    Problem problem;
    PREnactor<false> enactor;
    util::DeviceContext context;
    hipEvent_t start = nullptr, stop = nullptr;
    explicit PrRunner(int device) : enactor(false), context(device)
    {
        util::GRError(hipEventCreate(&start), "hipEventCreate failed", __FILE__, __LINE__);
        util::GRError(hipEventCreate(&stop), "hipEventCreate failed", __FILE__, __LINE__);
    }
    ~PrRunner()
    {
        if (start) hipEventDestroy(start);
        if (stop) hipEventDestroy(stop);
    }
    hipError_t Enact(int max_iter, int max_grid_size, float *ms)
    {
        hipStream_t stream = problem.graph_slices[0]->stream;
        hipError_t retval = hipSuccess;
        GR_CHECK(hipEventRecord(start, stream), "hipEventRecord failed");
        hipError_t run = enactor.template Enact<Problem>(context, &problem, max_iter, 0, max_grid_size);
        GR_CHECK(hipEventRecord(stop, stream), "hipEventRecord failed");
        GR_CHECK(hipEventSynchronize(stop), "hipEventSynchronize failed");
        float t = 0;
        GR_CHECK(hipEventElapsedTime(&t, start, stop), "hipEventElapsedTime failed");
        if (ms) *ms = t;
        return run;
    }
};

// ---- TopK ----
__global__ void DegreeKeysKernel(const int *d_row_offsets, const int *d_col_offsets, long long nodes, unsigned long long *d_keys)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long v = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; v < nodes; v += stride) {
        const unsigned out = static_cast<unsigned>(d_row_offsets[v + 1] - d_row_offsets[v]);
        const unsigned in = d_col_offsets ? static_cast<unsigned>(d_col_offsets[v + 1] - d_col_offsets[v]) : 0u;
        d_keys[v] = (static_cast<unsigned long long>(~(out + in)) << 32) | static_cast<unsigned>(v);  // ascending key = descending total
    }
}
__global__ void DegreeUnpackKernel(const unsigned long long *d_keys, const int *d_row_offsets, const int *d_col_offsets, long long count,
                                   int *d_ids, int *d_in, int *d_out)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride) {
        const int v = static_cast<int>(d_keys[i] & 0xFFFFFFFFull);
        d_ids[i] = v;
        d_out[i] = d_row_offsets[v + 1] - d_row_offsets[v];
        d_in[i] = d_col_offsets ? d_col_offsets[v + 1] - d_col_offsets[v] : 0;
    }
}

}  // namespace

struct grx_pr {
    PrRunner *runner = nullptr;
};

extern "C" {

int grx_pr_create(grx_pr **out, int device)
{
    if (!out) return -1;
    grx_pr *h = new grx_pr();
    h->runner = new PrRunner(device);
    *out = h;
    return 0;
}

int grx_pr_init(grx_pr *p, int nodes, int edges, const int *row_offsets, const int *col_indices)
{
    if (!p || !row_offsets || nodes < 0 || edges < 0) return -1;
    Csr<int, float, int> wrap(false);
    wrap.nodes = nodes;
    wrap.edges = edges;
    wrap.row_offsets = const_cast<int *>(row_offsets);
    wrap.column_indices = const_cast<int *>(col_indices);
    hipError_t rc = p->runner->problem.Init(false, wrap, 1);
    wrap.row_offsets = nullptr;
    wrap.column_indices = nullptr;
    return static_cast<int>(rc);
}

int grx_pr_init_device(grx_pr *p, int nodes, int edges, int *d_row_offsets, int *d_col_indices)
{
    if (!p || !d_row_offsets || nodes < 0 || edges < 0) return -1;
    return static_cast<int>(p->runner->problem.InitFromDevice(nodes, edges, d_row_offsets, d_col_indices));
}

int grx_pr_set_inverse_graph(grx_pr *p, const int *d_inv_row_offsets, const int *d_inv_col_indices, int build_if_null)
{
    if (!p || !p->runner->problem.graph_slices) return -1;
    if (d_inv_row_offsets && d_inv_col_indices) {
        p->runner->problem.SetInverseGraph(d_inv_row_offsets, d_inv_col_indices);
        return 0;
    }
    if (build_if_null) return static_cast<int>(p->runner->problem.BuildInverse());
    p->runner->problem.InverseIsSelf();
    return 0;
}

int grx_pr_reset(grx_pr *p, int src, float delta, float threshold)
{
    if (!p) return -1;
    return static_cast<int>(p->runner->problem.Reset(src, delta, threshold, p->runner->enactor.GetFrontierType()));
}

int grx_pr_enact(grx_pr *p, int max_iter, int max_grid_size, float *elapsed_ms)
{
    if (!p) return -1;
    return static_cast<int>(p->runner->Enact(max_iter, max_grid_size, elapsed_ms));
}

int grx_pr_stats(grx_pr *p, long long *iterations, long long *peeling_rounds, long long *surviving_nodes)
{
    if (!p) return -1;
    long long queued = 0, iters = 0;
    double duty = 0;
    p->runner->enactor.GetStatistics(queued, duty, iters);
    if (iterations) *iterations = iters;
    if (peeling_rounds) *peeling_rounds = p->runner->enactor.PeelingRounds();
    if (surviving_nodes) *surviving_nodes = p->runner->enactor.SurvivingNodes();
    return 0;
}

int grx_pr_extract(grx_pr *p, float *h_rank_sorted, int *h_node_ids, int count)
{
    if (!p) return -1;
    return static_cast<int>(p->runner->problem.Extract(h_rank_sorted, h_node_ids, count));
}

int grx_pr_device_results(grx_pr *p, float **d_rank_by_vertex, int **d_node_ids_by_rank)
{
    if (!p || !p->runner->problem.data_slices) return -1;
    if (d_rank_by_vertex) *d_rank_by_vertex = p->runner->problem.data_slices[0]->d_rank_curr;
    if (d_node_ids_by_rank) *d_node_ids_by_rank = p->runner->problem.data_slices[0]->d_node_ids;
    return 0;
}

void grx_pr_destroy(grx_pr *p)
{
    if (!p) return;
    delete p->runner;
    delete p;
}

void gunrock_pr_func(struct GunrockGraph *graph_out, void *node_ids, void *page_rank, const struct GunrockGraph *graph_in,
                     struct GunrockConfig pr_config, struct GunrockDataType data_type)
{
    (void)graph_out;
    if (!graph_in || !node_ids || !page_rank) return;
    if (data_type.VTXID_TYPE != VTXID_INT || data_type.SIZET_TYPE != SIZET_INT) return;
    if (data_type.VALUE_TYPE != VALUE_FLOAT) {
        std::printf("Not Yet Support This DataType Combination.\n");  // pr_app.cu:201-210
        return;
    }
    Csr<int, float, int> csr(false);
    csr.nodes = static_cast<int>(graph_in->num_nodes);
    csr.edges = static_cast<int>(graph_in->num_edges);
    csr.row_offsets = static_cast<int *>(graph_in->row_offsets);
    csr.column_indices = static_cast<int *>(graph_in->col_indices);
    int src_node = -1;
    switch (pr_config.src_mode) {  // pr_app.cu:229-252
        case randomize: src_node = graphio::RandomNode(csr.nodes); break;
        case largest_degree: {
            int max_degree = 0;
            src_node = csr.GetNodeWithHighestDegree(max_degree);
            break;
        }
        case manually: src_node = pr_config.src_node; break;
        default: src_node = -1; break;
    }
    PrRunner runner(pr_config.device);
    float elapsed = 0;
    hipError_t rc = util::GRError(runner.problem.Init(false, csr, 1), "Page Rank Problem Initialization Failed", __FILE__, __LINE__);
    if (!rc) {
        // in-neighbour lists: the CSR itself when every edge has its mirror, else its transpose, built on the device.  The CSC
        // fields of GunrockGraph are not read: the reference's PageRank ignores them and its test leaves them uninitialised
        // (shared_lib_tests/test_pr.c:34-40).
        bool symmetric = false;
        GraphSlice<int, int, float> *gs = runner.problem.graph_slices[0];
        rc = util::GRError(graphio::DeviceIsSymmetric(csr.nodes, csr.edges, gs->d_row_offsets, gs->d_column_indices, gs->stream, symmetric),
                           "Page Rank symmetry check failed", __FILE__, __LINE__);
        if (!rc && symmetric) runner.problem.InverseIsSelf();
        else if (!rc) rc = util::GRError(runner.problem.BuildInverse(), "Page Rank transpose failed", __FILE__, __LINE__);
    }
    if (!rc) rc = util::GRError(runner.problem.Reset(src_node, pr_config.delta, pr_config.error, runner.enactor.GetFrontierType()),
                                "Page Rank Problem Data Reset Failed", __FILE__, __LINE__);
    if (!rc) rc = util::GRError(runner.Enact(pr_config.max_iter, 0, &elapsed), "Page Rank Problem Enact Failed", __FILE__, __LINE__);
    int count = csr.nodes;
    if (pr_config.top_nodes > 0 && pr_config.top_nodes < count) count = pr_config.top_nodes;
    if (!rc) rc = util::GRError(runner.problem.Extract(static_cast<float *>(page_rank), static_cast<int *>(node_ids), count),
                                "Page Rank Problem Data Extraction Failed", __FILE__, __LINE__);
    std::printf("[GPU PageRank] finished.  elapsed: %.3f ms\n", elapsed);
    csr.row_offsets = nullptr;
    csr.column_indices = nullptr;
    util::GRError(hipDeviceSynchronize(), "hipDeviceSynchronize failed", __FILE__, __LINE__);
}

void gunrock_topk_func(struct GunrockGraph *graph_out, void *node_ids, void *in_degrees, void *out_degrees,
                       const struct GunrockGraph *graph_in, struct GunrockConfig topk_config, struct GunrockDataType data_type)
{
    (void)graph_out;
    if (!graph_in || !node_ids || !in_degrees || !out_degrees) return;
    if (data_type.VTXID_TYPE != VTXID_INT || data_type.SIZET_TYPE != SIZET_INT || data_type.VALUE_TYPE != VALUE_INT) {
        std::printf("Not Yet Support This DataType Combination.\n");
        return;
    }
    if (util::GRError(hipSetDevice(topk_config.device), "hipSetDevice failed", __FILE__, __LINE__)) return;
    const int n = static_cast<int>(graph_in->num_nodes);
    int count = topk_config.top_nodes;
    if (count > n) count = n;  // (topk_app.cu clamps the same way before Extract)
    if (n <= 0 || count <= 0) return;
    int *d_ro = nullptr, *d_co = nullptr, *d_out3 = nullptr;
    hipStream_t stream = 0;
    graphio::DeviceKeySort sorter;
    hipError_t rc = hipMalloc(&d_ro, sizeof(int) * (static_cast<size_t>(n) + 1));
    if (!rc) rc = hipMemcpy(d_ro, graph_in->row_offsets, sizeof(int) * (static_cast<size_t>(n) + 1), hipMemcpyHostToDevice);
    if (!rc && graph_in->col_offsets) {
        rc = hipMalloc(&d_co, sizeof(int) * (static_cast<size_t>(n) + 1));
        if (!rc) rc = hipMemcpy(d_co, graph_in->col_offsets, sizeof(int) * (static_cast<size_t>(n) + 1), hipMemcpyHostToDevice);
    }
    if (!rc) rc = hipMalloc(&d_out3, sizeof(int) * 3 * static_cast<size_t>(count));
    if (!rc) rc = sorter.Reserve(n);
    unsigned long long *sorted = nullptr;
    if (!rc) {
        hipLaunchKernelGGL(DegreeKeysKernel, dim3(1024), dim3(256), 0, stream, d_ro, d_co, static_cast<long long>(n), sorter.Keys());
        rc = hipGetLastError();
    }
    if (!rc) rc = sorter.Sort(n, 64, stream, &sorted);
    if (!rc) {
        hipLaunchKernelGGL(DegreeUnpackKernel, dim3(256), dim3(256), 0, stream, sorted, d_ro, d_co, static_cast<long long>(count), d_out3,
                           d_out3 + count, d_out3 + 2 * count);
        rc = hipGetLastError();
    }
    if (!rc) rc = hipMemcpy(node_ids, d_out3, sizeof(int) * count, hipMemcpyDeviceToHost);
    if (!rc) rc = hipMemcpy(in_degrees, d_out3 + count, sizeof(int) * count, hipMemcpyDeviceToHost);
    if (!rc) rc = hipMemcpy(out_degrees, d_out3 + 2 * count, sizeof(int) * count, hipMemcpyDeviceToHost);
    util::GRError(rc, "TOPK failed", __FILE__, __LINE__);
    if (d_ro) hipFree(d_ro);
    if (d_co) hipFree(d_co);
    if (d_out3) hipFree(d_out3);
    std::printf("==> GPU Top K Degree Centrality Complete.\n");
}

}  // extern "C"
