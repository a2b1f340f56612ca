// lib/bc_app.hip -- betweenness-centrality entry points of libgunrock.so.
//  * gunrock_bc_func: the C entry point of the reference (gunrock/gunrock.h:113-117, gunrock/app/bc/bc_app.cu:40-265):
//    only <VTXID_INT, SIZET_INT, VALUE_FLOAT> is supported (:163-232); the source comes from src_mode
//    (manually -> src_node, where -1 means "every vertex in turn", :90-100); bc_values are halved after the last
//    source (:112-113); node_values = float bc_values[n], edge_values = float ebc_values[m] (all zero: the reference never
//    accumulates them, bc_functor.cuh:203), both malloc()ed here and owned by the caller.
//  * grx_bc_*: Problem / Enactor phases as separate C calls.
#include <gunrock/gunrock.h>
#include <gunrock/gunrock_mi355x.h>

#include <cstdio>
#include <cstdlib>

#include <gunrock/app/bc/bc_enactor.hpp>
#include <gunrock/app/bc/bc_problem.hpp>
#include <gunrock/csr.hpp>
#include <gunrock/graphio/utils.hpp>
#include <gunrock/util/context.hpp>

using namespace gunrock;
using namespace gunrock::app;
using namespace gunrock::app::bc;

namespace {
typedef BCProblem<int, int, float, true, false> Problem;  // the reference's instantiation (bc_app.cu:61-66)
}

struct grx_bc {
    util::DeviceContext context;
    Problem problem;
    BCEnactor<false> enactor;
    hipEvent_t start = nullptr, stop = nullptr;
    explicit grx_bc(int device) : context(device), enactor(false)
    {
        util::GRError(hipEventCreate(&start), "hipEventCreate failed", __FILE__, __LINE__);
        util::GRError(hipEventCreate(&stop), "hipEventCreate failed", __FILE__, __LINE__);
    }
    ~grx_bc()
    {
        if (start) hipEventDestroy(start);
        if (stop) hipEventDestroy(stop);
    }
    // one or all sources, then the reference's 0.5 scaling; elapsed = device time of the whole loop
    hipError_t Run(int src, int max_grid_size, double queue_sizing, float *ms)
    {
        hipError_t retval = hipSuccess;
        hipStream_t stream = problem.graph_slices[0]->stream;
        GR_CHECK(problem.ClearBcValues(), "BC clear failed");
        GR_CHECK(hipEventRecord(start, stream), "hipEventRecord failed");
        const int first = (src == -1) ? 0 : src;
        const int last = (src == -1) ? problem.nodes : src + 1;
        for (int s = first; s < last; ++s) {
            GR_CHECK(problem.Reset(s, enactor.GetFrontierType(), queue_sizing), "BC Problem Data Reset Failed");
            GR_CHECK(enactor.Enact<Problem>(context, &problem, s, max_grid_size), "BC Problem Enact Failed");
        }
        GR_CHECK(problem.ScaleBcValues(0.5f), "BC scale failed");
        GR_CHECK(hipEventRecord(stop, stream), "hipEventRecord failed");
        GR_CHECK(hipEventSynchronize(stop), "hipEventSynchronize failed");
        float t = 0;
        GR_CHECK(hipEventElapsedTime(&t, start, stop), "hipEventElapsedTime failed");
        if (ms) *ms = t;
        return retval;
    }
};

extern "C" {

int grx_bc_create(grx_bc **out, int device)
{
    if (!out) return -1;
    *out = new grx_bc(device);
    return 0;
}

int grx_bc_init(grx_bc *p, int nodes, int edges, const int *row_offsets, const int *col_indices)
{
    if (!p || !row_offsets || nodes < 0 || edges < 0) return -1;
    Csr<int, float, int> wrap(false);
    wrap.nodes = nodes;
    wrap.edges = edges;
    wrap.row_offsets = const_cast<int *>(row_offsets);
    wrap.column_indices = const_cast<int *>(col_indices);
    hipError_t rc = p->problem.Init(false, wrap, 1);
    wrap.row_offsets = nullptr;
    wrap.column_indices = nullptr;
    return static_cast<int>(rc);
}

int grx_bc_init_device(grx_bc *p, int nodes, int edges, int *d_row_offsets, int *d_col_indices)
{
    if (!p || !d_row_offsets || nodes < 0 || edges < 0) return -1;
    return static_cast<int>(p->problem.InitFromDevice(nodes, edges, d_row_offsets, d_col_indices));
}

int grx_bc_run(grx_bc *p, int src, int max_grid_size, double queue_sizing, float *elapsed_ms)
{
    if (!p || !p->problem.data_slices || src < -1 || src >= p->problem.nodes) return -1;
    return static_cast<int>(p->Run(src, max_grid_size, queue_sizing, elapsed_ms));
}

int grx_bc_extract(grx_bc *p, float *h_sigmas, float *h_bc_values, float *h_ebc_values)
{
    if (!p || !p->problem.data_slices) return -1;
    return static_cast<int>(p->problem.Extract(h_sigmas, h_bc_values, h_ebc_values));
}

void grx_bc_destroy(grx_bc *p) { delete p; }

void gunrock_bc_func(struct GunrockGraph *graph_out, const struct GunrockGraph *graph_in, struct GunrockConfig config,
                     struct GunrockDataType data_type)
{
    if (!graph_out || !graph_in) {
        std::fprintf(stderr, "[gunrock] gunrock_bc_func: null graph\n");
        return;
    }
    if (data_type.VTXID_TYPE != VTXID_INT || data_type.SIZET_TYPE != SIZET_INT || data_type.VALUE_TYPE != VALUE_FLOAT) {
        std::printf("Not Yet Support This DataType Combination.\n");  // bc_app.cu:170-179, 226-232
        return;
    }
    const int nodes = static_cast<int>(graph_in->num_nodes);
    const int edges = static_cast<int>(graph_in->num_edges);
    Csr<int, float, int> view(false);
    view.nodes = nodes;
    view.edges = edges;
    view.row_offsets = static_cast<int *>(graph_in->row_offsets);
    view.column_indices = static_cast<int *>(graph_in->col_indices);
    int src = -1;
    switch (config.src_mode) {  // bc_app.cu:196-219
        case randomize: src = graphio::RandomNode(nodes); break;
        case largest_degree: {
            int max_deg = 0;
            src = view.GetNodeWithHighestDegree(max_deg);
            break;
        }
        case manually: src = config.src_node; break;
        default: src = -1; break;
    }
    view.row_offsets = nullptr;
    view.column_indices = nullptr;
    if (src < -1 || src >= nodes) {
        std::fprintf(stderr, "[gunrock] gunrock_bc_func: source %d outside the graph\n", src);
        return;
    }
    grx_bc *h = nullptr;
    float ms = 0.0f;
    float *bc_values = static_cast<float *>(std::malloc(sizeof(float) * static_cast<size_t>(nodes > 0 ? nodes : 1)));
    float *ebc_values = static_cast<float *>(std::malloc(sizeof(float) * static_cast<size_t>(edges > 0 ? edges : 1)));
    int rc = grx_bc_create(&h, config.device);
    if (rc == 0) rc = grx_bc_init(h, nodes, edges, static_cast<const int *>(graph_in->row_offsets), static_cast<const int *>(graph_in->col_indices));
    if (rc == 0) rc = grx_bc_run(h, src, 0, config.queue_size > 0 ? config.queue_size : 1.0, &ms);
    if (rc == 0) rc = grx_bc_extract(h, nullptr, bc_values, ebc_values);
    grx_bc_destroy(h);
    if (rc != 0) {
        std::fprintf(stderr, "[gunrock] gunrock_bc_func failed (%d)\n", rc);
        std::free(bc_values);
        std::free(ebc_values);
        return;
    }
    graph_out->node_values = bc_values;
    graph_out->edge_values = ebc_values;
    std::printf("GPU Betweeness Centrality finished in %lf msec.\n", static_cast<double>(ms));  // bc_app.cu:128
}

}  // extern "C"
