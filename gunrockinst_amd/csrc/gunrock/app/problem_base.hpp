// app/problem_base.hpp -- device graph storage and frontier queues shared by all primitives.
//
// Same roles and names as the reference's ProblemBase / GraphSlice / FrontierType
// (gunrock/app/problem_base.cuh:35-39, 73-140, 226-431):
//   ProblemBase::Init  uploads the CSR once (problem_base.cuh:280-303),
//   ProblemBase::Reset sizes the two ping-pong frontier queues from queue_sizing
//                      (VERTEX_FRONTIERS -> nodes, EDGE_FRONTIERS -> edges, MIXED -> both; :378-382).
// MI355X-first differences: a frontier is (vertex, row start, degree prefix) (util/frontier.hpp); the
// graph may already be resident in HBM (InitFromDevice: 288 GB lets callers keep graphs on the card
// between primitives instead of re-uploading over PCIe per call); one HIP stream per problem.
// Multi-GPU: the reference has only a TODO (problem_base.cuh:336-338); vertex-cut partitioning lives
// in the package's multi_gpu layer and hands each rank's LOCAL CSR to this class unchanged.
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/csr.hpp>
#include <gunrock/util/error_utils.hpp>
#include <gunrock/util/frontier.hpp>

namespace gunrock {
namespace app {

enum FrontierType {
    VERTEX_FRONTIERS,  // O(n) ping-pong frontier queues
    EDGE_FRONTIERS,    // O(m) ping-pong frontier queues
    MIXED_FRONTIERS    // O(n) + O(m)
};

template <typename VertexId, typename SizeT, typename Value>
struct GraphSlice {
    int index = 0;
    SizeT nodes = 0;
    SizeT edges = 0;
    SizeT *d_row_offsets = nullptr;
    VertexId *d_column_indices = nullptr;
    Value *d_edge_values = nullptr;  // only when a primitive asks for weights
    bool owns_graph = false;

    util::Frontier<VertexId, SizeT> frontier_queues[2];
    SizeT frontier_elements[2] = {0, 0};
    hipStream_t stream = 0;

    ~GraphSlice() { Release(); }

    void ReleaseQueues()
    {
        for (int i = 0; i < 2; ++i) {
            if (frontier_queues[i].v) util::GRError(hipFree(frontier_queues[i].v), "GraphSlice hipFree queue failed", __FILE__, __LINE__);
            if (frontier_queues[i].row_start) util::GRError(hipFree(frontier_queues[i].row_start), "GraphSlice hipFree queue failed", __FILE__, __LINE__);
            if (frontier_queues[i].scan) util::GRError(hipFree(frontier_queues[i].scan), "GraphSlice hipFree queue failed", __FILE__, __LINE__);
            frontier_queues[i] = util::Frontier<VertexId, SizeT>();
            frontier_elements[i] = 0;
        }
    }

    void Release()
    {
        ReleaseQueues();
        if (owns_graph) {
            if (d_row_offsets) util::GRError(hipFree(d_row_offsets), "GraphSlice hipFree d_row_offsets failed", __FILE__, __LINE__);
            if (d_column_indices) util::GRError(hipFree(d_column_indices), "GraphSlice hipFree d_column_indices failed", __FILE__, __LINE__);
            if (d_edge_values) util::GRError(hipFree(d_edge_values), "GraphSlice hipFree d_edge_values failed", __FILE__, __LINE__);
        }
        d_row_offsets = nullptr;
        d_column_indices = nullptr;
        d_edge_values = nullptr;
        if (stream) {
            util::GRError(hipStreamDestroy(stream), "GraphSlice hipStreamDestroy failed", __FILE__, __LINE__);
            stream = 0;
        }
    }
};

template <typename _VertexId, typename _SizeT, typename _Value, bool _USE_DOUBLE_BUFFER = false>
struct ProblemBase {
    typedef _VertexId VertexId;
    typedef _SizeT SizeT;
    typedef _Value Value;
    static constexpr bool USE_DOUBLE_BUFFER = _USE_DOUBLE_BUFFER;

    int num_gpus = 1;
    SizeT nodes = 0;
    SizeT edges = 0;
    GraphSlice<VertexId, SizeT, Value> **graph_slices = nullptr;

    virtual ~ProblemBase()
    {
        if (graph_slices) {
            for (int i = 0; i < num_gpus; ++i) delete graph_slices[i];
            delete[] graph_slices;
        }
    }

    hipError_t MakeSlice()
    {
        hipError_t retval = hipSuccess;
        if (graph_slices) {
            for (int i = 0; i < num_gpus; ++i) delete graph_slices[i];
            delete[] graph_slices;
        }
        num_gpus = 1;  // one process per GPU; N-GPU runs partition above this class
        graph_slices = new GraphSlice<VertexId, SizeT, Value> *[1];
        graph_slices[0] = new GraphSlice<VertexId, SizeT, Value>();
        graph_slices[0]->nodes = nodes;
        graph_slices[0]->edges = edges;
        GR_CHECK(hipStreamCreateWithFlags(&graph_slices[0]->stream, hipStreamNonBlocking),
                 "ProblemBase hipStreamCreate failed");
        return retval;
    }

    // Host CSR -> HBM (problem_base.cuh:226-342).  `load_edge_values` also uploads graph.edge_values.
    hipError_t Init(bool /*stream_from_host: mapped-host streaming is not offered, HBM holds the graph*/,
                    const Csr<VertexId, Value, SizeT> &graph, int /*_num_gpus*/ = 1, bool load_edge_values = false)
    {
        hipError_t retval = hipSuccess;
        nodes = graph.nodes;
        edges = graph.edges;
        if ((retval = MakeSlice())) return retval;
        GraphSlice<VertexId, SizeT, Value> *gs = graph_slices[0];
        gs->owns_graph = true;
        const size_t ro_bytes = sizeof(SizeT) * (static_cast<size_t>(nodes) + 1);
        const size_t ci_bytes = sizeof(VertexId) * static_cast<size_t>(edges > 0 ? edges : 1);
        GR_CHECK(hipMalloc(&gs->d_row_offsets, ro_bytes), "ProblemBase hipMalloc d_row_offsets failed");
        GR_CHECK(hipMalloc(&gs->d_column_indices, ci_bytes), "ProblemBase hipMalloc d_column_indices failed");
        GR_CHECK(hipMemcpy(gs->d_row_offsets, graph.row_offsets, ro_bytes, hipMemcpyHostToDevice),
                 "ProblemBase hipMemcpy d_row_offsets failed");
        if (edges > 0)
            GR_CHECK(hipMemcpy(gs->d_column_indices, graph.column_indices, sizeof(VertexId) * static_cast<size_t>(edges),
                               hipMemcpyHostToDevice),
                     "ProblemBase hipMemcpy d_column_indices failed");
        if (load_edge_values && graph.edge_values) {
            GR_CHECK(hipMalloc(&gs->d_edge_values, sizeof(Value) * static_cast<size_t>(edges > 0 ? edges : 1)),
                     "ProblemBase hipMalloc d_edge_values failed");
            if (edges > 0)
                GR_CHECK(hipMemcpy(gs->d_edge_values, graph.edge_values, sizeof(Value) * static_cast<size_t>(edges),
                                   hipMemcpyHostToDevice),
                         "ProblemBase hipMemcpy d_edge_values failed");
        }
        return retval;
    }

    // CSR already resident in HBM (borrowed, not freed).
    hipError_t InitFromDevice(SizeT nodes_, SizeT edges_, SizeT *d_row_offsets, VertexId *d_column_indices,
                              Value *d_edge_values = nullptr)
    {
        hipError_t retval = hipSuccess;
        nodes = nodes_;
        edges = edges_;
        if ((retval = MakeSlice())) return retval;
        graph_slices[0]->owns_graph = false;
        graph_slices[0]->d_row_offsets = d_row_offsets;
        graph_slices[0]->d_column_indices = d_column_indices;
        graph_slices[0]->d_edge_values = d_edge_values;
        return retval;
    }

    // (Re)size the ping-pong queues; memory is reused when large enough (problem_base.cuh:352-431).
    hipError_t Reset(FrontierType frontier_type, double queue_sizing)
    {
        hipError_t retval = hipSuccess;
        GraphSlice<VertexId, SizeT, Value> *gs = graph_slices[0];
        double base = 0;
        switch (frontier_type) {
            case VERTEX_FRONTIERS: base = nodes; break;
            case EDGE_FRONTIERS: base = edges > nodes ? edges : nodes; break;
            case MIXED_FRONTIERS: base = static_cast<double>(nodes) + edges; break;
        }
        double want = base * queue_sizing;
        if (want < 1024) want = 1024;
        if (want > 2147483000.0) want = 2147483000.0;
        const SizeT elements = static_cast<SizeT>(want);
        for (int i = 0; i < 2; ++i) {
            if (gs->frontier_elements[i] >= elements) continue;
            if (gs->frontier_queues[i].v) {
                util::GRError(hipFree(gs->frontier_queues[i].v), "GraphSlice hipFree queue failed", __FILE__, __LINE__);
                util::GRError(hipFree(gs->frontier_queues[i].row_start), "GraphSlice hipFree queue failed", __FILE__, __LINE__);
                util::GRError(hipFree(gs->frontier_queues[i].scan), "GraphSlice hipFree queue failed", __FILE__, __LINE__);
            }
            GR_CHECK(hipMalloc(&gs->frontier_queues[i].v, sizeof(VertexId) * static_cast<size_t>(elements)),
                     "ProblemBase hipMalloc frontier queue failed");
            GR_CHECK(hipMalloc(&gs->frontier_queues[i].row_start, sizeof(SizeT) * static_cast<size_t>(elements)),
                     "ProblemBase hipMalloc frontier queue failed");
            GR_CHECK(hipMalloc(&gs->frontier_queues[i].scan, sizeof(SizeT) * static_cast<size_t>(elements)),
                     "ProblemBase hipMalloc frontier queue failed");
            gs->frontier_queues[i].capacity = elements;
            gs->frontier_elements[i] = elements;
        }
        return retval;
    }
};

}  // namespace app
}  // namespace gunrock
