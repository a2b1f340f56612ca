// app/pr/pr_problem.hpp -- device data for PageRank.
//
// Contract of the reference's PRProblem (gunrock/app/pr/pr_problem.cuh:36-467):
//   DataSlice { d_rank_curr, d_rank_next, d_degrees, d_degrees_pong, d_node_ids, delta, threshold, src_node }   (:57-68)
//   Init(stream_from_host, graph, num_gpus)                                                                  (:186-307)
//   Reset(src, delta, threshold, frontier_type): rank_curr = 1 - delta, degrees = out-degrees, node_ids = iota,
//        queue = every vertex                                                                                (:316-457)
//   Extract(h_rank, h_node_id): ranks sorted descending with their vertex ids                                (:139-175)
// Differences: delta / threshold / src_node travel by value inside the DataSlice kernel argument (the reference keeps
// one-element device arrays for them); contrib[v] = rank[v] / degree[v] is kept next to the ranks because ranks are pulled
// over the IN-neighbour lists (pr_functor.hpp), which needs the inverse graph: the caller's CSC, the CSR itself when the
// graph is symmetric, or a transpose built on the device.
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/app/problem_base.hpp>
#include <gunrock/graphio/device_sort.hpp>
#include <gunrock/util/memset_kernel.hpp>

namespace gunrock {
namespace app {
namespace pr {

template <typename SizeT>
__global__ void OutDegreeKernel(const SizeT *d_row_offsets, long long nodes, SizeT *d_degrees, SizeT *d_degrees_pong)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long v = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; v < nodes; v += stride) {
        const SizeT d = d_row_offsets[v + 1] - d_row_offsets[v];
        d_degrees[v] = d;
        d_degrees_pong[v] = d;
    }
}

template <typename _VertexId, typename _SizeT, typename _Value>
struct PRProblem : ProblemBase<_VertexId, _SizeT, _Value, false> {
    typedef ProblemBase<_VertexId, _SizeT, _Value, false> Base;
    typedef _VertexId VertexId;
    typedef _SizeT SizeT;
    typedef _Value Value;
    static constexpr bool MARK_PREDECESSORS = false;
    static constexpr bool ENABLE_IDEMPOTENCE = false;

    struct DataSlice {
        Value *d_rank_curr = nullptr;     // rank per vertex
        Value *d_rank_next = nullptr;     // sum of the in-neighbours' contributions (the reducing advance writes it)
        Value *d_contrib = nullptr;       // rank_curr / degree of the vertices that still have out-edges
        SizeT *d_degrees = nullptr;       // out-degree after peeling (-1: peeled off)
        SizeT *d_degrees_pong = nullptr;
        int *d_zero_flag = nullptr;       // peeling round: 1 = lost its last out-edge in the previous round
        int *d_zero_count = nullptr;      // peeling round: per queue position, out-neighbours with the flag set
        VertexId *d_node_ids = nullptr;   // vertex ids by descending rank (after Enact)
        Value *d_rank_sorted = nullptr;   // their ranks
        Value delta = 0;
        Value threshold = 0;
        VertexId src_node = -1;
    };

    DataSlice **data_slices = nullptr;
    DataSlice **d_data_slices = nullptr;  // kept for source compatibility; unused
    const SizeT *d_inv_row_offsets = nullptr;  // in-neighbour lists
    const VertexId *d_inv_column_indices = nullptr;
    SizeT *d_own_inv_row_offsets = nullptr;    // set when the transpose was built here
    VertexId *d_own_inv_column_indices = nullptr;
    util::Frontier<VertexId, SizeT> inv_frontier;  // surviving vertices with the degree prefix of their in-lists
    graphio::DeviceKeySort sorter;

    ~PRProblem() override
    {
        if (data_slices) {
            DataSlice *ds = data_slices[0];
            if (ds) {
                void *ptrs[] = {ds->d_rank_curr, ds->d_rank_next, ds->d_contrib, ds->d_degrees, ds->d_degrees_pong, ds->d_zero_flag,
                                ds->d_zero_count, ds->d_node_ids, ds->d_rank_sorted};
                for (void *p : ptrs)
                    if (p) util::GRError(hipFree(p), "PRProblem hipFree failed", __FILE__, __LINE__);
                delete ds;
            }
            delete[] data_slices;
        }
        if (d_own_inv_row_offsets) util::GRError(hipFree(d_own_inv_row_offsets), "PRProblem hipFree failed", __FILE__, __LINE__);
        if (d_own_inv_column_indices) util::GRError(hipFree(d_own_inv_column_indices), "PRProblem hipFree failed", __FILE__, __LINE__);
        void *q[] = {inv_frontier.v, inv_frontier.row_start, inv_frontier.scan};
        for (void *p : q)
            if (p) util::GRError(hipFree(p), "PRProblem hipFree failed", __FILE__, __LINE__);
    }

    hipError_t AllocData()
    {
        hipError_t retval = hipSuccess;
        data_slices = new DataSlice *[1];
        data_slices[0] = new DataSlice();
        DataSlice *ds = data_slices[0];
        const size_t n = static_cast<size_t>(this->nodes > 0 ? this->nodes : 1);
        GR_CHECK(hipMalloc(&ds->d_rank_curr, sizeof(Value) * n), "PRProblem hipMalloc failed");
        GR_CHECK(hipMalloc(&ds->d_rank_next, sizeof(Value) * n), "PRProblem hipMalloc failed");
        GR_CHECK(hipMalloc(&ds->d_contrib, sizeof(Value) * n), "PRProblem hipMalloc failed");
        GR_CHECK(hipMalloc(&ds->d_rank_sorted, sizeof(Value) * n), "PRProblem hipMalloc failed");
        GR_CHECK(hipMalloc(&ds->d_degrees, sizeof(SizeT) * n), "PRProblem hipMalloc failed");
        GR_CHECK(hipMalloc(&ds->d_degrees_pong, sizeof(SizeT) * n), "PRProblem hipMalloc failed");
        GR_CHECK(hipMalloc(&ds->d_zero_flag, sizeof(int) * n), "PRProblem hipMalloc failed");
        GR_CHECK(hipMalloc(&ds->d_zero_count, sizeof(int) * n), "PRProblem hipMalloc failed");
        GR_CHECK(hipMalloc(&ds->d_node_ids, sizeof(VertexId) * n), "PRProblem hipMalloc failed");
        GR_CHECK(hipMalloc(&inv_frontier.v, sizeof(VertexId) * (n + 1)), "PRProblem hipMalloc failed");
        GR_CHECK(hipMalloc(&inv_frontier.row_start, sizeof(SizeT) * (n + 1)), "PRProblem hipMalloc failed");
        GR_CHECK(hipMalloc(&inv_frontier.scan, sizeof(SizeT) * (n + 1)), "PRProblem hipMalloc failed");
        inv_frontier.capacity = static_cast<SizeT>(n + 1);
        return retval;
    }

    hipError_t Init(bool stream_from_host, const Csr<VertexId, Value, SizeT> &graph, int num_gpus = 1)
    {
        hipError_t retval = hipSuccess;
        if ((retval = Base::Init(stream_from_host, graph, num_gpus))) return retval;
        return AllocData();
    }
    hipError_t InitFromDevice(SizeT nodes, SizeT edges, SizeT *d_row_offsets, VertexId *d_column_indices)
    {
        hipError_t retval = hipSuccess;
        if ((retval = Base::InitFromDevice(nodes, edges, d_row_offsets, d_column_indices))) return retval;
        return AllocData();
    }

    // In-neighbour lists: borrowed CSC arrays in HBM ...
    void SetInverseGraph(const SizeT *d_iro, const VertexId *d_ici)
    {
        d_inv_row_offsets = d_iro;
        d_inv_column_indices = d_ici;
    }
    // ... the graph's own CSR (symmetric input) ...
    void InverseIsSelf() { SetInverseGraph(this->graph_slices[0]->d_row_offsets, this->graph_slices[0]->d_column_indices); }
    // ... or the transpose, built on the device
    hipError_t BuildInverse()
    {
        hipError_t retval = hipSuccess;
        GraphSlice<VertexId, SizeT, Value> *gs = this->graph_slices[0];
        if (!d_own_inv_row_offsets) {
            GR_CHECK(hipMalloc(&d_own_inv_row_offsets, sizeof(SizeT) * (static_cast<size_t>(this->nodes) + 2)), "PRProblem hipMalloc failed");
            GR_CHECK(hipMalloc(&d_own_inv_column_indices, sizeof(VertexId) * static_cast<size_t>(this->edges > 0 ? this->edges : 1)),
                     "PRProblem hipMalloc failed");
        }
        GR_CHECK(graphio::DeviceTransposeCsr(this->nodes, this->edges, gs->d_row_offsets, gs->d_column_indices, d_own_inv_row_offsets,
                                             d_own_inv_column_indices, gs->stream),
                 "PRProblem transpose failed");
        SetInverseGraph(d_own_inv_row_offsets, d_own_inv_column_indices);
        return retval;
    }

    hipError_t Reset(VertexId src, Value delta, Value threshold, FrontierType frontier_type)
    {
        hipError_t retval = hipSuccess;
        if ((retval = Base::Reset(frontier_type, 1.0))) return retval;  // "Default queue sizing is 1.0" (pr_problem.cuh:326)
        GraphSlice<VertexId, SizeT, Value> *gs = this->graph_slices[0];
        DataSlice *ds = data_slices[0];
        hipStream_t stream = gs->stream;
        ds->delta = delta;
        ds->threshold = threshold;
        ds->src_node = src;
        if (this->nodes > 0) {
            util::Memset(ds->d_rank_curr, static_cast<Value>(1.0 - delta), this->nodes, stream);  // pr_problem.cuh:423
            util::Memset(ds->d_rank_next, static_cast<Value>(0), this->nodes, stream);
            util::Memset(ds->d_contrib, static_cast<Value>(0), this->nodes, stream);
            util::MemsetIdx(ds->d_node_ids, this->nodes, stream);
            hipLaunchKernelGGL((OutDegreeKernel<SizeT>), dim3(1024), dim3(256), 0, stream, gs->d_row_offsets, static_cast<long long>(this->nodes),
                               ds->d_degrees, ds->d_degrees_pong);
            GR_CHECK(hipGetLastError(), "OutDegreeKernel launch failed");
        }
        return retval;
    }

    // ranks in descending order with their vertex ids, as the reference returns them; `count` <= nodes entries
    hipError_t Extract(Value *h_rank, VertexId *h_node_id, SizeT count = -1)
    {
        hipError_t retval = hipSuccess;
        DataSlice *ds = data_slices[0];
        GR_CHECK(hipStreamSynchronize(this->graph_slices[0]->stream), "PRProblem Extract sync failed");
        if (count < 0 || count > this->nodes) count = this->nodes;
        if (count > 0 && h_rank)
            GR_CHECK(hipMemcpy(h_rank, ds->d_rank_sorted, sizeof(Value) * static_cast<size_t>(count), hipMemcpyDeviceToHost), "PRProblem hipMemcpy failed");
        if (count > 0 && h_node_id)
            GR_CHECK(hipMemcpy(h_node_id, ds->d_node_ids, sizeof(VertexId) * static_cast<size_t>(count), hipMemcpyDeviceToHost), "PRProblem hipMemcpy failed");
        return retval;
    }
};

}  // namespace pr
}  // namespace app
}  // namespace gunrock
