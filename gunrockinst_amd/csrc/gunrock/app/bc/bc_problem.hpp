// app/bc/bc_problem.hpp -- device data for betweenness centrality (Brandes, one source per Enact).
//
// Same contract as the reference's BCProblem (gunrock/app/bc/bc_problem.cuh:36-485):
//   DataSlice { d_labels, d_preds, d_sigmas, d_deltas, d_bc_values, d_ebc_values, d_src_node }   (:52-61)
//   Init(stream_from_host, graph, num_gpus)   bc_values / ebc_values start at 0 and ACCUMULATE over sources   (:203-330)
//   Reset(src, frontier_type, queue_sizing)   labels = -1, sigmas = deltas = 0, label[src] = 0, sigma[src] = 1   (:341-460)
//   Extract(h_sigmas, h_bc_values, h_ebc_values)                                                              (:140-192)
// The reference never accumulates edge centralities (the atomicAdd on d_ebc_values is commented out,
// bc_functor.cuh:203) but allocates and extracts the zero array; so does this.  Frontier storage differs: every level's
// frontier is kept (the backward phase walks them deepest first), all in ONE queue of n entries -- a vertex is enqueued
// once -- where the reference keeps per-level offsets into a 2 x (n x sizing) double buffer.
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/app/problem_base.hpp>
#include <gunrock/util/memset_kernel.hpp>

namespace gunrock {
namespace app {
namespace bc {

template <typename Value>
__global__ void ScaleKernel(Value *d_out, Value factor, long long length)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < length; i += stride) d_out[i] *= factor;
}

// bc_values[v] += deltas[v] (the source's delta is never accumulated, so it adds 0)
template <typename Value>
__global__ void AccumulateKernel(Value *d_bc, const Value *d_deltas, long long length)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < length; i += stride) d_bc[i] += d_deltas[i];
}

template <typename _VertexId, typename _SizeT, typename _Value, bool _MARK_PREDECESSORS, bool _USE_DOUBLE_BUFFER>
struct BCProblem : ProblemBase<_VertexId, _SizeT, _Value, _USE_DOUBLE_BUFFER> {
    typedef ProblemBase<_VertexId, _SizeT, _Value, _USE_DOUBLE_BUFFER> Base;
    typedef _VertexId VertexId;
    typedef _SizeT SizeT;
    typedef _Value Value;
    static constexpr bool MARK_PREDECESSORS = _MARK_PREDECESSORS;
    static constexpr bool ENABLE_IDEMPOTENCE = false;

    struct DataSlice {
        VertexId *d_labels = nullptr;   // BFS depth from the source, -1 unreached
        Value *d_sigmas = nullptr;      // number of shortest paths from the source
        Value *d_deltas = nullptr;      // dependency of the source on the vertex
        // backward phase: (label, (1 + delta) / sigma) side by side, so that an edge into the level below costs ONE 8-byte gather
        // (the label test and the dependency term come with the same sector) instead of three 4-byte gathers from three arrays
        int2 *d_packed = nullptr;
        Value *d_bc_values = nullptr;   // accumulated over sources
        Value *d_ebc_values = nullptr;  // per edge, stays 0 (see header)
        VertexId src_node = -1;
        VertexId iteration = 0;         // level of the frontier being expanded
    };

    DataSlice **data_slices = nullptr;
    SizeT src_row[2] = {0, 0};

    ~BCProblem() override
    {
        if (data_slices) {
            DataSlice *ds = data_slices[0];
            if (ds) {
                if (ds->d_labels) hipFree(ds->d_labels);
                if (ds->d_sigmas) hipFree(ds->d_sigmas);
                if (ds->d_deltas) hipFree(ds->d_deltas);
                if (ds->d_packed) hipFree(ds->d_packed);
                if (ds->d_bc_values) hipFree(ds->d_bc_values);
                if (ds->d_ebc_values) hipFree(ds->d_ebc_values);
                delete ds;
            }
            delete[] data_slices;
        }
    }

    hipError_t AllocData()
    {
        hipError_t retval = hipSuccess;
        data_slices = new DataSlice *[1];
        data_slices[0] = new DataSlice();
        DataSlice *ds = data_slices[0];
        const size_t n = static_cast<size_t>(this->nodes > 0 ? this->nodes : 1);
        const size_t m = static_cast<size_t>(this->edges > 0 ? this->edges : 1);
        GR_CHECK(hipMalloc(&ds->d_labels, sizeof(VertexId) * n), "BCProblem hipMalloc d_labels failed");
        GR_CHECK(hipMalloc(&ds->d_sigmas, sizeof(Value) * n), "BCProblem hipMalloc d_sigmas failed");
        GR_CHECK(hipMalloc(&ds->d_deltas, sizeof(Value) * n), "BCProblem hipMalloc d_deltas failed");
        GR_CHECK(hipMalloc(&ds->d_packed, sizeof(int2) * n), "BCProblem hipMalloc d_packed failed");
        GR_CHECK(hipMalloc(&ds->d_bc_values, sizeof(Value) * n), "BCProblem hipMalloc d_bc_values failed");
        GR_CHECK(hipMalloc(&ds->d_ebc_values, sizeof(Value) * m), "BCProblem hipMalloc d_ebc_values failed");
        GR_CHECK(hipMemset(ds->d_bc_values, 0, sizeof(Value) * n), "BCProblem hipMemset failed");
        GR_CHECK(hipMemset(ds->d_ebc_values, 0, sizeof(Value) * m), "BCProblem hipMemset failed");
        GR_CHECK(hipDeviceSynchronize(), "BCProblem sync failed");  // (null-stream memsets: the problem's stream is not ordered behind them)
        return retval;
    }

    hipError_t Init(bool stream_from_host, Csr<VertexId, Value, SizeT> &graph, int num_gpus = 1)
    {
        hipError_t retval = hipSuccess;
        if ((retval = Base::Init(stream_from_host, graph, num_gpus, false))) return retval;
        return AllocData();
    }

    hipError_t InitFromDevice(SizeT nodes, SizeT edges, SizeT *d_row_offsets, VertexId *d_column_indices)
    {
        hipError_t retval = hipSuccess;
        if ((retval = Base::InitFromDevice(nodes, edges, d_row_offsets, d_column_indices))) return retval;
        return AllocData();
    }

    // bc_values = 0 (the reference's drivers do this between runs, tests/bc/test_bc.cu:433-434)
    hipError_t ClearBcValues()
    {
        return util::GRError(hipMemsetAsync(data_slices[0]->d_bc_values, 0, sizeof(Value) * static_cast<size_t>(this->nodes > 0 ? this->nodes : 1),
                                            this->graph_slices[0]->stream),
                             "BCProblem clear bc_values failed", __FILE__, __LINE__);
    }

    // bc_values *= factor (the reference halves them after the last source, bc_app.cu:112-113)
    hipError_t ScaleBcValues(Value factor)
    {
        if (this->nodes <= 0) return hipSuccess;
        hipLaunchKernelGGL((ScaleKernel<Value>), dim3(util::MemsetGrid(this->nodes)), dim3(256), 0, this->graph_slices[0]->stream,
                           data_slices[0]->d_bc_values, factor, static_cast<long long>(this->nodes));
        return util::GRError(hipGetLastError(), "BCProblem ScaleKernel launch failed", __FILE__, __LINE__);
    }

    hipError_t Reset(VertexId src, FrontierType frontier_type, double queue_sizing)
    {
        hipError_t retval = hipSuccess;
        // the queue must hold every level of one search: at least n entries
        if ((retval = Base::Reset(frontier_type, queue_sizing < 1.0 ? 1.0 : queue_sizing))) return retval;
        GraphSlice<VertexId, SizeT, Value> *gs = this->graph_slices[0];
        DataSlice *ds = data_slices[0];
        hipStream_t stream = gs->stream;
        util::Memset(ds->d_labels, static_cast<VertexId>(-1), this->nodes, stream);
        util::Memset(ds->d_sigmas, static_cast<Value>(0), this->nodes, stream);
        util::Memset(ds->d_deltas, static_cast<Value>(0), this->nodes, stream);
        ds->src_node = src;
        src_row[0] = src_row[1] = 0;
        if (src >= 0 && src < this->nodes) {
            const VertexId zero_label = 0;
            const Value one = static_cast<Value>(1);
            const SizeT zero_prefix = 0;
            GR_CHECK(hipMemcpyAsync(src_row, gs->d_row_offsets + src, 2 * sizeof(SizeT), hipMemcpyDeviceToHost, stream),
                     "BCProblem read source row failed");
            GR_CHECK(hipMemcpyAsync(ds->d_labels + src, &zero_label, sizeof(VertexId), hipMemcpyHostToDevice, stream), "BCProblem seed failed");
            GR_CHECK(hipMemcpyAsync(ds->d_sigmas + src, &one, sizeof(Value), hipMemcpyHostToDevice, stream), "BCProblem seed failed");
            GR_CHECK(hipMemcpyAsync(gs->frontier_queues[0].v, &src, sizeof(VertexId), hipMemcpyHostToDevice, stream), "BCProblem seed queue failed");
            GR_CHECK(hipMemcpyAsync(gs->frontier_queues[0].scan, &zero_prefix, sizeof(SizeT), hipMemcpyHostToDevice, stream),
                     "BCProblem seed queue failed");
            GR_CHECK(hipStreamSynchronize(stream), "BCProblem Reset sync failed");
            GR_CHECK(hipMemcpyAsync(gs->frontier_queues[0].row_start, &src_row[0], sizeof(SizeT), hipMemcpyHostToDevice, stream),
                     "BCProblem seed queue failed");
        }
        GR_CHECK(hipStreamSynchronize(stream), "BCProblem Reset sync failed");
        return retval;
    }

    SizeT SourceDegree() const { return src_row[1] - src_row[0]; }

    hipError_t Extract(Value *h_sigmas, Value *h_bc_values, Value *h_ebc_values)
    {
        hipError_t retval = hipSuccess;
        DataSlice *ds = data_slices[0];
        GR_CHECK(hipStreamSynchronize(this->graph_slices[0]->stream), "BCProblem Extract sync failed");
        if (this->nodes > 0 && h_sigmas)
            GR_CHECK(hipMemcpy(h_sigmas, ds->d_sigmas, sizeof(Value) * static_cast<size_t>(this->nodes), hipMemcpyDeviceToHost),
                     "BCProblem hipMemcpy d_sigmas failed");
        if (this->nodes > 0 && h_bc_values)
            GR_CHECK(hipMemcpy(h_bc_values, ds->d_bc_values, sizeof(Value) * static_cast<size_t>(this->nodes), hipMemcpyDeviceToHost),
                     "BCProblem hipMemcpy d_bc_values failed");
        if (this->edges > 0 && h_ebc_values)
            GR_CHECK(hipMemcpy(h_ebc_values, ds->d_ebc_values, sizeof(Value) * static_cast<size_t>(this->edges), hipMemcpyDeviceToHost),
                     "BCProblem hipMemcpy d_ebc_values failed");
        return retval;
    }
};

}  // namespace bc
}  // namespace app
}  // namespace gunrock
