// app/bc/bc_enactor.hpp -- host loop for betweenness centrality from one source (Brandes).
//
// Public contract of the reference's BCEnactor (gunrock/app/bc/bc_enactor.cuh:36-634):
//   template <bool INSTRUMENT> class BCEnactor : EnactorBase
//   Enact<BCProblem>(context, problem, src, max_grid_size = 0)     (:573-630)
//   GetStatistics(avg_duty)                                          (:150-165)
// Loop shape kept from EnactBC (:188-560): a forward phase that is a BFS with shortest-path counting, every level's
// frontier kept; then a backward phase over the kept frontiers, deepest first, accumulating dependencies.  The
// reference runs advance + filter per forward level and two launches per backward level with blocking queue-length
// reads in both phases; here a forward level is ONE advance launch + one read-back (the advance output is already the
// compacted next frontier), and the backward phase is one advance launch per level with no read-back at all (it has
// no output queue), one synchronisation at the end.
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

#include <gunrock/app/bc/bc_functor.hpp>
#include <gunrock/app/bc/bc_problem.hpp>
#include <gunrock/app/enactor_base.hpp>
#include <gunrock/oprtr/advance/kernel.hpp>
#include <gunrock/util/context.hpp>

namespace gunrock {
namespace app {
namespace bc {

// d_packed[v] = (label, 1 / sigma): the term a vertex offers before anything was accumulated into its delta (deltas are zero)
template <typename VertexId, typename Value>
__global__ void PackTermsKernel(const VertexId *d_labels, const Value *d_sigmas, int2 *d_packed, long long nodes)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long v = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; v < nodes; v += stride) {
        const VertexId l = d_labels[v];
        d_packed[v] = make_int2(l, l >= 0 ? __float_as_int(static_cast<Value>(1) / d_sigmas[v]) : 0);
    }
}
// ... and (1 + delta) / sigma once the deltas of a level's vertices are final
template <typename VertexId, typename Value>
__global__ void RefreshTermsKernel(const VertexId *d_queue, long long length, const Value *d_sigmas, const Value *d_deltas, int2 *d_packed)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < length; i += stride) {
        const VertexId v = d_queue[i];
        d_packed[v].y = __float_as_int((static_cast<Value>(1) + d_deltas[v]) / d_sigmas[v]);
    }
}

template <bool INSTRUMENT>
class BCEnactor : public EnactorBase {
   public:
    explicit BCEnactor(bool DEBUG = false) : EnactorBase(VERTEX_FRONTIERS, DEBUG) {}
    ~BCEnactor() override {}

    void GetStatistics(double &avg_duty) { avg_duty = 0.0; }
    void GetStatistics(long long &total_queued, long long &search_depth, double &avg_duty)
    {
        total_queued = enactor_stats.total_queued;
        search_depth = enactor_stats.iteration;
        avg_duty = 0.0;
    }

    typedef oprtr::advance::KernelPolicy<256, 4, 8, oprtr::advance::LB> AdvancePolicy;

    template <typename BCProblem>
    hipError_t Enact(util::DeviceContext & /*context*/, BCProblem *problem, typename BCProblem::VertexId src, int max_grid_size = 0)
    {
        typedef typename BCProblem::VertexId VertexId;
        typedef typename BCProblem::SizeT SizeT;
        typedef typename BCProblem::Value Value;
        typedef ForwardFunctor<VertexId, SizeT, Value, BCProblem> Forward;
        typedef BackwardReduceFunctor<VertexId, SizeT, Value, BCProblem> Backward;

        hipError_t retval = hipSuccess;
        if ((retval = EnactorBase::Setup(max_grid_size, AdvancePolicy::MIN_BLOCKS, 8))) return retval;
        GraphSlice<VertexId, SizeT, Value> *gs = problem->graph_slices[0];
        typename BCProblem::DataSlice *ds = problem->data_slices[0];
        hipStream_t stream = gs->stream;
        if (src < 0 || src >= problem->nodes) return retval;
        if ((retval = work_progress.Reset(stream))) return retval;

        // level L's frontier lives at [level_offset[L], level_offset[L] + level_len[L]) of queue 0
        const util::Frontier<VertexId, SizeT> &q = gs->frontier_queues[0];
        auto view = [&](SizeT offset) {
            util::Frontier<VertexId, SizeT> f;
            f.v = q.v + offset;
            f.row_start = q.row_start + offset;
            f.scan = q.scan + offset;
            f.capacity = q.capacity - offset;
            return f;
        };
        level_offset.clear();
        level_len.clear();
        level_edges.clear();

        unsigned queue_length = problem->SourceDegree() > 0 ? 1u : 0u;
        unsigned queue_edges = static_cast<unsigned>(problem->SourceDegree());
        if ((retval = work_progress.SetTail(0, queue_length, queue_edges, stream))) return retval;
        SizeT offset = 0;
        long long iteration = 0;

        // ---- forward: BFS + sigma ----
        while (queue_length > 0) {
            level_offset.push_back(offset);
            level_len.push_back(static_cast<SizeT>(queue_length));
            level_edges.push_back(static_cast<SizeT>(queue_edges));
            enactor_stats.total_queued += queue_length;
            ds->iteration = static_cast<VertexId>(iteration);
            oprtr::advance::AdvanceArgs<VertexId, SizeT> args;
            args.in = view(offset);
            args.out = view(offset + static_cast<SizeT>(queue_length));
            args.in_len = static_cast<SizeT>(queue_length);
            args.in_edges = static_cast<SizeT>(queue_edges);
            args.d_row_offsets = gs->d_row_offsets;
            args.d_column_indices = gs->d_column_indices;
            args.d_tail_out = work_progress.d_tail + ((iteration + 1) & 3);
            args.d_tail_clear = work_progress.d_tail + ((iteration + 2) & 3);
            args.d_overflow = work_progress.d_overflow;
            if ((retval = oprtr::advance::LaunchKernel<AdvancePolicy, BCProblem, Forward>(args, *ds, max_grid_size, stream,
                                                                                            oprtr::advance::V2V)))
                return retval;
            offset += static_cast<SizeT>(queue_length);
            ++iteration;
            if ((retval = work_progress.GetTail(static_cast<int>(iteration & 3), queue_length, queue_edges, stream))) return retval;
            if (work_progress.OverflowAtLastSync())
                return util::GRError(hipErrorInvalidConfiguration, "Frontier queue overflow. Please increase queue-sizing factor.",
                                     __FILE__, __LINE__);
        }
        enactor_stats.iteration = iteration;

        // (label, 1 / sigma) side by side for the backward gathers (bc_functor.hpp BackwardReduceFunctor)
        hipLaunchKernelGGL((PackTermsKernel<VertexId, Value>), dim3(util::MemsetGrid(problem->nodes)), dim3(256), 0, stream, ds->d_labels,
                           ds->d_sigmas, ds->d_packed, static_cast<long long>(problem->nodes));
        if ((retval = util::GRError("PackTermsKernel launch failed", __FILE__, __LINE__))) return retval;

        // ---- backward: dependencies, deepest recorded frontier first.  The recorded levels are the ones that have out-edges: the
        // last one can still have children -- vertices WITHOUT out-edges (sinks of a directed graph), which the frontier writer
        // does not enqueue -- so it takes part.  (Level 0 is the source alone, which accumulates nothing: bc_functor.cuh:205-208.)
        for (long long level = static_cast<long long>(level_len.size()) - 1; level >= 1; --level) {
            if (level_edges[level] <= 0) continue;
            ds->iteration = static_cast<VertexId>(level);
            oprtr::advance::AdvanceArgs<VertexId, SizeT> args;
            args.in = view(level_offset[level]);
            args.out = util::Frontier<VertexId, SizeT>();
            args.in_len = level_len[level];
            args.in_edges = level_edges[level];
            args.d_row_offsets = gs->d_row_offsets;
            args.d_column_indices = gs->d_column_indices;
            args.d_tail_out = nullptr;  // count-only launch, count not wanted
            args.d_tail_clear = nullptr;
            args.d_overflow = work_progress.d_overflow;
            // delta[s] of this level's vertices <- sum over their edges into the level below (deltas are zero since Reset and
            // every vertex is reduced into exactly once, so no identity fill)
            if ((retval = oprtr::advance::LaunchReduce<AdvancePolicy, BCProblem, Backward, oprtr::advance::VERTEX, oprtr::advance::PLUS, Value,
                                                       true>(args, *ds, static_cast<const Value *>(nullptr), ds->d_deltas, max_grid_size,
                                                             stream, static_cast<long long>(problem->nodes), false)))
                return retval;
            // this level's deltas are final: the term its vertices offer to the level above
            if (level > 1) {
                hipLaunchKernelGGL((RefreshTermsKernel<VertexId, Value>), dim3(util::MemsetGrid(level_len[level])), dim3(256), 0, stream,
                                   q.v + level_offset[level], static_cast<long long>(level_len[level]), ds->d_sigmas, ds->d_deltas, ds->d_packed);
                if ((retval = util::GRError("RefreshTermsKernel launch failed", __FILE__, __LINE__))) return retval;
            }
        }
        // dependencies are final: fold them into the running centralities, once per vertex
        hipLaunchKernelGGL((AccumulateKernel<Value>), dim3(util::MemsetGrid(problem->nodes)), dim3(256), 0, stream, ds->d_bc_values,
                           ds->d_deltas, static_cast<long long>(problem->nodes));
        if ((retval = util::GRError("AccumulateKernel launch failed", __FILE__, __LINE__))) return retval;
        return util::GRError(hipStreamSynchronize(stream), "BCEnactor sync failed", __FILE__, __LINE__);
    }

   private:
    std::vector<int> level_offset, level_len, level_edges;
};

}  // namespace bc
}  // namespace app
}  // namespace gunrock
