// app/pr/pr_enactor.hpp -- host loop of PageRank.
//
// Contract of the reference's PREnactor (gunrock/app/pr/pr_enactor.cuh:36-622):
//   template <bool INSTRUMENT> class PREnactor : EnactorBase
//   Enact<PRProblem>(context, problem, max_iteration, traversal_mode, max_grid_size)       (:536-618)
//   GetStatistics(total_queued, avg_duty, num_iter)                                        (:150-161)
// Schedule kept from EnactPR (:163-520):
//   1. peel off the vertices without out-edges, round by round, lowering the degree of their in-neighbours, until a round
//      removes nobody (:220-300);
//   2. iterate over the surviving vertices: distribute rank / degree along the edges, then per vertex
//      rank = delta * sum + (1 - delta) * [source or no source], count the vertices that moved by more than the threshold,
//      stop when none did or after max_iteration iterations (:312-498);
//   3. order the vertices by descending rank (:513-516).
// What differs: step 2's per-edge atomicAdd advance is a reducing advance over the in-neighbour lists (no per-edge atomics),
// step 1 uses the same operator over the out-neighbour lists, the rank copy / clear passes over all vertices per iteration are
// folded into the filter functor and the reduction's identity fill, and one packed tail is read back per round / iteration
// (the reference: 2-3 blocking 4-byte reads).
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/app/enactor_base.hpp>
#include <gunrock/app/pr/pr_functor.hpp>
#include <gunrock/app/pr/pr_problem.hpp>
#include <gunrock/oprtr/advance/kernel.hpp>
#include <gunrock/oprtr/filter/kernel.hpp>
#include <gunrock/util/context.hpp>

namespace gunrock {
namespace app {
namespace pr {

// start of a peeling round: flag the vertices that have just run out of out-edges, retire them
template <typename SizeT>
__global__ void PeelFlagKernel(const SizeT *d_degrees, SizeT *d_degrees_pong, int *d_zero_flag, long long nodes)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long v = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; v < nodes; v += stride) {
        const SizeT d = d_degrees[v];
        d_zero_flag[v] = d == 0 ? 1 : 0;
        d_degrees_pong[v] = d == 0 ? static_cast<SizeT>(-1) : d;  // (pr_functor.cuh:153-155)
    }
}

// per queue position: degree - (out-neighbours that were flagged)   (pr_functor.cuh:136-139, one atomicAdd(-1) per such edge)
template <typename VertexId, typename SizeT>
__global__ void PeelApplyKernel(const VertexId *d_queue, long long length, const int *d_zero_count, const SizeT *d_degrees, SizeT *d_degrees_pong)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < length; i += stride) {
        const VertexId v = d_queue[i];
        d_degrees_pong[v] = d_degrees[v] - static_cast<SizeT>(d_zero_count[i]);
    }
}

// before the first iteration: contributions of the survivors; after it the peeled vertices hold rank 0 like in the reference,
// whose whole-array copy of rank_next reaches them too (pr_enactor.cuh:478-485)
template <typename SizeT, typename Value>
__global__ void ContribKernel(const SizeT *d_degrees, const Value *d_rank_curr, Value *d_contrib, long long nodes)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long v = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; v < nodes; v += stride)
        d_contrib[v] = d_degrees[v] > 0 ? d_rank_curr[v] / static_cast<Value>(d_degrees[v]) : static_cast<Value>(0);
}
template <typename SizeT, typename Value>
__global__ void ZeroPeeledKernel(const SizeT *d_degrees, Value *d_rank_curr, long long nodes)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long v = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; v < nodes; v += stride)
        if (d_degrees[v] <= 0) d_rank_curr[v] = static_cast<Value>(0);
}

// (rank, vertex) -> one 64-bit key whose ascending order is "rank descending, then vertex ascending": the order a stable
// descending pair sort produces (util::CUBRadixSort<Value, VertexId>(false, ...), pr_enactor.cuh:513-516)
__device__ __forceinline__ unsigned DescendingFloatBits(float x)
{
    unsigned b = __float_as_uint(x);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // ascending order of floats as unsigned
    return ~b;
}
template <typename Value>
__global__ void RankKeysKernel(const Value *d_rank, long long nodes, unsigned long long *d_keys)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long v = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; v < nodes; v += stride)
        d_keys[v] = (static_cast<unsigned long long>(DescendingFloatBits(static_cast<float>(d_rank[v]))) << 32) | static_cast<unsigned>(v);
}
template <typename VertexId, typename Value>
__global__ void RankUnpackKernel(const unsigned long long *d_keys, const Value *d_rank, long long nodes, VertexId *d_node_ids, Value *d_sorted)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < nodes; i += stride) {
        const VertexId v = static_cast<VertexId>(d_keys[i] & 0xFFFFFFFFull);
        d_node_ids[i] = v;
        d_sorted[i] = d_rank[v];
    }
}

template <bool INSTRUMENT>
class PREnactor : public EnactorBase {
   public:
    explicit PREnactor(bool DEBUG = false) : EnactorBase(VERTEX_FRONTIERS, DEBUG) {}
    ~PREnactor() override {}

    void GetStatistics(long long &total_queued, double &avg_duty, long long &num_iter)
    {
        total_queued = enactor_stats.total_queued;
        avg_duty = (enactor_stats.total_lifetimes > 0) ? enactor_stats.total_runtimes / enactor_stats.total_lifetimes : 0.0;
        num_iter = enactor_stats.iteration;
    }
    long long PeelingRounds() const { return peeling_rounds; }
    long long SurvivingNodes() const { return surviving; }

    typedef oprtr::advance::KernelPolicy<256, 8, 8, oprtr::advance::LB> AdvancePolicy;
    typedef oprtr::filter::KernelPolicy<256, 4, 8> FilterPolicy;

    template <typename PRProblem>
    hipError_t Enact(util::DeviceContext & /*context*/, PRProblem *problem, typename PRProblem::SizeT max_iteration,
                     int /*traversal_mode: the reference's LB / TWC choice*/ = 0, int max_grid_size = 0)
    {
        typedef typename PRProblem::VertexId VertexId;
        typedef typename PRProblem::SizeT SizeT;
        typedef typename PRProblem::Value Value;
        typedef PRFunctor<VertexId, SizeT, Value, PRProblem> PrFunctor;
        typedef RemoveZeroDegreeNodeFunctor<VertexId, SizeT, Value, PRProblem> RemoveZeroFunctor;
        typedef HasOutEdgesFunctor<VertexId, SizeT, Value, PRProblem> HasEdgesFunctor;
        using oprtr::advance::PLUS;
        using oprtr::advance::VERTEX;

        hipError_t retval = hipSuccess;
        if ((retval = EnactorBase::Setup(max_grid_size, AdvancePolicy::MIN_BLOCKS, FilterPolicy::MIN_BLOCKS))) return retval;
        GraphSlice<VertexId, SizeT, Value> *gs = problem->graph_slices[0];
        typename PRProblem::DataSlice *ds = problem->data_slices[0];
        hipStream_t stream = gs->stream;
        const long long n = problem->nodes;
        peeling_rounds = 0;
        surviving = 0;
        if (n <= 0) return retval;
        if (!problem->d_inv_row_offsets) return util::GRError(hipErrorNotInitialized, "PREnactor: no in-neighbour lists (SetInverseGraph)", __FILE__, __LINE__);
        if ((retval = work_progress.Reset(stream))) return retval;
        const int sweep = cu_count * 8;
        int slot = 0;  // ring slot that receives the next packed tail
        // two ring slots take turns: the kernel that fills one clears the other, which the host has read by then
        auto read_tail = [&](unsigned &len, unsigned &edges) -> hipError_t {
            hipError_t rc = work_progress.GetTail(slot, len, edges, stream);
            slot ^= 1;
            return rc;
        };

        // ---- 1. every vertex that has out-edges, with the degree prefix of its out-list ----
        int selector = 0;
        unsigned len = 0, edges = 0;
        {
            oprtr::filter::FilterArgs<VertexId, SizeT> f;
            f.d_in = nullptr;  // identity queue 0 .. n-1
            f.num_elements = static_cast<SizeT>(n);
            f.out = gs->frontier_queues[selector];
            f.d_tail_out = work_progress.d_tail + slot;
            f.d_tail_clear = work_progress.d_tail + (slot ^ 1);
            f.d_overflow = work_progress.d_overflow;
            f.d_row_offsets = gs->d_row_offsets;
            if ((retval = oprtr::filter::LaunchKernel<FilterPolicy, PRProblem, HasEdgesFunctor, true>(f, *ds, enactor_stats.filter_grid_size, stream)))
                return retval;
            if ((retval = read_tail(len, edges))) return retval;
        }
        // ---- peeling rounds ----
        for (;;) {
            hipLaunchKernelGGL((PeelFlagKernel<SizeT>), dim3(sweep), dim3(256), 0, stream, ds->d_degrees, ds->d_degrees_pong, ds->d_zero_flag, n);
            if ((retval = util::GRError("PeelFlagKernel launch failed", __FILE__, __LINE__))) return retval;
            if (len > 0) {
                oprtr::advance::AdvanceArgs<VertexId, SizeT> a;
                a.in = gs->frontier_queues[selector];
                a.in_len = static_cast<SizeT>(len);
                a.in_edges = static_cast<SizeT>(edges);
                a.d_row_offsets = gs->d_row_offsets;
                a.d_column_indices = gs->d_column_indices;
                a.d_tail_out = nullptr;
                a.d_tail_clear = nullptr;
                a.d_overflow = work_progress.d_overflow;
                if ((retval = oprtr::advance::LaunchReduce<AdvancePolicy, PRProblem, RemoveZeroFunctor, VERTEX, PLUS, int>(
                         a, *ds, ds->d_zero_flag, ds->d_zero_count, max_grid_size, stream)))
                    return retval;
                hipLaunchKernelGGL((PeelApplyKernel<VertexId, SizeT>), dim3(sweep), dim3(256), 0, stream, gs->frontier_queues[selector].v,
                                   static_cast<long long>(len), ds->d_zero_count, ds->d_degrees, ds->d_degrees_pong);
                if ((retval = util::GRError("PeelApplyKernel launch failed", __FILE__, __LINE__))) return retval;
            }
            unsigned new_len = 0, new_edges = 0;
            if (len > 0) {
                oprtr::filter::FilterArgs<VertexId, SizeT> f;
                f.d_in = gs->frontier_queues[selector].v;
                f.num_elements = static_cast<SizeT>(len);
                f.out = gs->frontier_queues[selector ^ 1];
                f.d_tail_out = work_progress.d_tail + slot;
                f.d_tail_clear = work_progress.d_tail + (slot ^ 1);
                f.d_overflow = work_progress.d_overflow;
                f.d_row_offsets = gs->d_row_offsets;
                if ((retval = oprtr::filter::LaunchKernel<FilterPolicy, PRProblem, RemoveZeroFunctor, true>(f, *ds, enactor_stats.filter_grid_size, stream)))
                    return retval;
                if ((retval = read_tail(new_len, new_edges))) return retval;
            }
            SizeT *tmp = ds->d_degrees;  // degrees <- degrees_pong (the reference copies the vector, pr_enactor.cuh:292-296)
            ds->d_degrees = ds->d_degrees_pong;
            ds->d_degrees_pong = tmp;
            ++peeling_rounds;
            const bool changed = new_len != len;
            selector ^= (len > 0) ? 1 : 0;
            len = new_len;
            edges = new_edges;
            if (!changed) break;
        }
        surviving = len;
        const util::Frontier<VertexId, SizeT> &alive = gs->frontier_queues[selector];

        // ---- 2. rank iterations over the survivors ----
        unsigned inv_len = 0, inv_edges = 0;
        if (len > 0) {  // the survivors again, now with the degree prefix of their IN-lists (vertices nobody points at drop out:
                        // their sum stays 0)
            oprtr::filter::FilterArgs<VertexId, SizeT> f;
            f.d_in = alive.v;
            f.num_elements = static_cast<SizeT>(len);
            f.out = problem->inv_frontier;
            f.d_tail_out = work_progress.d_tail + slot;
            f.d_tail_clear = work_progress.d_tail + (slot ^ 1);
            f.d_overflow = work_progress.d_overflow;
            f.d_row_offsets = problem->d_inv_row_offsets;
            if ((retval = oprtr::filter::LaunchKernel<FilterPolicy, PRProblem, HasEdgesFunctor, true>(f, *ds, enactor_stats.filter_grid_size, stream)))
                return retval;
            if ((retval = read_tail(inv_len, inv_edges))) return retval;
        }
        hipLaunchKernelGGL((ContribKernel<SizeT, Value>), dim3(sweep), dim3(256), 0, stream, ds->d_degrees, ds->d_rank_curr, ds->d_contrib, n);
        if ((retval = util::GRError("ContribKernel launch failed", __FILE__, __LINE__))) return retval;
        while (len > 0) {
            if (INSTRUMENT && (retval = InstrumentBegin(stream))) return retval;
            oprtr::advance::AdvanceArgs<VertexId, SizeT> a;
            a.in = problem->inv_frontier;
            a.in_len = static_cast<SizeT>(inv_len);
            a.in_edges = static_cast<SizeT>(inv_edges);
            a.d_row_offsets = problem->d_inv_row_offsets;
            a.d_column_indices = problem->d_inv_column_indices;
            a.d_tail_out = nullptr;
            a.d_tail_clear = nullptr;
            a.d_overflow = work_progress.d_overflow;
            // rank_next[v] = sum over in-neighbours u of contrib[u]   (identity fill = the reference's rank_next <- 0 pass)
            if ((retval = oprtr::advance::LaunchReduce<AdvancePolicy, PRProblem, PrFunctor, VERTEX, PLUS, Value, true>(
                     a, *ds, ds->d_contrib, ds->d_rank_next, max_grid_size, stream, n)))
                return retval;
            oprtr::filter::FilterArgs<VertexId, SizeT> f;
            f.d_in = alive.v;
            f.num_elements = static_cast<SizeT>(len);
            f.out = gs->frontier_queues[selector ^ 1];  // (scratch: the reference passes no output queue at all)
            f.d_tail_out = work_progress.d_tail + slot;
            f.d_tail_clear = work_progress.d_tail + (slot ^ 1);
            f.d_overflow = work_progress.d_overflow;
            f.d_row_offsets = gs->d_row_offsets;
            if ((retval = oprtr::filter::LaunchKernel<FilterPolicy, PRProblem, PrFunctor, false>(f, *ds, enactor_stats.filter_grid_size, stream)))
                return retval;
            if (enactor_stats.iteration == 0) {
                hipLaunchKernelGGL((ZeroPeeledKernel<SizeT, Value>), dim3(sweep), dim3(256), 0, stream, ds->d_degrees, ds->d_rank_curr, n);
                if ((retval = util::GRError("ZeroPeeledKernel launch failed", __FILE__, __LINE__))) return retval;
            }
            if (INSTRUMENT && (retval = InstrumentEnd(stream))) return retval;
            unsigned active = 0, unused = 0;
            if ((retval = read_tail(active, unused))) return retval;
            if (INSTRUMENT) InstrumentCollect(len, inv_edges, 0);
            enactor_stats.iteration++;
            enactor_stats.total_queued += active;
            if (DEBUG) std::printf("iteration %lld: %u vertices moved by more than the threshold\n", enactor_stats.iteration, active);
            if (active == 0 || enactor_stats.iteration >= max_iteration) break;
        }
        if (len == 0) {
            // Peeling removed every vertex (any DAG): the reference's `while (done[0] < 0)` loop still runs ONE pass over its empty
            // queue (pr_enactor.cuh:341-498) -- iteration becomes 1 and the whole-array rank_next -> rank_curr copy (:478-482)
            // leaves every rank at 0, not at the initial 1 - delta.
            hipLaunchKernelGGL((ZeroPeeledKernel<SizeT, Value>), dim3(sweep), dim3(256), 0, stream, ds->d_degrees, ds->d_rank_curr, n);
            if ((retval = util::GRError("ZeroPeeledKernel launch failed", __FILE__, __LINE__))) return retval;
            enactor_stats.iteration = 1;
        }

        // ---- 3. vertices by descending rank ----
        if ((retval = problem->sorter.Reserve(n))) return retval;
        hipLaunchKernelGGL((RankKeysKernel<Value>), dim3(sweep), dim3(256), 0, stream, ds->d_rank_curr, n, problem->sorter.Keys());
        if ((retval = util::GRError("RankKeysKernel launch failed", __FILE__, __LINE__))) return retval;
        unsigned long long *sorted = nullptr;
        if ((retval = problem->sorter.Sort(n, 64, stream, &sorted))) return retval;
        hipLaunchKernelGGL((RankUnpackKernel<VertexId, Value>), dim3(sweep), dim3(256), 0, stream, sorted, ds->d_rank_curr, n, ds->d_node_ids,
                           ds->d_rank_sorted);
        if ((retval = util::GRError("RankUnpackKernel launch failed", __FILE__, __LINE__))) return retval;
        bool overflow = false;
        if ((retval = work_progress.CheckOverflow(overflow, stream))) return retval;
        if (overflow) retval = util::GRError(hipErrorInvalidConfiguration, "Frontier queue overflow. Please increase queue-sizing factor.", __FILE__, __LINE__);
        return retval;
    }

   private:
    long long peeling_rounds = 0;
    long long surviving = 0;
};

}  // namespace pr
}  // namespace app
}  // namespace gunrock
