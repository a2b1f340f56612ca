// app/sssp/sssp_enactor.hpp -- host loop for delta-stepping SSSP with a near/far pile.
//
// Public contract of the reference's SSSPEnactor (gunrock/app/sssp/sssp_enactor.cuh:36-563):
//   template <bool INSTRUMENT> class SSSPEnactor : EnactorBase
//   Enact<SSSPProblem>(context, problem, src, queue_sizing, max_grid_size = 0, traversal_mode = 0)   (:485-563)
//   GetStatistics(total_queued, search_depth, avg_duty)                                              (:150-170)
// Loop shape kept from EnactSSSP (:284-431): advance with the relaxation functor -> split the improved vertices
// into near (next frontier) and far (parked) -> when near runs dry raise the priority level and re-split the far
// pile until something is near.  The reference also runs a compacting filter between advance and split
// (:354-370); the advance here emits a hole-free queue, so that pass has nothing left to do.  Per iteration the
// reference makes 1 + 2 blocking reads and allocates the pile inside Enact (:235-238); here: two 8-byte reads,
// no allocation.
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/app/enactor_base.hpp>
#include <gunrock/app/sssp/sssp_functor.hpp>
#include <gunrock/app/sssp/sssp_problem.hpp>
#include <gunrock/oprtr/advance/kernel.hpp>
#include <gunrock/oprtr/filter/kernel.hpp>
#include <gunrock/priority_queue/kernel.hpp>
#include <gunrock/util/context.hpp>

namespace gunrock {
namespace app {
namespace sssp {

template <bool INSTRUMENT>
class SSSPEnactor : public EnactorBase {
   public:
    explicit SSSPEnactor(bool DEBUG = false) : EnactorBase(VERTEX_FRONTIERS, DEBUG) {}
    ~SSSPEnactor() override {}

    long long relaxed_vertices = 0;  // vertices dequeued by advances (sum of frontier lengths)
    long long relaxed_edges = 0;     // edge slots expanded
    long long pull_levels = 0;       // levels relaxed by pulling over the in-neighbour lists

    void GetStatistics(long long &total_queued, long long &search_depth, double &avg_duty)
    {
        total_queued = enactor_stats.total_queued;
        search_depth = enactor_stats.iteration;
        avg_duty = 0.0;
    }
    void GetKernelStatistics(long long &launches, double &kernel_ms)
    {
        launches = enactor_stats.kernel_launches;
        kernel_ms = enactor_stats.kernel_ms;
    }

    typedef oprtr::advance::KernelPolicy<256, 8, 4, oprtr::advance::LB> AdvancePolicy;

    template <typename SSSPProblem>
    hipError_t Enact(util::DeviceContext & /*context*/, SSSPProblem *problem, typename SSSPProblem::VertexId src,
                     double /*queue_sizing*/ = 1.0, int max_grid_size = 0, int /*traversal_mode*/ = 0)
    {
        typedef typename SSSPProblem::VertexId VertexId;
        typedef typename SSSPProblem::SizeT SizeT;
        typedef typename SSSPProblem::Value Value;
        typedef SSSPFunctor<VertexId, SizeT, SSSPProblem> SsspFunctor;
        typedef PQFunctor<VertexId, SizeT, SSSPProblem> PqFunctor;
        typedef SSSPPullFunctor<VertexId, SizeT, SSSPProblem> PullFunctor;
        typedef typename PullFunctor::PullValue PullValue;
        typedef oprtr::filter::KernelPolicy<256, 4, 8> FilterPolicy;

        hipError_t retval = hipSuccess;
        if ((retval = EnactorBase::Setup(max_grid_size, AdvancePolicy::MIN_BLOCKS, 8))) return retval;
        GraphSlice<VertexId, SizeT, Value> *gs = problem->graph_slices[0];
        typename SSSPProblem::DataSlice *ds = problem->data_slices[0];
        hipStream_t stream = gs->stream;
        relaxed_vertices = relaxed_edges = 0;
        pull_levels = 0;
        if (src < 0 || src >= problem->nodes) return retval;
        if ((retval = work_progress.Reset(stream))) return retval;

        // tail words: [0] advance output (candidates), [1] near frontier, [2]/[3] far pile (ping-pong)
        unsigned long long *d_tail = work_progress.d_tail;
        unsigned long long *h_tail = work_progress.h_tail;
        // one read-back per kernel; it also re-arms the words the NEXT kernel expects clean (advance output count, near tail,
        // the far pile that is not in use, the far-minimum bucket) instead of a hipMemsetAsync per word
        int far_selector = 0;  // far pile ping-pong: pile lives in d_far_*[far_selector], tail word 2 + far_selector
        auto read_tails = [&]() -> hipError_t {
            return work_progress.Sync(stream, 0x3u | (1u << (2 + (far_selector ^ 1))), 1u << 5);
        };
        unsigned *d_far_min = reinterpret_cast<unsigned *>(d_tail + 5);
        unsigned *h_far_min = reinterpret_cast<unsigned *>(h_tail + 5);
        auto arm_far_min = [&]() -> hipError_t {
            return util::GRError(hipMemsetAsync(d_far_min, 0xFF, sizeof(unsigned), stream), "SSSPEnactor arm far-min failed",
                                 __FILE__, __LINE__);
        };

        unsigned queue_length = problem->SourceDegree() > 0 ? 1u : 0u;
        unsigned queue_edges = static_cast<unsigned>(problem->SourceDegree());
        unsigned far_length = 0;
        unsigned far_min_bucket = 0xFFFFFFFFu;  // smallest bucket currently parked (lower bound)
        unsigned level = 0;
        int selector = 0;      // frontier ping-pong
        int tag = 0;
        if ((retval = arm_far_min())) return retval;  // (the first read-back re-arms from then on)
        const int grid = enactor_stats.advance_grid_size;

        while (queue_length > 0 || far_length > 0) {
            if (queue_length > 0) {
                relaxed_vertices += queue_length;
                relaxed_edges += queue_edges;
                enactor_stats.total_queued += queue_length;
                // ---- advance: relax every out-edge of the frontier; improved destinations -> candidates ----
                oprtr::advance::AdvanceArgs<VertexId, SizeT> args;
                args.in = gs->frontier_queues[selector];
                args.out = util::Frontier<VertexId, SizeT>();
                args.out.v = problem->d_candidates;
                args.out.capacity = problem->candidate_capacity;
                args.in_len = static_cast<SizeT>(queue_length);
                args.in_edges = static_cast<SizeT>(queue_edges);
                args.d_row_offsets = gs->d_row_offsets;
                args.d_column_indices = gs->d_column_indices;
                args.d_tail_out = d_tail + 0;
                args.d_tail_clear = nullptr;
                args.d_overflow = work_progress.d_overflow;
                if (INSTRUMENT && (retval = InstrumentBegin(stream))) break;
                const bool pull = problem->HasInverse() && problem->pull_min_edges != 0 &&
                                  static_cast<long long>(queue_edges) > problem->PullMinEdges();
                if (pull) {
                    // ---- dense level: every vertex takes the minimum over its in-edges of (neighbour's distance + weight), one
                    //      reducing advance over the in-neighbour lists, no atomic per edge; then a sweep over the vertices keeps
                    //      the improvements and hands the improved vertices to the split as the push would ----
                    ++pull_levels;
                    relaxed_edges += problem->inv_frontier_edges - queue_edges;  // (what was actually swept)
                    oprtr::advance::AdvanceArgs<VertexId, SizeT> pa;
                    pa.in = problem->inv_frontier;
                    pa.in_len = problem->inv_frontier_len;
                    pa.in_edges = problem->inv_frontier_edges;
                    pa.d_row_offsets = problem->d_inv_row_offsets;
                    pa.d_column_indices = problem->d_inv_column_indices;
                    pa.d_tail_out = nullptr;
                    pa.d_tail_clear = nullptr;
                    pa.d_overflow = work_progress.d_overflow;
                    if ((retval = oprtr::advance::LaunchReduce<AdvancePolicy, SSSPProblem, PullFunctor, oprtr::advance::VERTEX,
                                                               oprtr::advance::MINIMUM, PullValue, true>(
                             pa, *ds, static_cast<const PullValue *>(nullptr), static_cast<PullValue *>(ds->d_pull), max_grid_size, stream,
                             static_cast<long long>(problem->nodes))))
                        break;
                    oprtr::filter::FilterArgs<VertexId, SizeT> f;
                    f.d_in = nullptr;  // every vertex
                    f.num_elements = problem->nodes;
                    f.out = util::Frontier<VertexId, SizeT>();
                    f.out.v = problem->d_candidates;
                    f.out.capacity = problem->candidate_capacity;
                    f.d_tail_out = d_tail + 0;
                    f.d_tail_clear = nullptr;
                    f.d_overflow = work_progress.d_overflow;
                    f.d_row_offsets = gs->d_row_offsets;
                    if ((retval = oprtr::filter::LaunchKernel<FilterPolicy, SSSPProblem, PullFunctor, false>(f, *ds, cu_count * 8, stream))) break;
                } else if ((retval = oprtr::advance::LaunchKernel<AdvancePolicy, SSSPProblem, SsspFunctor, false>(
                                args, *ds, max_grid_size, stream, oprtr::advance::V2V)))
                    break;
                if (INSTRUMENT && (retval = InstrumentEnd(stream))) break;
                // The split follows the advance on the stream: it reads the candidate count from the advance's device tail, so the
                // host makes ONE round trip per iteration, not two (the instrumented enactor keeps the two, for per-kernel times).
                unsigned candidates = static_cast<unsigned>(problem->candidate_capacity);  // (upper bound: sizes the split's grid)
                if (INSTRUMENT) {
                    if ((retval = read_tails())) break;
                    InstrumentCollect(queue_length, queue_edges, 0);
                    candidates = util::TailCount(h_tail[0]);
                }

                // ---- split candidates into near (next frontier) and far (parked) ----
                queue_length = 0;
                queue_edges = 0;
                if (candidates > 0) {
                    priority_queue::BisectArgs<VertexId, SizeT> b;
                    b.d_far_min = d_far_min;
                    b.d_in = problem->d_candidates;
                    b.d_in_dist = nullptr;
                    b.num_elements = static_cast<SizeT>(candidates);
                    b.d_num_elements = INSTRUMENT ? nullptr : d_tail + 0;
                    b.level = level;
                    b.tag = ++tag;
                    b.near = gs->frontier_queues[selector ^ 1];
                    b.d_near_tail = d_tail + 1;
                    b.d_far_v = problem->d_far_v[far_selector];
                    b.d_far_d = problem->d_far_d[far_selector];
                    b.far_capacity = problem->far_capacity;
                    b.d_far_tail = d_tail + 2 + far_selector;
                    b.d_overflow = work_progress.d_overflow;
                    b.d_row_offsets = gs->d_row_offsets;
                    if (INSTRUMENT && (retval = InstrumentBegin(stream))) break;
                    if ((retval = priority_queue::Bisect<256, 4, SSSPProblem, PqFunctor>(b, *ds, grid, stream))) break;
                    if (INSTRUMENT && (retval = InstrumentEnd(stream))) break;
                    if ((retval = read_tails())) break;
                    if (INSTRUMENT) InstrumentCollect(candidates, 0, 1);
                    queue_length = util::TailCount(h_tail[1]);
                    queue_edges = util::TailEdges(h_tail[1]);
                    far_length = util::TailCount(h_tail[2 + far_selector]);
                    if (*h_far_min < far_min_bucket) far_min_bucket = *h_far_min;
                    selector ^= 1;
                }
                ++enactor_stats.iteration;
            }
            // ---- near ran dry: raise the level and re-split the far pile until something is near ----
            while (queue_length == 0 && far_length > 0) {
                // next level; skip levels no parked vertex can be in
                ++level;
                if (far_min_bucket != 0xFFFFFFFFu && far_min_bucket > level) level = far_min_bucket;
                priority_queue::BisectArgs<VertexId, SizeT> b;
                b.d_far_min = d_far_min;
                b.d_in = problem->d_far_v[far_selector];
                b.d_in_dist = problem->d_far_d[far_selector];
                b.num_elements = static_cast<SizeT>(far_length);
                b.level = level;
                b.tag = ++tag;
                b.near = gs->frontier_queues[selector];
                b.d_near_tail = d_tail + 1;
                b.d_far_v = problem->d_far_v[far_selector ^ 1];
                b.d_far_d = problem->d_far_d[far_selector ^ 1];
                b.far_capacity = problem->far_capacity;
                b.d_far_tail = d_tail + 2 + (far_selector ^ 1);
                b.d_overflow = work_progress.d_overflow;
                b.d_row_offsets = gs->d_row_offsets;
                if (INSTRUMENT && (retval = InstrumentBegin(stream))) break;
                if ((retval = priority_queue::Bisect<256, 4, SSSPProblem, PqFunctor>(b, *ds, grid, stream))) break;
                if (INSTRUMENT && (retval = InstrumentEnd(stream))) break;
                far_selector ^= 1;  // (before the read-back: it clears the tail of the pile that is out of use from now on)
                if ((retval = read_tails())) break;
                if (INSTRUMENT) InstrumentCollect(far_length, 0, 2);
                queue_length = util::TailCount(h_tail[1]);
                queue_edges = util::TailEdges(h_tail[1]);
                far_length = util::TailCount(h_tail[2 + far_selector]);
                far_min_bucket = *h_far_min;  // the pile was rewritten: this call saw every entry
            }
            if (retval) break;
        }
        if (retval) return retval;
        bool overflow = false;
        if ((retval = work_progress.CheckOverflow(overflow, stream))) return retval;
        if (overflow)
            retval = util::GRError(hipErrorInvalidConfiguration, "Frontier queue overflow. Please increase queue-sizing factor.",
                                   __FILE__, __LINE__);
        return retval;
    }
};

}  // namespace sssp
}  // namespace app
}  // namespace gunrock
