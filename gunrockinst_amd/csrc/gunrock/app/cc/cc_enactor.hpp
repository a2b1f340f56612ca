// app/cc/cc_enactor.hpp -- host loop for connected components.
//
// Public contract of the reference's CCEnactor (gunrock/app/cc/cc_enactor.cuh:36-919):
//   template <bool INSTRUMENT> class CCEnactor : EnactorBase;  Enact<CCProblem>(problem, max_grid_size)
//   GetStatistics(total_queued, num_iter, avg_duty)                                           (:140-160)
// Schedule kept from EnactCC (:165-873): HookInit over edges -> PtrJump over vertices until stable ->
// UpdateMask -> repeat { HookMax over edges; stop if nothing hooked; PtrJumpMask until stable; PtrJumpUnmask;
// UpdateMask }.  Every sweep is the filter operator in its non-compacting form with one of the functors of
// cc_functor.hpp (the reference launches filter::Kernel with filtering_flag=false the same way, :407-424).
// Differences: the two convergence flags live next to each other in HBM and are reset by one 8-byte
// hipMemsetAsync instead of a blocking 4-byte H2D copy per sweep (:445-450, 532-538); sweeps take the identity
// queue (NULL) rather than reading an iota array; launch grids fill 256 CUs.
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/app/cc/cc_functor.hpp>
#include <gunrock/app/cc/cc_problem.hpp>
#include <gunrock/app/enactor_base.hpp>
#include <gunrock/oprtr/filter/kernel.hpp>

namespace gunrock {
namespace app {
namespace cc {

template <bool INSTRUMENT>
class CCEnactor : public EnactorBase {
   public:
    explicit CCEnactor(bool DEBUG = false) : EnactorBase(EDGE_FRONTIERS, DEBUG) {}
    ~CCEnactor() override {}

    long long edge_sweeps = 0;    // I_h of SURVEY 8(d)
    long long vertex_sweeps = 0;  // I_j

    void GetStatistics(long long &total_queued, long long &num_iter, double &avg_duty)
    {
        total_queued = enactor_stats.total_queued;
        num_iter = enactor_stats.iteration;
        avg_duty = 0.0;
    }
    void GetKernelStatistics(long long &launches, double &kernel_ms)
    {
        launches = enactor_stats.kernel_launches;
        kernel_ms = enactor_stats.kernel_ms;
    }

    typedef oprtr::filter::KernelPolicy<256, 4, 8> FilterPolicy;

    template <typename CCProblem>
    hipError_t Enact(CCProblem *problem, int max_grid_size = 0)
    {
        typedef typename CCProblem::VertexId VertexId;
        typedef typename CCProblem::SizeT SizeT;
        typedef typename CCProblem::Value Value;
        typedef UpdateMaskFunctor<VertexId, SizeT, Value, CCProblem> UpdateMask;
        typedef HookInitFunctor<VertexId, SizeT, Value, CCProblem> HookInit;
        typedef HookMaxFunctor<VertexId, SizeT, Value, CCProblem> HookMax;
        typedef PtrJumpFunctor<VertexId, SizeT, Value, CCProblem> PtrJump;
        typedef PtrJumpMaskFunctor<VertexId, SizeT, Value, CCProblem> PtrJumpMask;
        typedef PtrJumpUnmaskFunctor<VertexId, SizeT, Value, CCProblem> PtrJumpUnmask;

        hipError_t retval = hipSuccess;
        if ((retval = EnactorBase::Setup(max_grid_size, 8, 8))) return retval;
        const int grid = enactor_stats.filter_grid_size * 2;
        typename CCProblem::DataSlice *ds = problem->data_slices[0];
        hipStream_t stream = problem->graph_slices[0]->stream;
        const SizeT n = problem->nodes, m = problem->sweep_edges;  // (mirrored input: the from > to orientation only, cc_problem.hpp)
        edge_sweeps = vertex_sweeps = 0;
        if (n <= 0) return retval;

#define GR_CC_SWEEP(Functor, count, kind)                                                                            \
    do {                                                                                                             \
        if (INSTRUMENT && (retval = InstrumentBegin(stream))) return retval;                                         \
        if ((retval = oprtr::filter::LaunchApply<FilterPolicy, CCProblem, Functor>(nullptr, (count), nullptr, slice, \
                                                                                    grid, stream)))                 \
            return retval;                                                                                           \
        if (INSTRUMENT) {                                                                                            \
            if ((retval = InstrumentEnd(stream))) return retval;                                                     \
            GR_CHECK(hipStreamSynchronize(stream), "CCEnactor sync failed");                                         \
            InstrumentCollect((count), 0, (kind));                                                                   \
        }                                                                                                            \
        if (kind) ++edge_sweeps; else ++vertex_sweeps;                                                               \
        enactor_stats.total_queued += (count);                                                                       \
    } while (0)

        // Convergence flags: the two ints of WorkProgress slot 0 (vertex flag low, edge flag high).  A sweep that changes
        // something clears its flag; poll() is the read-back kernel, which mirrors the word to pinned memory and sets it back
        // to all ones for the next sweep -- no copy up, no copy down, no stream synchronisation per sweep.
        typename CCProblem::DataSlice slice = *ds;
        slice.d_vertex_flag = reinterpret_cast<int *>(work_progress.d_tail);
        slice.d_edge_flag = slice.d_vertex_flag + 1;
        if ((retval = work_progress.Reset(stream))) return retval;
        GR_CHECK(hipMemsetAsync(work_progress.d_tail, 0xFF, sizeof(unsigned long long), stream), "CCEnactor arm flags failed");
        bool vertex_stable = false, edge_stable = false;
        int hook_sweeps = 0;
        auto poll = [&]() -> hipError_t {
            hipError_t rc = work_progress.Sync(stream, 0u, 1u);
            vertex_stable = (work_progress.h_tail[0] & 0xFFFFFFFFull) != 0;
            edge_stable = (work_progress.h_tail[0] >> 32) != 0;
            return rc;
        };

        if (m > 0) {
            if (slice.d_first_lower) {  // mirrored input, compact layout: the opening hooks as a vertex sweep (cc_functor.hpp)
                typedef HookInitRowFunctor<VertexId, SizeT, Value, CCProblem> HookInitRow;
                GR_CC_SWEEP(HookInitRow, n, 1);
            } else
                GR_CC_SWEEP(HookInit, m, 1);
        }
        for (;;) {  // first pointer-jumping round (cc_enactor.cuh:442-493)
            GR_CC_SWEEP(PtrJump, n, 0);
            if ((retval = poll())) return retval;
            ++enactor_stats.iteration;
            if (vertex_stable) break;
        }
        GR_CC_SWEEP(UpdateMask, n, 0);

        while (m > 0) {  // cc_enactor.cuh:524-862
            // From the third hooking sweep on nearly every edge is marked done, and the sweep is a scan of the flags: it tests 16 of
            // them per lane (filter::LaunchApplySkip: 272 -> 50-66 us at scale-24).  The first two sweeps still touch most edges, and
            // there one edge per lane keeps the endpoint loads coalesced (16 consecutive edges per lane: 1.4 -> 7.8 ms for the first sweep;
            // 4 per lane: +0.7 ms).
            if (INSTRUMENT && (retval = InstrumentBegin(stream))) return retval;
            if (hook_sweeps >= 2) {
                if ((retval = oprtr::filter::LaunchApplySkip<FilterPolicy, CCProblem, HookMax>(m, slice.d_marks, slice, grid, stream))) return retval;
            } else if ((retval = oprtr::filter::LaunchApply<FilterPolicy, CCProblem, HookMax>(nullptr, m, nullptr, slice, grid, stream)))
                return retval;
            ++hook_sweeps;
            if (INSTRUMENT) {
                if ((retval = InstrumentEnd(stream))) return retval;
                GR_CHECK(hipStreamSynchronize(stream), "CCEnactor sync failed");
                InstrumentCollect(m, 0, 1);
            }
            ++edge_sweeps;
            enactor_stats.total_queued += m;
            if ((retval = poll())) return retval;
            ++enactor_stats.iteration;
            if (edge_stable) break;  // no edge hooked anything: done
            for (;;) {
                GR_CC_SWEEP(PtrJumpMask, n, 0);
                if ((retval = poll())) return retval;
                if (vertex_stable) break;
            }
            GR_CC_SWEEP(PtrJumpUnmask, n, 0);
            GR_CC_SWEEP(UpdateMask, n, 0);
        }
#undef GR_CC_SWEEP
        return retval;
    }
};

}  // namespace cc
}  // namespace app
}  // namespace gunrock
