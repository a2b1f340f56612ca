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
#include <cstdlib>

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
                GR_CC_SWEEP(HookInitRow, n, 0);  // (a vertex sweep: n parents written from a per-vertex array, no edge is read)
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

        // ---- row form of the hooking sweeps (cc_functor.hpp HookMaxRowFunctor): mirrored input whose sampled giant component
        //      holds at least a quarter of the samples.  It starts AFTER the first edge-form sweep and its jumps: before that the
        //      giant component is still a forest of many trees (measured: started at the first sweep, a vertex with a long row
        //      outside the sampled tree sent the run back to the edge form at once, 1.68 -> 1.81 ms). ----
        bool row_form = false;
        const bool row_form_wanted = m > 0 && slice.d_first_lower && slice.d_row_offsets && problem->row_form;  // (rows: every edge mirrored)
        VertexId *d_giant = reinterpret_cast<VertexId *>(work_progress.d_tail + 4);  // slots 4, 5: mirrored by every read-back
        if (row_form_wanted) {
            slice.d_giant = d_giant;
            if (const char *env = std::getenv("GUNROCK_CC_ROW_LIMIT")) slice.row_form_limit = static_cast<SizeT>(std::atoi(env));
        }
        auto pick_giant = [&]() -> hipError_t {  // (queued in front of a read-back the schedule makes anyway)
            hipLaunchKernelGGL((PickGiantKernel<VertexId>), dim3(1), dim3(kGiantSamples), 0, stream, ds->d_component_ids, static_cast<long long>(n), d_giant);
            return util::GRError("PickGiantKernel launch failed", __FILE__, __LINE__);
        };

        // ---- neighbour rounds (cc_functor.hpp HookNeighbourFunctor): HookMax over ONE edge per vertex, then the jumps of a
        //      hooking round.  The opening already took every vertex's smallest lower neighbour; round r takes the (r + 1)-th.
        //      After them the giant is sampled, and if it dominates, the first FULL hooking sweep is already in row form. ----
        if (row_form_wanted && slice.d_low_offsets) {
            typedef HookNeighbourFunctor<VertexId, SizeT, Value, CCProblem> HookNeighbour;
            for (int r = 1; r <= problem->neighbour_rounds; ++r) {
                slice.neighbour_round = r;
                GR_CC_SWEEP(HookNeighbour, n, 0);
                for (;;) {
                    GR_CC_SWEEP(PtrJumpMask, n, 0);
                    if ((retval = poll())) return retval;
                    ++enactor_stats.iteration;
                    if (vertex_stable) break;
                }
                GR_CC_SWEEP(PtrJumpUnmask, n, 0);
                GR_CC_SWEEP(UpdateMask, n, 0);
            }
            if (problem->neighbour_rounds > 0) {
                if ((retval = pick_giant())) return retval;
                if ((retval = poll())) return retval;
                const long long samples = n >= kGiantSamples ? kGiantSamples : n;
                row_form = static_cast<long long>(work_progress.h_tail[4] >> 32) * 4 >= samples;
            }
        }

        bool hooks_pending = false;  // a row-form sweep of this round hooked something before the edge form took over
        while (m > 0) {  // cc_enactor.cuh:524-862
            if (row_form) {
                typedef HookMaxRowFunctor<VertexId, SizeT, Value, CCProblem> HookMaxRow;
                GR_CC_SWEEP(HookMaxRow, n, 0);  // (counted with the vertex sweeps: it reads the parents, not the edge list)
                if ((retval = poll())) return retval;
                const bool rows_hooked = !edge_stable;
                const bool long_row = (work_progress.h_tail[5] & 0xFFFFFFFFull) != 0;
                if (!long_row) {
                    ++enactor_stats.iteration;
                    if (!rows_hooked) break;
                    for (;;) {
                        GR_CC_SWEEP(PtrJumpMask, n, 0);
                        if ((retval = poll())) return retval;
                        if (vertex_stable) break;
                    }
                    GR_CC_SWEEP(PtrJumpUnmask, n, 0);
                    GR_CC_SWEEP(UpdateMask, n, 0);
                    continue;
                }
                row_form = false;  // a vertex outside the giant has a row too long for one lane: the edge form from here on
                hooks_pending = rows_hooked;
            }
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
            if (edge_stable && !hooks_pending) break;  // no edge hooked anything: done
            hooks_pending = false;
            for (;;) {
                GR_CC_SWEEP(PtrJumpMask, n, 0);
                if ((retval = poll())) return retval;
                if (vertex_stable) break;
            }
            GR_CC_SWEEP(PtrJumpUnmask, n, 0);
            GR_CC_SWEEP(UpdateMask, n, 0);
            if (row_form_wanted && hook_sweeps == 1) {  // the first edge-form round is over: does one component dominate now?
                if ((retval = pick_giant())) return retval;
                if ((retval = poll())) return retval;
                const long long samples = n >= kGiantSamples ? kGiantSamples : n;
                const long long held = static_cast<long long>(work_progress.h_tail[4] >> 32);
                row_form = held * 4 >= samples;
            }
        }
#undef GR_CC_SWEEP
        return retval;
    }
};

}  // namespace cc
}  // namespace app
}  // namespace gunrock
