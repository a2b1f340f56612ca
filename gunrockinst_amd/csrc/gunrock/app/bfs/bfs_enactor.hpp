// app/bfs/bfs_enactor.hpp -- host BSP loop for breadth-first search.
//
// Public contract of the reference's BFSEnactor (gunrock/app/bfs/bfs_enactor.cuh:40-708):
//   template <bool INSTRUMENT> class BFSEnactor : EnactorBase
//   Enact<BFSProblem>(context, problem, src, max_grid_size = 0, traversal_mode = 0)   (:573-579)
//   GetStatistics(total_queued, search_depth, avg_duty)                               (:173-186)
// The reference loop runs advance + filter per level with three blocking 4-byte D2H reads, an event
// sync, 5-6 launches and 2 moderngpu calls (bfs_enactor.cuh:267-531), and cudaMallocs an
// E*4-byte scan buffer inside every timed Enact (:254-259).  Here one level = ONE kernel launch (the
// advance discovers, labels and enqueues; its FrontierWriter emits the next level's degree prefix) and
// ONE 8-byte read of the packed tail, which tells the host both the next frontier length and its edge
// count.  No allocation happens inside Enact.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdlib>

#include <gunrock/app/bfs/bfs_functor.hpp>
#include <gunrock/app/bfs/bfs_problem.hpp>
#include <gunrock/app/enactor_base.hpp>
#include <gunrock/oprtr/advance/bottom_up.hpp>
#include <gunrock/oprtr/advance/kernel.hpp>
#include <gunrock/oprtr/advance/twc.hpp>
#include <gunrock/util/context.hpp>

namespace gunrock {
namespace app {
namespace bfs {

template <bool INSTRUMENT>
class BFSEnactor : public EnactorBase {
   public:
    explicit BFSEnactor(bool DEBUG = false) : EnactorBase(VERTEX_FRONTIERS, DEBUG) {}
    ~BFSEnactor() override {}

    // INSTRUMENT extras: operator launches and their summed device time in the last Enact
    void GetKernelStatistics(long long &launches, double &kernel_ms)
    {
        launches = enactor_stats.kernel_launches;
        kernel_ms = enactor_stats.kernel_ms;
    }

    const std::vector<EnactorStats::LevelRecord> &GetLevelTrace() const { return enactor_stats.levels; }

    void GetStatistics(long long &total_queued, long long &search_depth, double &avg_duty)
    {
        total_queued = enactor_stats.total_queued;
        search_depth = enactor_stats.iteration;
        avg_duty = (enactor_stats.total_lifetimes > 0) ? enactor_stats.total_runtimes / enactor_stats.total_lifetimes : 0.0;
    }
    template <typename T>
    void GetStatistics(long long &total_queued, T &search_depth, double &avg_duty)
    {
        long long depth;
        GetStatistics(total_queued, depth, avg_duty);
        search_depth = static_cast<T>(depth);
    }

    // Load-balanced advance policy: 256 threads x 8 slots = 2048 edge slots per tile, 37 KB LDS,
    // 4 workgroups (16 waves) per CU.
#ifndef GRX_LB_ITEMS
#define GRX_LB_ITEMS 4
#endif
    typedef oprtr::advance::KernelPolicy<256, GRX_LB_ITEMS, 8, oprtr::advance::LB> LBAdvancePolicy;
    // Phase 1 of a binned level: 8 edge slots per thread in flight (a tile's column loads, status probes and pair stores are
    // dependent round trips of a few microseconds each under load: what hides them is bytes in flight per CU).
#ifndef GRX_BINNED_ITEMS
#define GRX_BINNED_ITEMS 8
#endif
    typedef oprtr::advance::KernelPolicy<256, GRX_BINNED_ITEMS, 8, oprtr::advance::LB> BinnedPolicy;
    // Multi-level tail: one 1024-thread workgroup keeps expanding levels while a level has at most this many edges.
    typedef oprtr::advance::KernelPolicy<1024, 4, 1, oprtr::advance::LB> TailPolicy;
    // Persistent mid-size levels: 1024-thread workgroups (one edge slot per thread: the level is latency-bound, so spread it
    // wide), at most one per CU, grid barrier between levels.
    typedef oprtr::advance::KernelPolicy<1024, 1, 1, oprtr::advance::LB> PersistentPolicy;
    static constexpr int kTailEdgeLimit = 8192;  // default of BFSProblem::tail_edge_limit
    static constexpr int kTailMaxLevels = 4096;

    // traversal_mode: 0 = load-balanced top-down advance (reference default, bfs_enactor.cuh:581-697);
    //                 1 = the reference's TWC choice for low-degree graphs: TWC workgroup for small frontiers
    //                     (oprtr/advance/twc.hpp), persistent mid-size levels, LB advance for the rest;
    //                 2 = direction-optimizing (reference app/dobfs): needs BFSProblem::SetInverseGraph.
    template <typename BFSProblem>
    hipError_t Enact(util::DeviceContext & /*context*/, BFSProblem *problem, typename BFSProblem::VertexId src,
                     int max_grid_size = 0, int traversal_mode = 0)
    {
        // The reference's driver picks mode 1 (TWC) for graphs of average degree <= 8 (tests/bfs/test_bfs.cu:563-566): long
        // runs of small levels.  Here that case is served by the TWC workgroup and the persistent levels kernel, enabled by
        // mode 1 or by the same average-degree rule; on scale-free graphs a mid-size level is a one-level transition and a plain launch wins.
        const bool low_degree = static_cast<long long>(problem->edges) <= 8ll * problem->nodes;
        return EnactBFS<LBAdvancePolicy, BFSProblem>(problem, src, max_grid_size,
                                                     traversal_mode == 2 && problem->direction_optimizing,
                                                     traversal_mode == 1 || low_degree);
    }

   protected:
    // Launch the multi-level tail kernel on queue[selector] (its length lives in ring slot iteration & 3), wait, and
    // bring the host's view (iteration, selector, frontier size, statistics) up to date.
    // `persistent`: use the multi-workgroup kernel with a grid barrier (mid-size frontiers) instead of the single workgroup;
    // `switch_factor` > 0 makes it hand back as soon as a level's edges * factor exceed the unexplored edges (the host then
    // applies the direction-optimizing rules).
    template <typename BFSProblem, typename BfsFunctor>
    hipError_t RunTail(BFSProblem *problem, long long &iteration, int &selector, unsigned &queue_length, unsigned &queue_edges,
                       long long &unexplored_edges, hipStream_t stream, bool persistent = false, double switch_factor = 0.0,
                       bool frontier_size_unknown = false, bool twc = false, bool closing = false)
    {
        typedef typename BFSProblem::VertexId VertexId;
        typedef typename BFSProblem::SizeT SizeT;
        typedef typename BFSProblem::Value Value;
        hipError_t retval = hipSuccess;
        GraphSlice<VertexId, SizeT, Value> *gs = problem->graph_slices[0];
        oprtr::advance::TailArgs<VertexId, SizeT> t;
        t.queue[0] = gs->frontier_queues[0];
        t.queue[1] = gs->frontier_queues[1];
        t.selector = selector;
        t.first_iteration = iteration;
        t.d_tail = work_progress.d_tail;
        t.edge_limit = problem->tail_edge_limit;
        t.max_levels = kTailMaxLevels;
        t.d_levels_done = work_progress.LevelsDone();
        t.d_level_sums = work_progress.d_sums;
        t.d_row_offsets = gs->d_row_offsets;
        t.d_column_indices = gs->d_column_indices;
        t.d_overflow = work_progress.d_overflow;
        if (twc) {
            // thread / wave / workgroup tiers in one resident workgroup, frontier in LDS (oprtr/advance/twc.hpp)
            t.edge_limit = problem->twc_edge_limit;
            if ((retval = oprtr::advance::LaunchTwcLevels<BFSProblem, BfsFunctor>(t, *problem->data_slices[0], stream))) return retval;
        } else if (persistent) {
            oprtr::advance::PersistentArgs<VertexId, SizeT> p;
            p.t = t;
            // Grid barrier cost grows with the number of workgroups (2048^2 grid graph, ~16 K edges per level: 16 us per
            // level with 32 workgroups, 23 us with 256), so size the grid to the entry frontier (one tile per workgroup,
            // at least 16) and hand back to the host when the frontier outgrows it 8-fold; the host relaunches larger.
            long long grid = (static_cast<long long>(queue_edges) + PersistentPolicy::TILE - 1) / PersistentPolicy::TILE;
            if (grid < 16) grid = 16;
            if (grid > cu_count || frontier_size_unknown) grid = cu_count;  // (unknown: a kernel upstream wrote the frontier)
            long long limit = grid * PersistentPolicy::TILE * 8;
            if (grid == cu_count || limit > problem->persistent_edge_limit) limit = problem->persistent_edge_limit;
            if (frontier_size_unknown) limit *= 4;  // all CUs are in: a larger level is still cheaper here than a round trip
            p.t.edge_limit = static_cast<SizeT>(limit);
            p.barrier.d_counter = work_progress.BarrierCounter();
            p.barrier.d_timeout = work_progress.BarrierTimeout();
            p.solo_edges = problem->tail_edge_limit / 4;  // really small levels: workgroup 0 alone, no grid barrier
            p.unexplored_edges = unexplored_edges;
            p.switch_factor = switch_factor;
            if ((retval = oprtr::advance::LaunchPersistentLevels<PersistentPolicy, BFSProblem, BfsFunctor>(
                     p, *problem->data_slices[0], cu_count, static_cast<int>(grid), stream, problem->cooperative_launch)))
                return retval;
        } else if ((retval = oprtr::advance::LaunchTailLevels<TailPolicy, BFSProblem, BfsFunctor>(t, *problem->data_slices[0],
                                                                                                   stream)))
            return retval;
        // closing = this launch usually ends the search (the levels after the return to top-down): the deferred labels' emit pass
        // is queued behind it without waiting for the read-back that confirms it -- one host round trip less per search
        if (closing && (retval = problem->EmitLabelsSpeculative(stream))) return retval;
        if (INSTRUMENT && (retval = InstrumentEnd(stream))) return retval;
        if ((retval = work_progress.GetAll(stream))) return retval;
        if (persistent && work_progress.HostBarrierTimedOut())
            return util::GRError(hipErrorLaunchTimeOut, "BFSEnactor persistent levels kernel: grid barrier timed out", __FILE__,
                                 __LINE__);
        const int done = work_progress.HostLevelsDone();
        iteration += done;
        selector ^= (done & 1);
        queue_length = util::TailCount(work_progress.h_tail[iteration & 3]);
        queue_edges = util::TailEdges(work_progress.h_tail[iteration & 3]);
        if (queue_length > 0) problem->emit_current = false;  // the search goes on: whatever it finds now needs another pass
        // the loop head already counted the first level's frontier; add the rest
        enactor_stats.total_queued += static_cast<long long>(work_progress.h_sums[0]);
        enactor_stats.total_edges_queued += static_cast<long long>(work_progress.h_sums[1]);
        unexplored_edges -= static_cast<long long>(work_progress.h_sums[1]);
        return retval;
    }

    template <typename AdvancePolicy, typename BFSProblem>
    hipError_t EnactBFS(BFSProblem *problem, typename BFSProblem::VertexId src, int max_grid_size, bool dobfs,
                        bool persistent_levels)
    {
        typedef typename BFSProblem::VertexId VertexId;
        typedef typename BFSProblem::SizeT SizeT;
        typedef typename BFSProblem::Value Value;
        typedef BFSFunctor<VertexId, SizeT, Value, BFSProblem> BfsFunctor;
        constexpr int BU_THREADS = 256;

        hipError_t retval = hipSuccess;
        if ((retval = EnactorBase::Setup(max_grid_size, AdvancePolicy::MIN_BLOCKS, 8))) return retval;

        GraphSlice<VertexId, SizeT, Value> *gs = problem->graph_slices[0];
        typename BFSProblem::DataSlice *ds = problem->data_slices[0];
        hipStream_t stream = gs->stream;
        if (src < 0 || src >= problem->nodes) return problem->EmitLabels(stream);
        // deferred labels (bfs_problem.hpp): vertex-ordered sweeps keep their output bitmap instead of storing labels
        const bool deferring = dobfs && problem->labels_deferred;
        ds->defer_labels = deferring ? 1 : 0;

        unsigned queue_length = problem->SourceDegree() > 0 ? 1u : 0u;
        unsigned queue_edges = static_cast<unsigned>(problem->SourceDegree());
        if (problem->armed_progress == &work_progress) {  // Reset armed this enactor's words and seeded the source (bfs_problem.hpp)
            work_progress.box->overflow = 0;
            problem->armed_progress = nullptr;
        } else if ((retval = work_progress.ResetWithTail(0, queue_length, queue_edges, stream)))
            return retval;
        problem->arm_progress = &work_progress;  // (from the next Reset on)
        if (INSTRUMENT && (retval = DutyBegin(stream))) return retval;

#ifndef GRX_CONV_GRID_MULT
#define GRX_CONV_GRID_MULT 2
#endif
        const int conv_grid = cu_count * GRX_CONV_GRID_MULT;
        const size_t mask_bytes = sizeof(unsigned) * static_cast<size_t>(problem->MaskWords() + 2);
        long long unexplored_edges = problem->edges;
        // (direction-optimizing: BFSProblem::Reset left "visited before the search" in d_snapshot)
        bool bottom_up = false;  // direction of the frontier representation: queue (false) or bitmap (true)
        bool force_bottom_up = false; // the last level ran count-only: its output exists only as a bitmap (already built)
        bool queue_emitted = false;   // the last (compacting) bottom-up sweep also wrote its finds to queue[selector] + ring slot
        bool snapshot_valid = true;  // d_snapshot holds "visited before the last top-down level" (Reset seeds it)
        bool out_slot_clean = false; // the last bottom-up read-back cleared the ring slot a bitmap -> queue conversion would fill
        int cur_mask = 0;            // index into the pool of frontier bitmaps (BFSProblem::AcquireMask)
        int selector = 0;
        long long iteration = 0;
        // bottom-up sweep: frontier = d_frontier_mask[in_mask], finds -> d_frontier_mask[out_mask]; heads_only = probe the
        // adjacency heads and stop
        auto launch_bottom_up = [&](int in_mask, int out_mask, int heads_only) -> hipError_t {
            queue_emitted = false;
            oprtr::advance::BottomUpArgs<VertexId, SizeT> bargs;
            bargs.nodes = problem->nodes;
            bargs.d_inv_row_offsets = ds->d_inv_row_offsets;
            bargs.d_inv_column_indices = ds->d_inv_column_indices;
            bargs.d_inv_heads = ds->d_inv_heads;
            oprtr::advance::BitmapLookup<VertexId> lookup{ds->d_frontier_mask[in_mask]};
            bargs.d_frontier_out = reinterpret_cast<unsigned long long *>(ds->d_frontier_mask[out_mask]);
            bargs.d_visited = reinterpret_cast<unsigned long long *>(ds->d_visited_mask);
            bargs.d_tail_out = work_progress.d_tail + ((iteration + 1) & 3);
            bargs.d_tail_clear = work_progress.d_tail + ((iteration + 2) & 3);
            bargs.d_wide = work_progress.d_wide;
            bargs.head_skip = 0;  // BFSProblem ranks the heads by degree: the row walk starts at its first entry
            bargs.heads_only = heads_only;
            bargs.d_never = reinterpret_cast<const unsigned long long *>(ds->d_never_mask);
            bargs.d_head_base = ds->d_head_base;
            bargs.d_duty = INSTRUMENT ? DutySlot() : nullptr;
            const long long bu_steps = ((static_cast<long long>(problem->nodes) + 63) / 64 + oprtr::advance::kBottomUpStepWords - 1) / oprtr::advance::kBottomUpStepWords;
            long long grid = (bu_steps + (BU_THREADS / 64) - 1) / (BU_THREADS / 64);
            // nearly finished search: few vertices can still be unvisited -> the compacting sweep (bottom_up.hpp)
            const long long open_estimate = problem->with_in_edges - enactor_stats.total_queued;
            if (problem->sparse_sweep_div > 0 && open_estimate * problem->sparse_sweep_div <= static_cast<long long>(problem->nodes)) {
                const long long chunks = ((static_cast<long long>(problem->nodes) + 63) / 64 + oprtr::advance::kSparseChunkWords - 1) / oprtr::advance::kSparseChunkWords;
                long long sgrid = (chunks + (BU_THREADS / 64) - 1) / (BU_THREADS / 64);
                if (sgrid < 1) sgrid = 1;
                typedef oprtr::advance::BitmapLookup<VertexId> L;
                // Emitting costs a flush (row extents, scan, writes) per workgroup, so only when this level will probably be the
                // last bottom-up one: its input frontier is already within 32x of the switch-back threshold.
                if (!heads_only && static_cast<double>(queue_length) * problem->beta <
                                       static_cast<double>(problem->emit_queue_factor) * static_cast<double>(problem->nodes)) {
                    // also emit the finds as queue[selector] (tail in this level's output ring slot): a switch back to
                    // top-down then starts from it directly.  Fewer workgroups: each ends with one packed atomic on the slot.
                    bargs.queue_out = gs->frontier_queues[selector];
                    bargs.d_fwd_row_offsets = gs->d_row_offsets;
                    bargs.d_queue_tail = work_progress.d_tail + ((iteration + 1) & 3);
                    bargs.d_queue_invalid = reinterpret_cast<int *>(work_progress.AuxTail());
                    bargs.d_overflow = work_progress.d_overflow;
                    long long scap = max_grid_size > 0 ? max_grid_size
                        : util::ResidentGrid(oprtr::advance::BottomUpSparseKernel<BU_THREADS, 8, 32, BFSProblem, L, true>, BU_THREADS);
                    if (scap > cu_count * 4) scap = cu_count * 4;
                    if (sgrid > scap) sgrid = scap;
                    hipLaunchKernelGGL((oprtr::advance::BottomUpSparseKernel<BU_THREADS, 8, 32, BFSProblem, L, true>),
                                       dim3(static_cast<unsigned>(sgrid)), dim3(BU_THREADS), 0, stream, bargs, *ds, lookup);
                    queue_emitted = true;
                    return util::GRError("BottomUpSparseKernel launch failed", __FILE__, __LINE__);
                }
                const long long scap = max_grid_size > 0 ? max_grid_size
                    : util::ResidentGrid(oprtr::advance::BottomUpSparseKernel<BU_THREADS, 8, 32, BFSProblem, L, false>, BU_THREADS);
                if (sgrid > scap) sgrid = scap;
                hipLaunchKernelGGL((oprtr::advance::BottomUpSparseKernel<BU_THREADS, 8, 32, BFSProblem, L, false>),
                                   dim3(static_cast<unsigned>(sgrid)), dim3(BU_THREADS), 0, stream, bargs, *ds, lookup);
                return util::GRError("BottomUpSparseKernel launch failed", __FILE__, __LINE__);
            }
            if (heads_only) {  // the lean instantiation (no row walks: fewer registers, more waves)
                typedef oprtr::advance::BitmapLookup<VertexId> L;
                const long long hcap = max_grid_size > 0 ? max_grid_size : util::ResidentGrid(oprtr::advance::BottomUpHeadsKernel<BU_THREADS, BFSProblem, L>, BU_THREADS);
                if (grid > hcap) grid = hcap;
                if (grid < 1) grid = 1;
                hipLaunchKernelGGL((oprtr::advance::BottomUpHeadsKernel<BU_THREADS, BFSProblem, L>), dim3(static_cast<unsigned>(grid)), dim3(BU_THREADS), 0,
                                   stream, bargs, *ds, lookup);
                return util::GRError("BottomUpHeadsKernel launch failed", __FILE__, __LINE__);
            }
            const long long cap = max_grid_size > 0 ? max_grid_size
                : util::ResidentGrid(oprtr::advance::BottomUpKernel<BU_THREADS, 8, 32, BFSProblem, oprtr::advance::BitmapLookup<VertexId>>, BU_THREADS);
            if (grid > cap) grid = cap;
            if (grid < 1) grid = 1;
            hipLaunchKernelGGL((oprtr::advance::BottomUpKernel<BU_THREADS, 8, 32, BFSProblem, oprtr::advance::BitmapLookup<VertexId>>),
                               dim3(static_cast<unsigned>(grid)), dim3(BU_THREADS), 0, stream, bargs, *ds, lookup);
            return util::GRError("BottomUpKernel launch failed", __FILE__, __LINE__);
        };
        // ---- a CHAIN of bottom-up sweeps behind one host round trip (bottom_up.hpp BottomUpAutoKernel) ----
        // The frontier is d_frontier_mask[cur_mask]; its size is `first_in`, or -- when the host has not read it back (the chain
        // follows a count-only level directly) -- the sum of wide set 0.  Every queued sweep decides on the device whether it runs
        // dense, compacting (with or without an emitted queue) or not at all; afterwards the host replays the same rules on the
        // published sums to learn how many ran, which bitmaps they filled and what the next frontier is.
        const int chain_len = (dobfs && problem->chain_sweeps > 0)
            ? (problem->chain_sweeps < oprtr::advance::kChainMax ? problem->chain_sweeps : oprtr::advance::kChainMax) : 0;
        work_progress.publish_sets = chain_len > 0 ? chain_len + 1 : 1;
        oprtr::advance::SweepRule sweep_rule;
        sweep_rule.with_in_edges = problem->with_in_edges;
        sweep_rule.nodes = problem->nodes;
        sweep_rule.beta = problem->beta;
        sweep_rule.emit_factor = problem->emit_queue_factor;
        sweep_rule.sparse_div = problem->sparse_sweep_div;
        auto run_sweep_chain = [&](long long first_in) -> hipError_t {
            typedef oprtr::advance::BitmapLookup<VertexId> L;
            hipError_t rc = hipSuccess;
            const long long it0 = iteration;
            const long long base_total = enactor_stats.total_queued;
            int masks[oprtr::advance::kChainMax + 1];
            masks[0] = cur_mask;
            int queued = 0;
            const long long bu_steps = ((static_cast<long long>(problem->nodes) + 63) / 64 + oprtr::advance::kBottomUpStepWords - 1) / oprtr::advance::kBottomUpStepWords;
            long long grid = (bu_steps + (BU_THREADS / 64) - 1) / (BU_THREADS / 64);
            const long long cap = max_grid_size > 0 ? max_grid_size
                : util::ResidentGrid(oprtr::advance::BottomUpAutoKernel<BU_THREADS, 8, 32, BFSProblem, L>, BU_THREADS);
            if (grid > cap) grid = cap;
            if (grid < 1) grid = 1;
            const long long chunks = ((static_cast<long long>(problem->nodes) + 63) / 64 + oprtr::advance::kSparseChunkWords - 1) / oprtr::advance::kSparseChunkWords;
            long long sparse_grid = (chunks + (BU_THREADS / 64) - 1) / (BU_THREADS / 64);
            if (sparse_grid < 1) sparse_grid = 1;
            long long emit_grid = sparse_grid < cu_count * 4 ? sparse_grid : cu_count * 4;  // (each workgroup ends with an atomic on one word)
            for (int k = 1; k <= chain_len; ++k) {
                bool got = false;
                if ((rc = problem->TryAcquireMask(stream, masks[k], masks, k, got))) return rc;
                if (!got) break;  // the pool is full of bitmaps this chain still reads: a shorter chain
                oprtr::advance::BottomUpArgs<VertexId, SizeT> bargs;
                bargs.nodes = problem->nodes;
                bargs.d_inv_row_offsets = ds->d_inv_row_offsets;
                bargs.d_inv_column_indices = ds->d_inv_column_indices;
                bargs.d_inv_heads = ds->d_inv_heads;
                L lookup{ds->d_frontier_mask[masks[k - 1]]};
                bargs.d_frontier_out = reinterpret_cast<unsigned long long *>(ds->d_frontier_mask[masks[k]]);
                bargs.d_visited = reinterpret_cast<unsigned long long *>(ds->d_visited_mask);
                const long long level = it0 + k - 1;  // the BSP iteration this sweep is
                bargs.d_tail_out = work_progress.d_tail + ((level + 1) & 3);
                bargs.d_tail_clear = work_progress.d_tail + ((level + 2) & 3);
                bargs.d_wide = work_progress.d_wide + static_cast<size_t>(k) * util::WorkProgress::kWideSetWords;
                bargs.head_skip = 0;
                bargs.heads_only = 0;
                bargs.d_never = reinterpret_cast<const unsigned long long *>(ds->d_never_mask);
                bargs.d_head_base = ds->d_head_base;
                bargs.d_duty = INSTRUMENT ? DutySlot() : nullptr;
                bargs.queue_out = gs->frontier_queues[selector];
                bargs.d_fwd_row_offsets = gs->d_row_offsets;
                bargs.d_queue_tail = work_progress.d_tail + ((level + 1) & 3);
                bargs.d_queue_invalid = reinterpret_cast<int *>(work_progress.AuxTail());
                bargs.d_overflow = work_progress.d_overflow;
                oprtr::advance::SweepChain chain;
                chain.d_sets = work_progress.d_wide;
                chain.base_total = base_total;
                chain.first_in = first_in;
                chain.rule = sweep_rule;
                chain.index = k;
                chain.first_may_switch = 0;  // (the host has just applied the switch-back rule to the first frontier, or has just turned bottom-up)
                chain.d_log = work_progress.d_chain_log;
                typename BFSProblem::DataSlice level_slice = *ds;
                level_slice.iteration = static_cast<VertexId>(level);
                hipLaunchKernelGGL((oprtr::advance::BottomUpAutoKernel<BU_THREADS, 8, 32, BFSProblem, L>), dim3(static_cast<unsigned>(grid)),
                                   dim3(BU_THREADS), 0, stream, bargs, level_slice, lookup, chain, static_cast<unsigned>(sparse_grid),
                                   static_cast<unsigned>(emit_grid));
                if ((rc = util::GRError("BottomUpAutoKernel launch failed", __FILE__, __LINE__))) return rc;
                queued = k;
            }
            if (queued == 0) return util::GRError(hipErrorInvalidValue, "BFSEnactor: no frontier bitmap for a bottom-up sweep", __FILE__, __LINE__);
            // ---- the closing levels behind the chain, still without a round trip (kernel.hpp ChainedPersistentLevelsKernel): they
            //      run when the chain ends with "back to top-down" right after a sweep that emitted its finds as a queue -- what a
            //      scale-free search nearly always does -- and the label pass behind them reads from the gate words how many of the
            //      chain's bitmaps were filled.  Otherwise both exit at once and the host carries on as before. ----
            const bool closing_queued = problem->chain_closing && problem->persistent_edge_limit > 0;
            int *d_gate = work_progress.d_chain_log + 8;
            bool emit_queued = false;
            if (closing_queued) {
                oprtr::advance::PersistentArgs<VertexId, SizeT> p;
                p.t.queue[0] = gs->frontier_queues[0];
                p.t.queue[1] = gs->frontier_queues[1];
                p.t.selector = selector;
                p.t.first_iteration = it0;  // (the kernel adds the sweeps that ran)
                p.t.d_tail = work_progress.d_tail;
                p.t.max_levels = kTailMaxLevels;
                p.t.d_levels_done = work_progress.LevelsDone();
                p.t.d_level_sums = work_progress.d_sums;
                p.t.d_row_offsets = gs->d_row_offsets;
                p.t.d_column_indices = gs->d_column_indices;
                p.t.d_overflow = work_progress.d_overflow;
                p.t.edge_limit = static_cast<SizeT>(static_cast<long long>(problem->persistent_edge_limit) * 4);  // (as RunTail's closing launch)
                p.barrier.d_counter = work_progress.BarrierCounter();
                p.barrier.d_timeout = work_progress.BarrierTimeout();
                p.solo_edges = problem->tail_edge_limit / 4;
                p.unexplored_edges = unexplored_edges;
                p.switch_factor = 0.0;
                oprtr::advance::SweepChain chain;
                chain.d_sets = work_progress.d_wide;
                chain.base_total = base_total;
                chain.first_in = first_in;
                chain.rule = sweep_rule;
                chain.index = queued;
                chain.first_may_switch = 0;
                chain.d_log = work_progress.d_chain_log;
                if ((rc = oprtr::advance::LaunchChainedPersistentLevels<PersistentPolicy, BFSProblem, BfsFunctor>(
                         p, *ds, cu_count, chain, queued, it0, reinterpret_cast<const int *>(work_progress.AuxTail()), d_gate, stream)))
                    return rc;
                if (deferring && problem->speculative_emit && problem->level_masks.count + queued <= app::bfs::kLevelMasks) {
                    VertexId labels[oprtr::advance::kChainMax];
                    for (int k = 1; k <= queued; ++k) labels[k - 1] = static_cast<VertexId>(it0 + k);
                    if ((rc = problem->EmitLabelsGated(stream, masks + 1, labels, queued, d_gate))) return rc;
                    emit_queued = true;
                }
            }
            if (INSTRUMENT && (rc = InstrumentEnd(stream))) return rc;
            if ((rc = work_progress.Sync(stream))) return rc;
            // ---- replay (the same SweepRule on the same numbers) ----
            long long total = base_total;
            int ran = 0, last_action = oprtr::advance::kSweepStop;
            long long next_in = 0;
            for (int k = 1; k <= queued + 1; ++k) {
                const long long in_k = (k == 1 && first_in >= 0) ? first_in
                    : static_cast<long long>(util::TailCount(k == 1 ? work_progress.box->wide : work_progress.box->wide_set[k - 1]));
                next_in = in_k;
                if (k > queued) break;  // the frontier the last queued sweep produced: the host loop takes it from here
                const int action = sweep_rule.Decide(in_k, total + in_k, k > 1);
                if (action != work_progress.box->chain_log[k])
                    return util::GRError(hipErrorUnknown, "BFSEnactor: host and device disagree about a chained sweep", __FILE__, __LINE__);
                if (action <= oprtr::advance::kSweepSwitch) break;
                total += in_k;
                ran = k;
                last_action = action;
                if (deferring) problem->KeepMask(masks[k], static_cast<VertexId>(it0 + k));
            }
            enactor_stats.total_queued = total;
            iteration = it0 + ran;
            cur_mask = masks[ran];
            queue_length = static_cast<unsigned>(next_in);
            queue_edges = 0;
            queue_emitted = ran > 0 && last_action == oprtr::advance::kSweepSparseEmit;
            out_slot_clean = ran > 0 && !queue_emitted;  // (a sweep that emits nothing zeroes its queue-tail slot itself)
            if (queue_emitted && work_progress.h_tail[util::WorkProgress::kAux] != 0) {  // a staging buffer overflowed: no usable queue
                queue_emitted = false;
                if ((rc = work_progress.ClearAux(stream))) return rc;
            }
            if (INSTRUMENT) InstrumentCollect(static_cast<long long>(first_in >= 0 ? first_in : 0), 0, 1);
            // ---- did the closing levels run?  (the kernel's own test, replayed) ----
            if (closing_queued && ran >= 1 && queue_emitted && queue_length > 0 &&
                sweep_rule.Decide(next_in, total + next_in, true) == oprtr::advance::kSweepSwitch) {
                if (work_progress.HostBarrierTimedOut())
                    return util::GRError(hipErrorLaunchTimeOut, "BFSEnactor persistent levels kernel: grid barrier timed out", __FILE__, __LINE__);
                const int done = work_progress.HostLevelsDone();
                iteration += done;
                selector ^= (done & 1);
                queue_length = util::TailCount(work_progress.h_tail[iteration & 3]);
                queue_edges = util::TailEdges(work_progress.h_tail[iteration & 3]);
                enactor_stats.total_queued += static_cast<long long>(work_progress.h_sums[0]);
                enactor_stats.total_edges_queued += static_cast<long long>(work_progress.h_sums[1]);
                unexplored_edges -= static_cast<long long>(work_progress.h_sums[1]);
                bottom_up = false;
                queue_emitted = false;
                out_slot_clean = false;
                snapshot_valid = false;
                // (the gated label pass ran behind them; it stands unless the search goes on)
                if (emit_queued) problem->emit_current = queue_length == 0;
            }
            return rc;
        };
        // count-only top-down advance over the current queue: unvisited destinations get their d_fresh byte set
        auto launch_count_only = [&](oprtr::advance::AdvanceArgs<VertexId, SizeT> args) -> hipError_t {
            ds->lite = 1;
            if (INSTRUMENT && !args.d_duty) args.d_duty = DutySlot();
            args.d_tail_out = nullptr;  // the advance's own count includes duplicates: not wanted
            hipError_t rc = oprtr::advance::LaunchKernel<AdvancePolicy, BFSProblem, BfsFunctor, true, true>(args, *ds, max_grid_size, stream,
                                                                                                              oprtr::advance::V2V);
            ds->lite = 0;
            return rc;
        };
        // closing pass of a count-only level: bytes -> d_frontier_mask[out_mask] (| d_merge), visited, labels (or the bitmap is kept
        // for the deferred labels), count (wide tail)
        auto launch_fresh_pass = [&](const unsigned long long *d_before, const unsigned long long *d_merge, int out_mask) -> hipError_t {
            const long long words64 = (static_cast<long long>(problem->nodes) + 63) / 64;
            long long fgrid = ((words64 + 15) / 16 + 3) / 4;  // 16 words per wave step, 4 waves per workgroup
            if (fgrid > cu_count * 4) fgrid = cu_count * 4;
            hipLaunchKernelGGL((oprtr::advance::FreshToBitmapKernel<VertexId>), dim3(static_cast<unsigned>(fgrid)), dim3(256), 0, stream,
                               ds->d_fresh, static_cast<long long>(problem->nodes), reinterpret_cast<unsigned long long *>(ds->d_visited_mask),
                               d_before, reinterpret_cast<unsigned long long *>(ds->d_frontier_mask[out_mask]),
                               deferring ? static_cast<VertexId *>(nullptr) : ds->d_labels, static_cast<VertexId>(iteration + 1), work_progress.d_tail + ((iteration + 1) & 3), work_progress.d_wide,
                               d_merge, work_progress.d_tail + ((iteration + 2) & 3));
            if (deferring) problem->KeepMask(out_mask, static_cast<VertexId>(iteration + 1));
            return util::GRError("FreshToBitmapKernel launch failed", __FILE__, __LINE__);
        };
        // frontier queue -> d_frontier_mask[out_mask]
        auto queue_to_mask = [&](int out_mask) -> hipError_t {
            hipError_t rc;
            if (!snapshot_valid) {  // after a multi-level kernel the snapshot is several levels old: rebuild from the queue
                if ((rc = util::GRError(hipMemsetAsync(ds->d_frontier_mask[out_mask], 0, mask_bytes, stream),
                                        "BFSEnactor hipMemsetAsync frontier mask failed", __FILE__, __LINE__)))
                    return rc;
                hipLaunchKernelGGL((oprtr::advance::QueueToBitmapKernel<VertexId, SizeT>), dim3(conv_grid), dim3(256), 0, stream,
                                   gs->frontier_queues[selector].v, static_cast<SizeT>(queue_length), ds->d_frontier_mask[out_mask]);
                return util::GRError("QueueToBitmapKernel launch failed", __FILE__, __LINE__);
            }
            // the frontier is exactly what the last top-down level added to the visited bitmap (zero-degree discoveries
            // included: they have no out-edges, so nobody can adopt them as parent)
            const long long words64 = static_cast<long long>(problem->MaskWords()) / 2;
            hipLaunchKernelGGL(oprtr::advance::BitmapDiffKernel, dim3(conv_grid), dim3(256), 0, stream,
                               reinterpret_cast<const unsigned long long *>(ds->d_visited_mask),
                               reinterpret_cast<const unsigned long long *>(ds->d_snapshot),
                               reinterpret_cast<unsigned long long *>(ds->d_frontier_mask[out_mask]), words64);
            return util::GRError("BitmapDiffKernel launch failed", __FILE__, __LINE__);
        };
        while (queue_length > 0) {
            const unsigned in_len = queue_length, in_edges = queue_edges;
            bool binned_level = false;
            if (INSTRUMENT && (retval = InstrumentBegin(stream))) break;

            // ---- direction choice (Beamer's edge rule for down->up, the reference's vertex rule for up->down,
            //      dobfs_enactor.cuh:397,569) ----
            if (dobfs && !bottom_up && !force_bottom_up && problem->head_pass_min_edges != 0 &&
                static_cast<long long>(queue_edges) >= problem->HeadPassMin() &&
                static_cast<long long>(queue_edges) <= problem->HeadPassMax() &&
                queue_edges > static_cast<unsigned>(problem->tail_edge_limit) &&
                static_cast<double>(queue_edges) * problem->alpha * problem->lite_factor > static_cast<double>(unexplored_edges)) {
                // ---- "heads, then the rest": a large level out of a sparse frontier is slow in either direction (top-down pays
                //      one scattered store per edge, bottom-up one adjacency fetch per unvisited vertex).  First a bottom-up pass
                //      that probes ONLY the adjacency heads -- the highest-degree in-neighbours, i.e. the likely members of such a
                //      frontier: it settles most of the level's discoveries for the price of streaming the heads.  Then the
                //      count-only top-down advance: its status screen now rejects every edge into a vertex the heads found, so the
                //      scattered stores shrink to the remainder.
                enactor_stats.total_queued += queue_length;
                enactor_stats.total_edges_queued += queue_edges;
                unexplored_edges -= queue_edges;
                ds->iteration = static_cast<VertexId>(iteration);
                int front_mask = 0, heads_mask = 0, out_mask = 0;
                if ((retval = problem->AcquireMask(stream, front_mask))) break;
                if ((retval = queue_to_mask(front_mask))) break;                                 // frontier -> bitmap
                if ((retval = problem->AcquireMask(stream, heads_mask, front_mask))) break;
                if ((retval = launch_bottom_up(front_mask, heads_mask, 1))) break;               // the heads' finds
                oprtr::advance::AdvanceArgs<VertexId, SizeT> args;
                args.in = gs->frontier_queues[selector];
                args.out = gs->frontier_queues[selector ^ 1];
                args.in_len = static_cast<SizeT>(queue_length);
                args.in_edges = static_cast<SizeT>(queue_edges);
                args.d_row_offsets = gs->d_row_offsets;
                args.d_column_indices = gs->d_column_indices;
                args.d_tail_out = nullptr;
                args.d_tail_clear = nullptr;
                args.d_overflow = work_progress.d_overflow;
                if ((retval = launch_count_only(args))) break;
                // the advance did not touch the visited bitmap, so the bitmap itself is "visited before" for the closing pass
                if ((retval = problem->AcquireMask(stream, out_mask, front_mask, heads_mask))) break;
                if ((retval = launch_fresh_pass(reinterpret_cast<const unsigned long long *>(ds->d_visited_mask),
                                                reinterpret_cast<const unsigned long long *>(ds->d_frontier_mask[heads_mask]), out_mask)))
                    break;
                cur_mask = out_mask;
                snapshot_valid = false;
                ++iteration;
                if (chain_len > 0) {  // the bottom-up sweeps follow without a round trip: the level's size stays on the device
                    if (INSTRUMENT) {
                        if ((retval = InstrumentEnd(stream))) break;
                        if ((retval = util::GRError(hipStreamSynchronize(stream), "BFSEnactor instrument sync failed", __FILE__, __LINE__))) break;
                        InstrumentCollect(in_len, in_edges, 6);
                        if ((retval = InstrumentBegin(stream))) break;
                    }
                    bottom_up = true;
                    if ((retval = run_sweep_chain(-1))) break;
                    continue;
                }
                force_bottom_up = true;
                if (INSTRUMENT && (retval = InstrumentEnd(stream))) break;
                if ((retval = work_progress.GetTailWide(static_cast<int>(iteration & 3), queue_length, queue_edges, stream))) break;
                if (INSTRUMENT) InstrumentCollect(in_len, in_edges, 6);
                continue;  // (selector unchanged: no queue was written)
            } else if (dobfs && !bottom_up &&
                (force_bottom_up || static_cast<double>(queue_edges) * problem->alpha > static_cast<double>(unexplored_edges))) {
                if (force_bottom_up) {  // the count-only level already left its discoveries in d_frontier_mask[cur_mask]
                    force_bottom_up = false;
                } else {  // queue -> bitmap
                    if ((retval = problem->AcquireMask(stream, cur_mask))) break;
                    if ((retval = queue_to_mask(cur_mask))) break;
                }
                bottom_up = true;
            } else if (dobfs && bottom_up &&
                       static_cast<double>(queue_length) * problem->beta < static_cast<double>(problem->nodes)) {
                // bitmap -> queue (exact forward degrees; zero out-degree vertices are dropped), written straight into
                // this iteration's ring slot so the multi-level tail kernel can take over without a host round trip
                if (!queue_emitted) {
                    unsigned long long *slot = work_progress.d_tail + (iteration & 3);
                    // (normally the read-back of the last sweep has already zeroed this slot: a fill blit here cost ~10 us of host
                    //  and device time between two kernels)
                    if (!out_slot_clean && (retval = util::GRError(hipMemsetAsync(slot, 0, sizeof(unsigned long long), stream),
                                                                   "BFSEnactor clear tail failed", __FILE__, __LINE__)))
                        break;
                    hipLaunchKernelGGL((oprtr::advance::BitmapToQueueKernel<256, VertexId, SizeT>), dim3(conv_grid), dim3(256),
                                       0, stream, ds->d_frontier_mask[cur_mask], problem->nodes,
                                       gs->frontier_queues[selector], slot, work_progress.d_overflow, gs->d_row_offsets);
                    if ((retval = util::GRError("BitmapToQueueKernel launch failed", __FILE__, __LINE__))) break;
                }  // (else: the compacting sweep already wrote queue[selector] and this ring slot)
                queue_emitted = false;
                bottom_up = false;
                snapshot_valid = false;
                // The rest of the search usually only shrinks: one persistent launch takes the converted frontier (all CUs,
                // grid barrier) and carries on alone in workgroup 0 once the levels are small -- no host round trip between.
                if ((retval = RunTail<BFSProblem, BfsFunctor>(problem, iteration, selector, queue_length, queue_edges,
                                                              unexplored_edges, stream, problem->persistent_edge_limit > 0, 0.0,
                                                              true, false, true)))
                    break;
                if (INSTRUMENT) InstrumentCollect(in_len, in_edges, 2);
                continue;  // the loop re-examines the frontier the tail kernel left (empty, or too large for it)
            } else if (persistent_levels && !bottom_up && problem->twc_edge_limit > 0 &&
                       queue_length <= static_cast<unsigned>(oprtr::advance::kTwcCapacity) &&
                       queue_edges <= static_cast<unsigned>(problem->twc_edge_limit) &&
                       !(dobfs && static_cast<double>(queue_edges) * problem->alpha * problem->lite_factor > static_cast<double>(unexplored_edges))) {
                // low-degree graph (or traversal_mode 1), small frontier: the TWC tiers keep the frontier in LDS and run levels
                // until one outgrows it -- a road-like graph's whole search
                const long long before = iteration;
                snapshot_valid = false;
                if ((retval = RunTail<BFSProblem, BfsFunctor>(problem, iteration, selector, queue_length, queue_edges,
                                                              unexplored_edges, stream, false, 0.0, false, true)))
                    break;
                if (INSTRUMENT) InstrumentCollect(in_len, in_edges, 8);
                if (iteration != before) continue;
                if (INSTRUMENT && (retval = InstrumentBegin(stream))) break;  // no level ran: fall through to the grid kernels
            } else if (!bottom_up && queue_edges <= static_cast<unsigned>(problem->tail_edge_limit)) {
                // small top-down frontier: run as many levels as stay small inside one launch
                const long long before = iteration;
                snapshot_valid = false;
                if ((retval = RunTail<BFSProblem, BfsFunctor>(problem, iteration, selector, queue_length, queue_edges,
                                                              unexplored_edges, stream)))
                    break;
                if (INSTRUMENT) InstrumentCollect(in_len, in_edges, 3);
                if (iteration != before) continue;
                if (INSTRUMENT && (retval = InstrumentBegin(stream))) break;  // no level ran: fall through to the grid kernel
            } else if (persistent_levels && !bottom_up &&
                       queue_edges <= static_cast<unsigned>(problem->persistent_edge_limit) &&
                       !(dobfs && static_cast<double>(queue_edges) * problem->alpha * problem->lite_factor >
                                      static_cast<double>(unexplored_edges))) {
                // mid-size top-down frontier: resident workgroups run consecutive levels with a grid barrier in between
                const long long before = iteration;
                snapshot_valid = false;
                if ((retval = RunTail<BFSProblem, BfsFunctor>(problem, iteration, selector, queue_length, queue_edges,
                                                              unexplored_edges, stream, true,
                                                              dobfs ? problem->alpha * problem->lite_factor : 0.0)))
                    break;
                if (INSTRUMENT) InstrumentCollect(in_len, in_edges, 5);
                if (iteration != before) continue;
                if (INSTRUMENT && (retval = InstrumentBegin(stream))) break;
            }
            if (bottom_up && chain_len > 0) {  // bottom-up levels: several sweeps behind one round trip
                if ((retval = run_sweep_chain(static_cast<long long>(queue_length)))) break;
                continue;
            }
            enactor_stats.total_queued += queue_length;
            enactor_stats.total_edges_queued += queue_edges;
            unexplored_edges -= queue_edges;
            ds->iteration = static_cast<VertexId>(iteration);
            // "Count-only" top-down level: when the search is about to turn bottom-up, the level's discoveries are needed
            // only as a bitmap, so the frontier writer (row-offset gather, 12-byte queue entries), the claims and the scattered
            // label stores are skipped; a closing sweep turns the flag bytes into the bitmap.
            const bool lite = dobfs && !bottom_up && queue_edges > static_cast<unsigned>(problem->tail_edge_limit) &&
                              static_cast<double>(queue_edges) * problem->alpha * problem->lite_factor > static_cast<double>(unexplored_edges);
            if (dobfs && !bottom_up && !lite) {
                // snapshot of the visited bitmap before this top-down level (n/8 bytes): what a switch to bottom-up after it
                // diffs against.  Own kernel: hipMemcpyAsync's blit path cost the host ~8 us more per call.  (A count-only
                // level needs none: it does not touch the visited bitmap.)
                const long long words64 = static_cast<long long>(problem->MaskWords()) / 2 + 1;
                hipLaunchKernelGGL(oprtr::advance::BitmapCopyKernel, dim3(conv_grid), dim3(256), 0, stream,
                                   reinterpret_cast<const unsigned long long *>(ds->d_visited_mask),
                                   reinterpret_cast<unsigned long long *>(ds->d_snapshot), words64);
                if ((retval = util::GRError("BitmapCopyKernel launch failed", __FILE__, __LINE__))) break;
                snapshot_valid = true;
            }

            if (bottom_up) {
                int out_mask = 0;
                if ((retval = problem->AcquireMask(stream, out_mask, cur_mask))) break;
                if ((retval = launch_bottom_up(cur_mask, out_mask, 0))) break;
                if (deferring) problem->KeepMask(out_mask, static_cast<VertexId>(iteration + 1));
                cur_mask = out_mask;
            } else {
                oprtr::advance::AdvanceArgs<VertexId, SizeT> args;
                args.in = gs->frontier_queues[selector];
                args.out = gs->frontier_queues[selector ^ 1];
                args.in_len = static_cast<SizeT>(queue_length);
                args.in_edges = static_cast<SizeT>(queue_edges);
                args.d_row_offsets = gs->d_row_offsets;
                args.d_column_indices = gs->d_column_indices;
                args.d_tail_out = work_progress.d_tail + ((iteration + 1) & 3);
                args.d_tail_clear = work_progress.d_tail + ((iteration + 2) & 3);
                args.d_overflow = work_progress.d_overflow;
                args.d_duty = INSTRUMENT ? DutySlot() : nullptr;
                if (lite) {
                    int out_mask = 0;
                    if ((retval = problem->AcquireMask(stream, out_mask))) break;
                    if ((retval = launch_count_only(args))) break;
                    // (the advance did not touch the visited bitmap: the bitmap itself is "visited before the level")
                    if ((retval = launch_fresh_pass(reinterpret_cast<const unsigned long long *>(ds->d_visited_mask), nullptr, out_mask))) break;
                    cur_mask = out_mask;
                    snapshot_valid = false;
                    ++iteration;
                    if (chain_len > 0) {  // (as after the heads-then-rest level)
                        if (INSTRUMENT) {
                            if ((retval = InstrumentEnd(stream))) break;
                            if ((retval = util::GRError(hipStreamSynchronize(stream), "BFSEnactor instrument sync failed", __FILE__, __LINE__))) break;
                            InstrumentCollect(in_len, in_edges, 4);
                            if ((retval = InstrumentBegin(stream))) break;
                        }
                        bottom_up = true;
                        if ((retval = run_sweep_chain(-1))) break;
                        continue;
                    }
                    force_bottom_up = true;
                    if (INSTRUMENT && (retval = InstrumentEnd(stream))) break;
                    if ((retval = work_progress.GetTailWide(static_cast<int>(iteration & 3), queue_length, queue_edges, stream))) break;
                    if (INSTRUMENT) InstrumentCollect(in_len, in_edges, 4);
                    continue;  // (selector unchanged: no queue was written)
                }
                if (problem->binned_min_edges > 0 && static_cast<long long>(queue_edges) >= problem->binned_min_edges) {
                    binned_level = true;
                    // ---- destination-binned level (oprtr/advance/binned.hpp): no claim atomics ----
                    int expand_grid = util::ResidentGrid(oprtr::advance::BinnedExpandKernel<BinnedPolicy, BFSProblem, BfsFunctor>,
                                                         BinnedPolicy::THREADS);
                    if (max_grid_size > 0) expand_grid = max_grid_size;
                    const int apply_grid = util::ResidentGrid(
                        oprtr::advance::BinnedApplyKernel<256, BFSProblem, BfsFunctor, BFSProblem::MARK_PREDECESSORS>, 256);
                    if ((retval = problem->EnsureBinned(expand_grid, work_progress.d_overflow))) break;
                    ds = problem->data_slices[0];
                    if ((retval = problem->bin_pool.Arm(stream))) break;
                    args.bins = problem->bin_pool.view;
                    typename BFSProblem::DataSlice expand_slice = *ds, apply_slice = *ds;
                    expand_slice.lite = 3;  // phase 1: screen against the visited bitmap (constant during the level)
                    apply_slice.lite = 2;   // phase 2: screen + claim on the destination's flag byte, on its owner XCD
                    retval = oprtr::advance::LaunchBinned<BinnedPolicy, BFSProblem, BfsFunctor>(args, expand_slice, apply_slice,
                                                                                               expand_grid, apply_grid, stream);
                    if (retval) break;
                    // closing sweep: flag bytes -> labels (vertex order), visited bitmap, this level's discoveries as a bitmap ...
                    int out_mask = 0;
                    if ((retval = problem->AcquireMask(stream, out_mask))) break;
                    if ((retval = launch_fresh_pass(reinterpret_cast<const unsigned long long *>(ds->d_visited_mask), nullptr, out_mask))) break;
                    // ... and as the next queue (vertex order, exact degrees), its packed tail in this level's output ring slot
                    hipLaunchKernelGGL((oprtr::advance::BitmapToQueueKernel<256, VertexId, SizeT>), dim3(conv_grid), dim3(256), 0,
                                       stream, ds->d_frontier_mask[out_mask], problem->nodes, gs->frontier_queues[selector ^ 1],
                                       work_progress.d_tail + ((iteration + 1) & 3), work_progress.d_overflow, gs->d_row_offsets);
                    if ((retval = util::GRError("BitmapToQueueKernel launch failed", __FILE__, __LINE__))) break;
                } else if ((retval = oprtr::advance::LaunchKernel<AdvancePolicy, BFSProblem, BfsFunctor>(
                         args, *ds, max_grid_size, stream, oprtr::advance::V2V)))
                    break;
                selector ^= 1;
            }
            if (INSTRUMENT && (retval = InstrumentEnd(stream))) break;

            ++iteration;
            if (bottom_up) {
                // (a sweep that emitted no queue leaves nothing in its output ring slot worth keeping: the read-back zeroes the
                //  slot, so a bitmap -> queue conversion can follow without a clear of its own)
                out_slot_clean = !queue_emitted;
                if ((retval = work_progress.Sync(stream, queue_emitted ? 0u : (1u << (iteration & 3))))) break;
                queue_length = util::TailCount(work_progress.box->wide);  // finds of the sweep
                queue_edges = 0;
                if (queue_emitted) {  // the ring slot holds the emitted queue's packed tail, not a count to add
                    if (work_progress.h_tail[util::WorkProgress::kAux] != 0) {  // a staging buffer overflowed: no usable queue
                        queue_emitted = false;
                        out_slot_clean = false;  // (the slot holds the partial queue's tail)
                        if ((retval = work_progress.ClearAux(stream))) break;
                    }
                } else {
                    queue_length += util::TailCount(work_progress.h_tail[iteration & 3]);
                }
            } else if ((retval = work_progress.GetTail(static_cast<int>(iteration & 3), queue_length, queue_edges, stream)))
                break;
            if (INSTRUMENT) InstrumentCollect(in_len, in_edges, bottom_up ? 1 : (binned_level ? 7 : 0));
            if (DEBUG) std::printf("iteration %lld (%s): queue length %u, edges %u\n", iteration,
                                   bottom_up ? "bottom-up" : "top-down", queue_length, queue_edges);
        }
        enactor_stats.iteration = iteration;
        if (retval) return retval;
        if (problem->labels_deferred) {  // every label of the search in one coalesced pass (bfs_problem.hpp EmitLabelsKernel)
            if (INSTRUMENT && (retval = InstrumentBegin(stream))) return retval;
            if ((retval = problem->EmitLabels(stream))) return retval;
            if (INSTRUMENT) {
                if ((retval = InstrumentEnd(stream))) return retval;
                if ((retval = util::GRError(hipStreamSynchronize(stream), "BFSEnactor EmitLabels sync failed", __FILE__, __LINE__))) return retval;
                InstrumentCollect(0, 0, 9);
            }
        }
        if (INSTRUMENT && (retval = DutyCollect(stream))) return retval;

        // (the loop's last read-back followed its last kernel, and SetTail cleared the flag of the previous search)
        const bool overflow = work_progress.OverflowAtLastSync();
        if (overflow) {
            // same diagnosis as bfs_enactor.cuh:540-545
            retval = util::GRError(hipErrorInvalidConfiguration,
                                   "Frontier queue overflow. Please increase queue-sizing factor.", __FILE__, __LINE__);
        }
        return retval;
    }
};

}  // namespace bfs
}  // namespace app
}  // namespace gunrock
