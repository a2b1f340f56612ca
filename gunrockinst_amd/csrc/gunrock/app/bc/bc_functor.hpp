// app/bc/bc_functor.hpp -- functors of Brandes' betweenness centrality for the advance operator.
//
// Roles as in the reference (gunrock/app/bc/bc_functor.cuh):
//   ForwardFunctor  (:33-137)  BFS from the source that also counts shortest paths: a destination one level below the
//                              source side receives sigma[d] += sigma[s]; the edge that discovers d enqueues it.
//   BackwardFunctor (:145-237) dependency accumulation, deepest level first: for an edge s -> d with
//                              label[d] == label[s] + 1: delta[s] += sigma[s] / sigma[d] * (1 + delta[d]); the same amount
//                              goes to bc_values[s] unless s is the source (:205-208).
// The reference claims a child with atomicCAS on preds and then repairs the label with a second atomicCAS (:47-76); here
// one atomicCAS on the label decides both questions: the old value says whether this edge discovered d (-1) or d already
// carries this level's label -- in either case the edge lies on shortest paths and contributes sigma.
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/util/device_intrinsics.hpp>

namespace gunrock {
namespace app {
namespace bc {

template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct ForwardFunctor {
    typedef typename ProblemData::DataSlice DataSlice;

    // side-effect-free screen (optional advance hook): a destination labelled on an EARLIER level takes no part -- no
    // memory-side atomic for it (most edges of the big levels)
    static __device__ __forceinline__ bool ScreenEdge(VertexId /*s_id*/, VertexId d_id, DataSlice *problem, VertexId /*e_id*/ = 0,
                                                      VertexId /*e_id_in*/ = 0)
    {
        const VertexId label = problem->d_labels[d_id];  // (a stale -1 only costs the atomicCAS below)
        return label == -1 || label == problem->iteration + 1;
    }

    static __device__ __forceinline__ bool CondEdge(VertexId s_id, VertexId d_id, DataSlice *problem, VertexId /*e_id*/ = 0,
                                                    VertexId /*e_id_in*/ = 0)
    {
        const VertexId new_label = problem->iteration + 1;  // s_id is in the frontier of level `iteration`
        const VertexId old = atomicCAS(problem->d_labels + d_id, static_cast<VertexId>(-1), new_label);
        if (old == -1 || old == new_label) atomicAdd(problem->d_sigmas + d_id, problem->d_sigmas[s_id]);
        return old == -1;  // the discovering edge enqueues d (exactly one per vertex)
    }
    // ---- staged forms (oprtr/advance/functor_hooks.hpp): the source's path count arrives with the staged frontier entry, the
    // atomicCAS of every surviving edge of a tile is issued before any of them is examined (CondEdge alone waits for its own CAS
    // and then loads sigma[s] inside the per-edge branch: two exposed round trips per edge) ----
    struct EdgeState {
        Value sigma_s;        // path count of the source, as staged with the frontier entry
        VertexId label_seen;  // the destination's label as the screen read it
    };
    static __device__ __forceinline__ unsigned SourceData(VertexId s_id, DataSlice *problem)
    {
        static_assert(sizeof(Value) == 4, "32-bit path counts");
        return __float_as_uint(problem->d_sigmas[s_id]);
    }
    static __device__ __forceinline__ bool ScreenEdge(VertexId s_id, VertexId d_id, DataSlice *problem, VertexId e_id, VertexId e_id_in,
                                                      unsigned source_data, EdgeState &state)
    {
        state.sigma_s = __uint_as_float(source_data);
        const VertexId label = problem->d_labels[d_id];  // (a stale -1 only costs the atomicCAS below)
        state.label_seen = label;
        return label == -1 || label == problem->iteration + 1;
    }
    // Round 3: a destination the screen already saw on THIS level's label needs no claim -- some earlier edge discovered it -- only
    // the path-count add.  On a scale-free level a vertex is reached over a dozen edges: all but the first few claims were
    // memory-side atomics that changed nothing (they retire at ~27 G/s whether they change anything or not).
    static __device__ __forceinline__ VertexId IssueEdge(VertexId, VertexId d_id, DataSlice *problem, VertexId, VertexId, unsigned, EdgeState &state)
    {
        const VertexId level = static_cast<VertexId>(problem->iteration + 1);
        if (state.label_seen == level) return level;
        return atomicCAS(problem->d_labels + d_id, static_cast<VertexId>(-1), level);
    }
    static __device__ __forceinline__ bool ResolveEdge(VertexId old, VertexId, VertexId d_id, DataSlice *problem, VertexId, VertexId, EdgeState &state)
    {
        // discovered here, or already on this level: either way the edge lies on shortest paths (result unused: nothing waits)
        if (old == -1 || old == problem->iteration + 1) atomicAdd(problem->d_sigmas + d_id, state.sigma_s);
        return old == -1;
    }
    static __device__ __forceinline__ void ApplyEdge(VertexId /*s_id*/, VertexId /*d_id*/, DataSlice * /*problem*/,
                                                     VertexId /*e_id*/ = 0, VertexId /*e_id_in*/ = 0)
    {
    }
    static __device__ __forceinline__ bool CondFilter(VertexId node, DataSlice * /*problem*/, Value /*v*/ = 0, SizeT /*nid*/ = 0)
    {
        return node != -1;
    }
    static __device__ __forceinline__ void ApplyFilter(VertexId /*node*/, DataSlice * /*problem*/, Value /*v*/ = 0, SizeT /*nid*/ = 0) {}
};

// (The reference's shape of the backward functor, kept for callers written against it; BCEnactor itself runs the backward phase
//  with BackwardReduceFunctor below.)
template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct BackwardFunctor {
    typedef typename ProblemData::DataSlice DataSlice;

    static __device__ __forceinline__ bool CondEdge(VertexId /*s_id*/, VertexId d_id, DataSlice *problem, VertexId /*e_id*/ = 0,
                                                    VertexId /*e_id_in*/ = 0)
    {
        return problem->d_labels[d_id] == problem->iteration + 1;  // label[s] == iteration for the whole frontier
    }
    static __device__ __forceinline__ void ApplyEdge(VertexId s_id, VertexId d_id, DataSlice *problem, VertexId /*e_id*/ = 0,
                                                     VertexId /*e_id_in*/ = 0)
    {
        // the reference adds the same amount to bc_values[s] here (bc_functor.cuh:205-208); deltas are final once the backward
        // phase is over, so bc_values += deltas runs once per vertex afterwards (BCEnactor) instead of once per edge
        const Value result = problem->d_sigmas[s_id] / problem->d_sigmas[d_id] * (static_cast<Value>(1) + problem->d_deltas[d_id]);
        if (s_id != problem->src_node) atomicAdd(problem->d_deltas + s_id, result);
    }
    // Wave form of ApplyEdge (advance hook, oprtr/advance/kernel.hpp): consecutive lanes hold consecutive edge slots, so the
    // edges of one source form runs of lanes.  A segmented sum over each run (6 shuffle steps) leaves the run's total in
    // its last lane, which issues the ONE atomicAdd -- a hub with 2e5 edges was 2e5 atomics on one address (they retire at
    // ~80 per microsecond), now ~3e3.
    static __device__ __forceinline__ void ApplyEdgeWave(VertexId s_id, VertexId d_id, bool live, DataSlice *problem, VertexId /*e_id*/ = 0,
                                                         VertexId /*e_id_in*/ = 0)
    {
        const unsigned lane = util::LaneId();
        live = live && s_id != problem->src_node;
        Value val = 0;
        if (live) val = problem->d_sigmas[s_id] / problem->d_sigmas[d_id] * (static_cast<Value>(1) + problem->d_deltas[d_id]);
        // run = maximal stretch of consecutive lanes with the same live source (an edge of the row that failed CondEdge ends a
        // run: its lane is "dead").  Run ids count the stretch starts up to the lane, so equal ids <=> same contiguous stretch.
        const int key = live ? static_cast<int>(s_id) : -1;
        const int prev_key = __shfl_up(key, 1, util::kWaveSize);
        const bool starts = lane == 0 || prev_key != key || !live;
        const unsigned long long start_mask = __ballot(starts);
        const int run = __popcll(start_mask & ((2ull << lane) - 1ull));  // stretch starts at or before this lane
#pragma unroll
        for (int o = 1; o < util::kWaveSize; o <<= 1) {
            const int other_run = __shfl_up(run, o, util::kWaveSize);
            const Value other_val = __shfl_up(val, o, util::kWaveSize);
            if (static_cast<int>(lane) >= o && other_run == run) val += other_val;
        }
        const int next_run = __shfl_down(run, 1, util::kWaveSize);
        const bool run_end = lane == util::kWaveSize - 1 || next_run != run;
        if (live && run_end) atomicAdd(problem->d_deltas + s_id, val);
    }
    static __device__ __forceinline__ bool CondFilter(VertexId node, DataSlice * /*problem*/, Value /*v*/ = 0, SizeT /*nid*/ = 0)
    {
        return node != -1;
    }
    static __device__ __forceinline__ void ApplyFilter(VertexId /*node*/, DataSlice * /*problem*/, Value /*v*/ = 0, SizeT /*nid*/ = 0) {}
};

// Backward phase as a REDUCING advance (advance::LaunchReduce, PLUS, results by vertex): delta[s] = sum over the out-edges s -> d
// with label[d] == label[s] + 1 of sigma[s] / sigma[d] * (1 + delta[d]).  The operator keeps the per-edge values in registers,
// sums each run of a source's consecutive edge slots inside the wave and issues one store (the whole list sat in the run) or one
// atomicAdd per run -- what BackwardFunctor::ApplyEdgeWave does by hand, but with all loads of a tile in flight together
// (ApplyEdgeWave is called slot by slot: one exposed round trip each).
template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct BackwardReduceFunctor {
    typedef typename ProblemData::DataSlice DataSlice;
    static __device__ __forceinline__ bool CondEdge(VertexId, VertexId, DataSlice *, VertexId = 0, VertexId = 0) { return true; }
    static __device__ __forceinline__ void ApplyEdge(VertexId, VertexId, DataSlice *, VertexId = 0, VertexId = 0) {}
    // the value an edge s -> d contributes to delta[s] -- zero unless d lies one level below s.  Branch-free: an edge that fails the label test reads the source's own entries (hot lines) instead of
    // the destination's, so a tile's label loads and then its sigma / delta loads are in flight together.
    // Round 3: label and term come from ONE 8-byte gather of d_packed[d] (BCEnactor packs (label, 1 / sigma) after the forward phase
    // and refreshes the term of a level's vertices once their deltas are final); the three separate 4-byte gathers were up to three
    // 64-byte sectors per edge.  The division happens once per vertex, not once per edge.
    static __device__ __forceinline__ Value ReduceValue(VertexId s_id, VertexId d_id, DataSlice *problem, VertexId /*e_id*/ = 0,
                                                        VertexId /*e_id_in*/ = 0)
    {
        static_assert(sizeof(Value) == 4, "32-bit dependencies");
        const int2 p = problem->d_packed[d_id];
        const bool below = p.x == problem->iteration + 1;
        return below ? problem->d_sigmas[s_id] * __int_as_float(p.y) : static_cast<Value>(0);
    }
    static __device__ __forceinline__ bool CondFilter(VertexId node, DataSlice *, Value = 0, SizeT = 0) { return node != -1; }
    static __device__ __forceinline__ void ApplyFilter(VertexId, DataSlice *, Value = 0, SizeT = 0) {}
};

}  // namespace bc
}  // namespace app
}  // namespace gunrock
