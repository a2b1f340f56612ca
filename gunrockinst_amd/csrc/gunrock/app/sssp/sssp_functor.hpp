// app/sssp/sssp_functor.hpp -- relaxation functor and priority-score functor for SSSP.
//
// Same roles and signatures as the reference (gunrock/app/sssp/sssp_functor.cuh:36-139):
//   SSSPFunctor::CondEdge  : new = dist[s] + weight[e];  return new < atomicMin(&dist[d], new)     (:52-64)
//   SSSPFunctor::ApplyEdge : record the predecessor                                                (:76-84)
//   PQFunctor::ComputePriorityScore : bucket = delta == 0 ? dist : dist / delta                   (:128-138)
// Overflow: the reference adds unsigned 32-bit values unchecked, so a path whose length wraps past 2^32 looks SHORT and
// (with huge or "negative" weights reinterpreted as unsigned) relaxation never settles.  Here a candidate that wraps or
// reaches UINT_MAX (the "unreachable" value) is rejected -- the saturating behaviour of the reference's CPU oracle
// (Boost closed_plus, tests/sssp/test_sssp.cu:242-343).  Identical results whenever no path length overflows.
// ScreenEdge (optional advance hook) rejects edges that cannot improve the destination with plain loads, so
// only potentially useful relaxations pay for the atomic.  With MARK_PATHS the atomicMin acts on the packed
// (distance << 32 | predecessor) word: the winning pair is consistent by construction.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

namespace gunrock {
namespace app {
namespace sssp {

template <typename VertexId, typename SizeT, typename ProblemData>
struct SSSPFunctor {
    typedef typename ProblemData::DataSlice DataSlice;

    static __device__ __forceinline__ bool ScreenEdge(VertexId s_id, VertexId d_id, DataSlice *problem, VertexId e_id = 0,
                                                      VertexId /*e_id_in*/ = 0)
    {
        const unsigned from = problem->Distance(s_id);
        const unsigned candidate = from + problem->d_weights[e_id];
        return candidate >= from && candidate != 0xFFFFFFFFu && candidate < problem->Distance(d_id);
    }

    static __device__ __forceinline__ bool CondEdge(VertexId s_id, VertexId d_id, DataSlice *problem, VertexId e_id = 0,
                                                    VertexId /*e_id_in*/ = 0)
    {
        const unsigned from = problem->Distance(s_id);
        const unsigned candidate = from + problem->d_weights[e_id];
        if (candidate < from || candidate == 0xFFFFFFFFu) return false;  // wrapped: not a path length
        if (ProblemData::MARK_PATHS) {
            const unsigned long long packed = (static_cast<unsigned long long>(candidate) << 32) | static_cast<unsigned>(s_id);
            const unsigned long long old = atomicMin(problem->d_dist_pred + d_id, packed);
            return candidate < static_cast<unsigned>(old >> 32);
        }
        return candidate < atomicMin(problem->d_labels + d_id, candidate);
    }

    // CondEdge in two halves (advance hook): the atomicMin of every edge of a tile is issued before any result is examined.
    // The token carries what the atomic returned UNTOUCHED (any arithmetic on it here would make the wave wait for the atomic
    // inside the branch that guards the call) next to the candidate; a wrapped candidate issues nothing and cannot win.
    struct Token {
        unsigned candidate;
        unsigned old_lo;            // previous distance (plain labels) ...
        unsigned long long old_hi;  // ... or previous packed (distance, predecessor)
    };
    static __device__ __forceinline__ Token IssueEdge(VertexId s_id, VertexId d_id, DataSlice *problem, VertexId e_id = 0,
                                                      VertexId /*e_id_in*/ = 0)
    {
        Token t;
        const unsigned from = problem->Distance(s_id);
        t.candidate = from + problem->d_weights[e_id];
        t.old_lo = 0u;
        t.old_hi = 0ull;
        if (t.candidate < from) t.candidate = 0xFFFFFFFFu;  // wrapped: not a path length
        if (t.candidate != 0xFFFFFFFFu) {
            if (ProblemData::MARK_PATHS)
                t.old_hi = atomicMin(problem->d_dist_pred + d_id, (static_cast<unsigned long long>(t.candidate) << 32) | static_cast<unsigned>(s_id));
            else
                t.old_lo = atomicMin(problem->d_labels + d_id, t.candidate);
        }
        return t;
    }
    static __device__ __forceinline__ bool ResolveEdge(const Token &t, VertexId, VertexId, DataSlice *, VertexId = 0, VertexId = 0)
    {
        const unsigned old = ProblemData::MARK_PATHS ? static_cast<unsigned>(t.old_hi >> 32) : t.old_lo;
        return t.candidate != 0xFFFFFFFFu && t.candidate < old;
    }

    static __device__ __forceinline__ void ApplyEdge(VertexId, VertexId, DataSlice *, VertexId = 0, VertexId = 0)
    {
        // the predecessor travels inside the packed atomicMin of CondEdge
    }

    // ---- staged forms (oprtr/advance/functor_hooks.hpp): the source's distance arrives with the staged frontier entry and the
    // candidate travels from the screen to the claim, so the claim loads nothing.  (The five-argument forms re-load the
    // source's distance and the weight inside the per-edge branch of the claim: one exposed round trip per edge, measured
    // 2.75 -> 2.55 ms per R-MAT scale-22 search.)
    // Measured and dropped: a 4-bit-per-vertex quantised distance bound (n/2 bytes, L2-resident) in front of the distance
    // gather rejected 77 % of the gathers and still cost 0.15 ms more than it saved -- the per-CU rate of random accesses is the
    // bound, and an L2 probe plus 23 % of the gathers is more random accesses than the gathers alone.
    typedef unsigned EdgeState;  // the candidate distance (UINT_MAX: wrapped, not a path length)
    typedef typename std::conditional<ProblemData::MARK_PATHS, unsigned long long, unsigned>::type StagedToken;
    static __device__ __forceinline__ unsigned SourceData(VertexId s_id, DataSlice *problem) { return problem->Distance(s_id); }
    static __device__ __forceinline__ bool ScreenEdge(VertexId, VertexId d_id, DataSlice *problem, VertexId e_id, VertexId, unsigned from,
                                                      EdgeState &candidate)
    {
        const unsigned sum = from + problem->d_weights[e_id];
        candidate = sum < from ? 0xFFFFFFFFu : sum;
        return (candidate != 0xFFFFFFFFu) & (candidate < problem->Distance(d_id));
    }
    static __device__ __forceinline__ StagedToken IssueEdge(VertexId s_id, VertexId d_id, DataSlice *problem, VertexId, VertexId, unsigned,
                                                            EdgeState &candidate)
    {
        if (ProblemData::MARK_PATHS)
            return static_cast<StagedToken>(atomicMin(problem->d_dist_pred + d_id, (static_cast<unsigned long long>(candidate) << 32) | static_cast<unsigned>(s_id)));
        return static_cast<StagedToken>(atomicMin(problem->d_labels + d_id, candidate));
    }
    static __device__ __forceinline__ bool ResolveEdge(StagedToken token, VertexId, VertexId, DataSlice *, VertexId, VertexId, EdgeState &candidate)
    {
        const unsigned old = ProblemData::MARK_PATHS ? static_cast<unsigned>(static_cast<unsigned long long>(token) >> 32) : static_cast<unsigned>(token);
        return candidate < old;
    }

    static __device__ __forceinline__ bool CondFilter(VertexId node, DataSlice *, unsigned = 0, SizeT = 0) { return node != -1; }
    static __device__ __forceinline__ void ApplyFilter(VertexId, DataSlice *, unsigned = 0, SizeT = 0) {}
};

// ---- pull form of a relaxation level (dense levels; sssp_enactor.hpp) ----
// The reducing advance runs over the IN-neighbour lists: s_id = the vertex that may improve, d_id = one of its in-neighbours,
// e_id = the in-edge (inverse-CSR position).  ReduceValue = that neighbour's distance + the edge's weight (packed with the
// neighbour when predecessors are kept); the operator takes the MINIMUM per vertex, no atomic per edge.
template <typename VertexId, typename SizeT, typename ProblemData>
struct SSSPPullFunctor {
    typedef typename ProblemData::DataSlice DataSlice;
    typedef typename std::conditional<ProblemData::MARK_PATHS, unsigned long long, unsigned>::type PullValue;

    static __device__ __forceinline__ bool CondEdge(VertexId, VertexId, DataSlice *, VertexId = 0, VertexId = 0) { return true; }
    static __device__ __forceinline__ void ApplyEdge(VertexId, VertexId, DataSlice *, VertexId = 0, VertexId = 0) {}
    static __device__ __forceinline__ PullValue ReduceValue(VertexId /*s_id*/, VertexId d_id, DataSlice *problem, VertexId e_id = 0,
                                                            VertexId /*e_id_in*/ = 0)
    {
        const unsigned from = problem->Distance(d_id);
        unsigned candidate = from + problem->d_inv_weights[e_id];
        if (candidate < from) candidate = 0xFFFFFFFFu;  // wrapped (an unreached neighbour wraps too): not a path length
        if (ProblemData::MARK_PATHS) return static_cast<PullValue>((static_cast<unsigned long long>(candidate) << 32) | static_cast<unsigned>(d_id));
        return static_cast<PullValue>(candidate);
    }
    // filter over all vertices after the reduction: a vertex whose best pulled candidate beats its distance takes it and
    // becomes a candidate of the near/far split, exactly like the destination of a successful push
    static __device__ __forceinline__ bool CondFilter(VertexId node, DataSlice *problem, int /*v*/ = 0, SizeT /*nid*/ = 0)
    {
        const PullValue best = static_cast<const PullValue *>(problem->d_pull)[node];
        if (ProblemData::MARK_PATHS) {
            const unsigned candidate = static_cast<unsigned>(static_cast<unsigned long long>(best) >> 32);
            if (candidate == 0xFFFFFFFFu || candidate >= problem->Distance(node)) return false;
            problem->d_dist_pred[node] = static_cast<unsigned long long>(best);
            return true;
        }
        const unsigned candidate = static_cast<unsigned>(best);
        if (candidate == 0xFFFFFFFFu || candidate >= problem->d_labels[node]) return false;
        problem->d_labels[node] = candidate;
        return true;
    }
    static __device__ __forceinline__ void ApplyFilter(VertexId, DataSlice *, int = 0, SizeT = 0) {}
};

template <typename VertexId, typename SizeT, typename ProblemData>
struct PQFunctor {
    typedef typename ProblemData::DataSlice DataSlice;
    static __device__ __forceinline__ unsigned ComputePriorityScore(VertexId node_id, DataSlice *problem)
    {
        const unsigned distance = problem->Distance(node_id);
        const float delta = problem->delta;
        return (delta == 0) ? distance : static_cast<unsigned>(distance / delta);
    }
};

}  // namespace sssp
}  // namespace app
}  // namespace gunrock
