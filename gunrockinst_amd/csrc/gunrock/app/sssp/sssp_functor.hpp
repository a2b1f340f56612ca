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

    static __device__ __forceinline__ void ApplyEdge(VertexId, VertexId, DataSlice *, VertexId = 0, VertexId = 0)
    {
        // the predecessor travels inside the packed atomicMin of CondEdge
    }

    static __device__ __forceinline__ bool CondFilter(VertexId node, DataSlice *, unsigned = 0, SizeT = 0) { return node != -1; }
    static __device__ __forceinline__ void ApplyFilter(VertexId, DataSlice *, unsigned = 0, SizeT = 0) {}
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
