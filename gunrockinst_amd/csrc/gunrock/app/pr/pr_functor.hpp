// app/pr/pr_functor.hpp -- PageRank functors for the advance / filter operators.
//
// Same rules as the reference's PRFunctor and RemoveZeroDegreeNodeFunctor (gunrock/app/pr/pr_functor.cuh:36-189):
//   edge (s -> d) takes part when both ends still have out-edges:      CondEdge   pr_functor.cuh:52-55
//   it moves rank_curr[s] / degree[s] into rank_next[d]:               ApplyEdge  pr_functor.cuh:68-71 (atomicAdd per edge)
//   a vertex then takes delta * rank_next + (1 - delta) * [it is the source, or there is no source] and stays "active" while
//   its rank moved by more than the threshold:                         CondFilter pr_functor.cuh:84-93
//   vertices without out-edges are peeled off first, round by round, each round lowering the degree of the vertices
//   that point at them:                                                pr_functor.cuh:123-170
// Here the per-edge atomicAdd is gone: ranks are PULLED.  The reducing advance (oprtr/advance/kernel.hpp LaunchReduce,
// the reference's R_TYPE / R_OP + SegReduceCsr) runs over the in-neighbour lists and sums contrib[u] = rank[u] / degree[u]
// per vertex; a float atomic only joins the pieces of a list that straddles waves.  (Scattered float atomics run at
// ~0.08 TB/s on MI355X; a push over 265 M edges would take ~13 ms per iteration.)  The peeling rounds use the same operator
// over the forward lists: how many of my out-neighbours have just lost their last edge.
#pragma once

#include <hip/hip_runtime.h>

namespace gunrock {
namespace app {
namespace pr {

template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct PRFunctor {
    typedef typename ProblemData::DataSlice DataSlice;

    // pull form: s_id = the vertex that receives, d_id = one of its in-neighbours.  Side-effect free, so it is evaluated for a
    // whole tile before any result is used (advance hook).
    // The reference's rule "the edge takes part when both ends still have out-edges" (pr_functor.cuh:52-55) needs no test per
    // edge here: a peeled in-neighbour's contribution is exactly 0 (ContribKernel; only surviving vertices are ever updated), and
    // the receiving vertex is in the frontier because it survived.  (Round 2 tested d_degrees[d_id] per edge: a second random
    // 4-byte gather -- one more 64-byte sector -- next to contrib[d_id]; adding the 0 instead is bit-identical.)
    static __device__ __forceinline__ bool ScreenEdge(VertexId, VertexId, DataSlice *, VertexId /*e_id*/ = 0, VertexId /*e_id_in*/ = 0)
    {
        return true;
    }
    static __device__ __forceinline__ bool CondEdge(VertexId, VertexId, DataSlice *, VertexId = 0, VertexId = 0) { return true; }
    static __device__ __forceinline__ void ApplyEdge(VertexId, VertexId, DataSlice *, VertexId = 0, VertexId = 0) {}

    static __device__ __forceinline__ bool CondFilter(VertexId node, DataSlice *problem, Value /*v*/ = 0, SizeT /*nid*/ = 0)
    {
        const Value delta = problem->delta;
        const VertexId src_node = problem->src_node;
        const Value next = delta * problem->d_rank_next[node] +
                           (static_cast<Value>(1) - delta) * ((src_node == node || src_node == -1) ? static_cast<Value>(1) : static_cast<Value>(0));
        const Value diff = fabsf(next - problem->d_rank_curr[node]);
        // the reference copies rank_next over rank_curr for every vertex after the filter (pr_enactor.cuh:478-485); the copy
        // and next iteration's rank / degree division happen here, once per active vertex
        problem->d_rank_curr[node] = next;
        problem->d_contrib[node] = next / static_cast<Value>(problem->d_degrees[node]);
        return diff > problem->threshold;
    }
    static __device__ __forceinline__ void ApplyFilter(VertexId, DataSlice *, Value = 0, SizeT = 0) {}
};

template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct RemoveZeroDegreeNodeFunctor {
    typedef typename ProblemData::DataSlice DataSlice;
    static __device__ __forceinline__ bool CondEdge(VertexId, VertexId, DataSlice *, VertexId = 0, VertexId = 0) { return true; }
    static __device__ __forceinline__ void ApplyEdge(VertexId, VertexId, DataSlice *, VertexId = 0, VertexId = 0) {}
    // keep the vertices that still have out-edges after this round
    static __device__ __forceinline__ bool CondFilter(VertexId node, DataSlice *problem, Value /*v*/ = 0, SizeT /*nid*/ = 0)
    {
        return problem->d_degrees_pong[node] > 0;
    }
    static __device__ __forceinline__ void ApplyFilter(VertexId, DataSlice *, Value = 0, SizeT = 0) {}
};

// every vertex, even those without any edge: the identity queue
template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct HasOutEdgesFunctor {
    typedef typename ProblemData::DataSlice DataSlice;
    static __device__ __forceinline__ bool CondFilter(VertexId node, DataSlice *problem, Value /*v*/ = 0, SizeT /*nid*/ = 0)
    {
        return problem->d_degrees[node] > 0;
    }
    static __device__ __forceinline__ void ApplyFilter(VertexId, DataSlice *, Value = 0, SizeT = 0) {}
};

}  // namespace pr
}  // namespace app
}  // namespace gunrock
