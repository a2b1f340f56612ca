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

namespace gunrock {
namespace app {
namespace bc {

template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct ForwardFunctor {
    typedef typename ProblemData::DataSlice DataSlice;

    static __device__ __forceinline__ bool CondEdge(VertexId s_id, VertexId d_id, DataSlice *problem, VertexId /*e_id*/ = 0,
                                                    VertexId /*e_id_in*/ = 0)
    {
        const VertexId new_label = problem->iteration + 1;  // s_id is in the frontier of level `iteration`
        const VertexId old = atomicCAS(problem->d_labels + d_id, static_cast<VertexId>(-1), new_label);
        if (old == -1 || old == new_label) atomicAdd(problem->d_sigmas + d_id, problem->d_sigmas[s_id]);
        return old == -1;  // the discovering edge enqueues d (exactly one per vertex)
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
        const Value result = problem->d_sigmas[s_id] / problem->d_sigmas[d_id] * (static_cast<Value>(1) + problem->d_deltas[d_id]);
        if (s_id != problem->src_node) {
            atomicAdd(problem->d_deltas + s_id, result);
            atomicAdd(problem->d_bc_values + s_id, result);
        }
    }
    static __device__ __forceinline__ bool CondFilter(VertexId node, DataSlice * /*problem*/, Value /*v*/ = 0, SizeT /*nid*/ = 0)
    {
        return node != -1;
    }
    static __device__ __forceinline__ void ApplyFilter(VertexId /*node*/, DataSlice * /*problem*/, Value /*v*/ = 0, SizeT /*nid*/ = 0) {}
};

}  // namespace bc
}  // namespace app
}  // namespace gunrock
