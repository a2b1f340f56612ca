// oprtr/advance/functor_hooks.hpp -- optional static methods an advance functor may add to the reference's
// CondEdge / ApplyEdge pair (gunrock/app/bfs/bfs_functor.cuh:49-88); detected at compile time, so a functor written for
// the reference compiles unchanged.
#pragma once

#include <type_traits>

namespace gunrock {
namespace oprtr {
namespace advance {

// Optional functor hook: `static bool ScreenEdge(s_id, d_id, problem, e_id, e_id_in)` -- a side-effect-free
// pre-test evaluated for all of a thread's edges before any CondEdge runs, so its loads overlap.  Functors
// without it (the reference's functor shape, bfs_functor.cuh:49-88) are screened by `true`.
template <typename Functor, typename VertexId, typename DataSlice, typename = void>
struct HasScreenEdge : std::false_type {};
template <typename Functor, typename VertexId, typename DataSlice>
struct HasScreenEdge<Functor, VertexId, DataSlice,
                     std::void_t<decltype(Functor::ScreenEdge(VertexId(), VertexId(), static_cast<DataSlice *>(nullptr),
                                                              VertexId(), VertexId()))>> : std::true_type {};

template <typename Functor, typename VertexId, typename DataSlice>
__device__ __forceinline__ bool ScreenEdge(VertexId s, VertexId d, DataSlice *slice, VertexId e, VertexId e_in)
{
    if constexpr (HasScreenEdge<Functor, VertexId, DataSlice>::value) return Functor::ScreenEdge(s, d, slice, e, e_in);
    else return true;
}

// Optional functor hook `ApplyEdgeWave(s_id, d_id, live, problem, e_id, e_id_in)`: called INSTEAD of ApplyEdge, by every lane
// of the wave (live = this lane's edge passed CondEdge), so the functor may combine lanes -- consecutive lanes hold consecutive
// edge slots, i.e. runs of the same source -- before it touches memory (BC's dependency sums: one atomic per run, not per edge).
template <typename Functor, typename VertexId, typename DataSlice, typename = void>
struct HasApplyEdgeWave : std::false_type {};
template <typename Functor, typename VertexId, typename DataSlice>
struct HasApplyEdgeWave<Functor, VertexId, DataSlice,
                        std::void_t<decltype(Functor::ApplyEdgeWave(VertexId(), VertexId(), false, static_cast<DataSlice *>(nullptr),
                                                                    VertexId(), VertexId()))>> : std::true_type {};

// Optional split of CondEdge for functors whose test is a returning atomic:
//   `static T IssueEdge(s_id, d_id, problem, e_id, e_id_in)`   issues the atomic and hands back what it returned, unexamined;
//   `static bool ResolveEdge(T, s_id, d_id, problem, e_id, e_id_in)` decides from that value.
// The advance issues all of a thread's edges first and resolves them afterwards, so the atomics' round trips overlap; with
// CondEdge alone every edge waits for its own atomic before the next one is issued (the test sits inside the branch that
// guards the call).  CondEdge must stay equivalent to ResolveEdge(IssueEdge(...)): operators without the batch path call it.
template <typename Functor, typename VertexId, typename DataSlice, typename = void>
struct HasIssueEdge : std::false_type {};
template <typename Functor, typename VertexId, typename DataSlice>
struct HasIssueEdge<Functor, VertexId, DataSlice,
                    std::void_t<decltype(Functor::IssueEdge(VertexId(), VertexId(), static_cast<DataSlice *>(nullptr), VertexId(), VertexId()))>>
    : std::true_type {};

// Optional STAGED form of the screen / claim hooks, for functors whose edge test needs a property of the SOURCE vertex and a
// value computed from it per edge (SSSP: distance of the source, candidate distance of the edge):
//   `static unsigned SourceData(s_id, problem)`            32 bits, fetched ONCE per frontier entry while the entry is staged
//                                                           (a coalesced load a tile ahead), kept in LDS next to the entry;
//   `typedef ... EdgeState;`                                per-edge registers that travel from the screen to the claim;
//   `static bool ScreenEdge(s_id, d_id, problem, e_id, e_id_in, unsigned source_data, EdgeState &state)`
//   `static T    IssueEdge (s_id, d_id, problem, e_id, e_id_in, unsigned source_data, EdgeState &state)`
//   `static bool ResolveEdge(T, s_id, d_id, problem, e_id, e_id_in, EdgeState &state)`
// Without it the claim has to re-load what the screen already had (the compiler cannot keep a load across the atomics in
// between), inside the per-edge branch -- one exposed round trip per edge.  The five-argument forms must stay equivalent:
// operators without the staged path call them.
template <typename Functor, typename VertexId, typename DataSlice, typename = void>
struct HasSourceData : std::false_type {};
template <typename Functor, typename VertexId, typename DataSlice>
struct HasSourceData<Functor, VertexId, DataSlice, std::void_t<decltype(Functor::SourceData(VertexId(), static_cast<DataSlice *>(nullptr)))>>
    : std::true_type {};
struct NoEdgeState {};
template <typename Functor, bool STAGED>
struct EdgeStateOf {
    typedef NoEdgeState type;
};
template <typename Functor>
struct EdgeStateOf<Functor, true> {
    typedef typename Functor::EdgeState type;
};

// Optional hook of a REDUCING advance: `static V ReduceValue(s_id, d_id, problem, e_id, e_id_in)` computes the value an edge
// contributes (the reference leaves this case open: "use user-specified function to generate value to reduce",
// edge_map_partitioned/kernel.cuh:427-429); without it the value is d_value_to_reduce[d_id] / [e_id].  Side-effect free and
// safe for any valid (vertex, vertex, edge): it is evaluated for every slot of a tile before any result is used.
template <typename Functor, typename VertexId, typename DataSlice, typename = void>
struct HasReduceValue : std::false_type {};
template <typename Functor, typename VertexId, typename DataSlice>
struct HasReduceValue<Functor, VertexId, DataSlice,
                      std::void_t<decltype(Functor::ReduceValue(VertexId(), VertexId(), static_cast<DataSlice *>(nullptr), VertexId(), VertexId()))>>
    : std::true_type {};

}  // namespace advance
}  // namespace oprtr
}  // namespace gunrock
