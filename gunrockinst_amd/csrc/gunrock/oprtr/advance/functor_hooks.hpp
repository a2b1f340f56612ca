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

}  // namespace advance
}  // namespace oprtr
}  // namespace gunrock
