// app/bfs/bfs_functor.hpp -- BFS functor for the advance / filter operators.
//
// Signature and role identical to the reference's BFSFunctor (gunrock/app/bfs/bfs_functor.cuh:33-122):
//   CondEdge(s_id, d_id, problem, e_id, e_id_in)  -> does this edge discover d_id?
//   ApplyEdge(...)                                 -> record the discovery
//   CondFilter(node, problem, v, nid)              -> keep `node` in the compacted frontier?
//   ApplyFilter(...)
// Reference semantics: non-idempotent modes claim the vertex with atomicCAS on
// labels/preds (bfs_functor.cuh:56-58); idempotent mode accepts every edge and lets the filter cull
// (bfs_functor.cuh:51-52, filter/cta.cuh:166-253).  All of them end with label = BFS depth.
//
// Here one rule serves all four modes: test-and-set the destination's bit in the L2-resident visited
// bitmap.  A relaxed plain load screens out already-visited destinations (the common case on the big
// R-MAT levels); only survivors pay the agent-scope atomicOr, whose return value elects exactly one
// discoverer.  The winner writes label = iteration + 1 (the value the reference passes as
// `label`/`iteration+1`, edge_map_partitioned/kernel.cuh:401-403, bfs_enactor.cuh:483) and, when
// MARK_PREDECESSORS, pred = s_id -- a valid parent because s_id is in the current frontier.
#pragma once

#include <hip/hip_runtime.h>

namespace gunrock {
namespace app {
namespace bfs {

template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct BFSFunctor {
    typedef typename ProblemData::DataSlice DataSlice;

    // side-effect-free screen (optional advance hook): is the destination already visited?
    static __device__ __forceinline__ bool ScreenEdge(VertexId /*s_id*/, VertexId d_id, DataSlice *problem,
                                                      VertexId /*e_id*/ = 0, VertexId /*e_id_in*/ = 0)
    {
        // ONE branch-free 32-bit load at a (uniformly) selected address.  The advance calls this for a whole batch of edges before it
        // looks at any result; a branch per edge would make the compiler wait for each load before issuing the next (measured:
        // the probes of a tile then cost one memory round trip EACH instead of one together).
        //  * lite == 2 -- phase 2 of a binned level (oprtr/advance/binned.hpp): the destination's flag byte; this workgroup runs
        //    on the XCD that owns it, so a load served by that XCD's L2 sees every earlier claim of the level;
        //  * otherwise the destination's bit of the visited bitmap.
        // With atomic claims (lite == 0) and in phase 2 the load bypasses L1 (global_load sc1, served by the XCD's L2): a line
        // parked in this CU's L1 is never refreshed during the launch, so hub words would keep reading "unvisited" and every
        // edge into a hub discovered on this level would pay a memory-side atomic.
        const bool flags = problem->lite == 2;
        const unsigned *base = flags ? reinterpret_cast<const unsigned *>(problem->d_fresh) : problem->d_visited_mask;
        const unsigned index = static_cast<unsigned>(d_id) >> (flags ? 2 : 5);
        const unsigned shift = flags ? (static_cast<unsigned>(d_id) & 3u) * 8u : (static_cast<unsigned>(d_id) & 31u);
        // lite == 1 / 3 (count-only level, phase 1 of a binned level): nothing writes the bitmap during the launch -> plain load
        // (hub words hit in L1; measured a few percent faster than sc1 on the scale-24 levels)
        const bool cached = problem->lite == 1 || problem->lite == 3;
        const unsigned word = cached ? base[index] : __hip_atomic_load(base + index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return ((word >> shift) & (flags ? 0xFFu : 1u)) == 0;  // (bitmap: stale-tolerant, a miss only costs an atomic)
    }

    static __device__ __forceinline__ bool CondEdge(VertexId /*s_id*/, VertexId d_id, DataSlice *problem,
                                                    VertexId /*e_id*/ = 0, VertexId /*e_id_in*/ = 0)
    {
        if (problem->lite) {  // (binned phase 2 included: ScreenEdge saw the flag clear; two claims inside one store latency
                              //  both pass, harmlessly -- the closing sweep dedupes, every same-level source is a valid parent)
            // count-only level (oprtr/advance/bottom_up.hpp FreshToBitmapKernel): every edge into an unvisited vertex may
            // "discover" it -- all sources of one level are equally valid parents -- so a plain byte store replaces the claim
            // (A best-effort same-level filter in the bitmap -- plain byte read-modify-write, the reference's bitmask cull,
            //  filter/cta.cuh:196-201 -- was measured too: it cost more than the duplicate byte stores it saved.)
            problem->d_fresh[d_id] = 1;
            return true;
        }
        unsigned *word = problem->d_visited_mask + (static_cast<unsigned>(d_id) >> 5);
        const unsigned bit = 1u << (d_id & 31);
        return (atomicOr(word, bit) & bit) == 0;          // exactly one winner per vertex
    }

    // CondEdge in two halves (advance hook): all claims of a tile are issued before any returned word is examined.
    // Token: the bitmap word as the atomicOr returned it, untouched (a test here would make the wave wait for the atomic
    // inside the branch that guards the call); lite levels claim nothing and return a clear word.
    static __device__ __forceinline__ unsigned IssueEdge(VertexId /*s_id*/, VertexId d_id, DataSlice *problem, VertexId /*e_id*/ = 0,
                                                         VertexId /*e_id_in*/ = 0)
    {
        if (problem->lite) {
            problem->d_fresh[d_id] = 1;
            return 0u;
        }
        return atomicOr(problem->d_visited_mask + (static_cast<unsigned>(d_id) >> 5), 1u << (d_id & 31));
    }
    static __device__ __forceinline__ bool ResolveEdge(unsigned token, VertexId, VertexId d_id, DataSlice *, VertexId = 0, VertexId = 0)
    {
        return (token & (1u << (d_id & 31))) == 0;  // the bit was clear: this edge claimed the vertex
    }

    static __device__ __forceinline__ void ApplyEdge(VertexId s_id, VertexId d_id, DataSlice *problem,
                                                     VertexId /*e_id*/ = 0, VertexId /*e_id_in*/ = 0)
    {
        if (!problem->lite) problem->d_labels[d_id] = problem->iteration + 1;  // (lite: FreshToBitmapKernel labels in vertex order)
        if (ProblemData::MARK_PREDECESSORS) problem->d_preds[d_id] = s_id;
    }

    static __device__ __forceinline__ bool CondFilter(VertexId node, DataSlice * /*problem*/, Value /*v*/ = 0,
                                                      SizeT /*nid*/ = 0)
    {
        return node != -1;
    }

    static __device__ __forceinline__ void ApplyFilter(VertexId /*node*/, DataSlice * /*problem*/, Value /*v*/ = 0,
                                                       SizeT /*nid*/ = 0)
    {
        // labels were written by the unique discoverer in ApplyEdge
    }
};

}  // namespace bfs
}  // namespace app
}  // namespace gunrock
