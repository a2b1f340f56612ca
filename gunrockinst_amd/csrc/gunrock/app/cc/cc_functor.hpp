// app/cc/cc_functor.hpp -- the seven filter functors of the hook / pointer-jump algorithm.
//
// Names, signatures and per-element effects follow the reference (gunrock/app/cc/cc_functor.cuh:18-410);
// `node` is an EDGE id for the hook functors and a VERTEX id for the others.  All writes store a strictly
// smaller id into a larger slot, so component_ids[v] <= v always holds and the smallest vertex of a component
// is never overwritten: the fixed point is component_ids[v] = min id of v's component for any interleaving
// (SURVEY 8(a) C3).  All accesses are PLAIN loads and stores: a sweep may read values that other lanes have already
// replaced (per-XCD L2s and per-CU L1s are not coherent inside a launch) -- that only delays convergence, because every
// value ever stored in component_ids[v] is an id of v's component not larger than v, and the convergence flags are
// re-evaluated by the next launch on fresh data.  Measured at scale-24: write-through (sc1) stores of the two flag
// words from millions of lanes serialised at the memory side and made a sweep ~40x slower than its traffic; plain
// stores collapse in the write-back L2.
#pragma once

#include <hip/hip_runtime.h>

namespace gunrock {
namespace app {
namespace cc {

template <typename T>
__device__ __forceinline__ T LoadFresh(const T *p)
{
    return *p;
}
template <typename T>
__device__ __forceinline__ void StoreFresh(T *p, T v)
{
    *p = v;
}

// mask[v] = (v is its own parent) ? 0 : 1      -- cc_functor.cuh:30-47
template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct UpdateMaskFunctor {
    typedef typename ProblemData::DataSlice DataSlice;
    static __device__ __forceinline__ bool CondFilter(VertexId, DataSlice *, Value = 0, SizeT = 0) { return true; }
    static __device__ __forceinline__ void ApplyFilter(VertexId node, DataSlice *problem, Value = 0, SizeT = 0)
    {
        problem->d_masks[node] = (problem->d_component_ids[node] == node) ? 0 : 1;
    }
};

// first hook: parent[max(from,to)] = min(from,to)      -- cc_functor.cuh:78-104
template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct HookInitFunctor {
    typedef typename ProblemData::DataSlice DataSlice;
    static __device__ __forceinline__ bool CondFilter(VertexId, DataSlice *, Value = 0, SizeT = 0) { return true; }
    static __device__ __forceinline__ void ApplyFilter(VertexId edge, DataSlice *problem, Value = 0, SizeT = 0)
    {
        const VertexId f = problem->d_froms[edge], t = problem->d_tos[edge];
        // Only the orientation whose SOURCE is the larger end hooks here: parent[f] = t.  All such edges of a row hit one
        // address (cheap), where the other orientation is 265 M scattered 4-byte stores at scale-24 (2.6 ms of an 8 ms run).
        // On a symmetric graph -- what the reference's CC drivers build -- the mirrored edge (t, f) performs exactly the hook
        // this skips; a directed edge without a mirror is hooked by the first HookMax sweep instead.  HookInit is only the
        // opening move: any set of "larger id -> smaller id" hooks keeps the invariant, the result (min id per component)
        // does not depend on it.
        if (f <= t) return;
        StoreFresh(problem->d_component_ids + f, t);
    }
};

// The same opening move as a VERTEX sweep, for the compact layout of a mirrored input (cc_problem.hpp): every from > to edge of a
// row writes parent[from] = to and the last store wins, i.e. "some neighbour below me becomes my parent" -- the row's first
// entry, precomputed per vertex, is one of them.  n 4-byte reads instead of a pass over the edge list.
template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct HookInitRowFunctor {
    typedef typename ProblemData::DataSlice DataSlice;
    static __device__ __forceinline__ bool CondFilter(VertexId, DataSlice *, Value = 0, SizeT = 0) { return true; }
    static __device__ __forceinline__ void ApplyFilter(VertexId node, DataSlice *problem, Value = 0, SizeT = 0)
    {
        const VertexId below = problem->d_first_lower[node];
        if (below < node) StoreFresh(problem->d_component_ids + node, below);
    }
};

// A NEIGHBOUR ROUND (Afforest's first phase inside this schedule): HookMax applied to ONE edge per vertex -- the edge to its
// (neighbour_round + 1)-th smallest lower neighbour.  n gathers instead of one per edge; after a couple of such rounds (each followed
// by the pointer jumps) a scale-free graph's giant component is essentially one tree, and the row-form sweeps below let everything
// rooted at it sit out.
template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct HookNeighbourFunctor {
    typedef typename ProblemData::DataSlice DataSlice;
    static __device__ __forceinline__ bool CondFilter(VertexId, DataSlice *, Value = 0, SizeT = 0) { return true; }
    static __device__ __forceinline__ void ApplyFilter(VertexId node, DataSlice *problem, Value = 0, SizeT = 0)
    {
        const SizeT at = problem->d_low_offsets[node] + problem->neighbour_round;
        if (at >= problem->d_low_offsets[node + 1]) return;
        const VertexId pf = LoadFresh(problem->d_component_ids + node);
        const VertexId pt = LoadFresh(problem->d_component_ids + problem->d_tos[at]);
        if (pf != pt) {
            const VertexId hi = pf > pt ? pf : pt, lo = pf > pt ? pt : pf;
            StoreFresh(problem->d_component_ids + hi, lo);
            StoreFresh(problem->d_edge_flag, 0);
        }
    }
};

// HookMax in ROW form, for a mirrored input with a dominant component (cc_problem.hpp PickGiantKernel; the idea of Afforest's
// "skip the largest intermediate component", Sutton et al., inside the reference's hook / jump schedule): a vertex whose root is
// the giant's root does nothing at all -- an edge between two such vertices needs no hook, and an edge to a vertex OUTSIDE the
// giant is seen from that vertex's own row, because every edge has its mirror.  Only the few vertices outside it walk their
// rows (both orientations) and hook exactly as HookMax does.  A scale-24 R-MAT sweep then reads 64 MB of parents instead of
// gathering a parent per edge (1.08 ms -> ~0.04 ms).  A row longer than row_form_limit raises word [2]: the enactor runs the
// edge form for that round as well (correct for any graph; the row form is only ever a shortcut).
template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct HookMaxRowFunctor {
    typedef typename ProblemData::DataSlice DataSlice;
    static __device__ __forceinline__ bool CondFilter(VertexId, DataSlice *, Value = 0, SizeT = 0) { return true; }
    static __device__ __forceinline__ void ApplyFilter(VertexId node, DataSlice *problem, Value = 0, SizeT = 0)
    {
        const VertexId pf = LoadFresh(problem->d_component_ids + node);
        VertexId giant = problem->d_giant[0];
        if (giant >= 0) {  // the giant's root may itself have been hooked under a smaller root since it was picked: follow it
#pragma unroll 1
            for (int hop = 0; hop < 16; ++hop) {
                const VertexId up = LoadFresh(problem->d_component_ids + giant);
                if (up == giant) break;
                giant = up;
            }
            if (pf == giant) return;
        }
        const SizeT begin = problem->d_row_offsets[node], end = problem->d_row_offsets[node + 1];
        if (end - begin > problem->row_form_limit) {
            problem->d_giant[2] = 1;
            return;
        }
        for (SizeT e = begin; e < end; ++e) {
            const VertexId pt = LoadFresh(problem->d_component_ids + problem->d_columns[e]);
            if (pt != pf) {
                const VertexId hi = pf > pt ? pf : pt, lo = pf > pt ? pt : pf;
                StoreFresh(problem->d_component_ids + hi, lo);
                StoreFresh(problem->d_edge_flag, 0);
            }
        }
    }
};

// hook the larger root under the smaller one; an edge whose ends share a root is marked done -- cc_functor.cuh:172-216
template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct HookMaxFunctor {
    typedef typename ProblemData::DataSlice DataSlice;
    static __device__ __forceinline__ bool CondFilter(VertexId, DataSlice *, Value = 0, SizeT = 0) { return true; }
    static __device__ __forceinline__ void ApplyFilter(VertexId edge, DataSlice *problem, Value = 0, SizeT = 0)
    {
        if (problem->d_marks[edge]) return;
        const VertexId f = problem->d_froms[edge], t = problem->d_tos[edge];
        // mirrored input (CCProblem checked it): the edge (t, f) does this very hook, so one orientation is enough; park the
        // other one for good
        if (problem->symmetric && f < t) {
            problem->d_marks[edge] = 1;
            return;
        }
        const VertexId pf = LoadFresh(problem->d_component_ids + f);
        const VertexId pt = LoadFresh(problem->d_component_ids + t);
        if (pf == pt) {
            problem->d_marks[edge] = 1;
        } else {
            const VertexId hi = pf > pt ? pf : pt, lo = pf > pt ? pt : pf;
            StoreFresh(problem->d_component_ids + hi, lo);
            StoreFresh(problem->d_edge_flag, 0);
        }
    }
};

// mirrored variant kept for API completeness (unused by the enactor, as in the reference: cc_enactor.cuh:540-559)
template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct HookMinFunctor {
    typedef typename ProblemData::DataSlice DataSlice;
    static __device__ __forceinline__ bool CondFilter(VertexId, DataSlice *, Value = 0, SizeT = 0) { return true; }
    static __device__ __forceinline__ void ApplyFilter(VertexId edge, DataSlice *problem, Value = 0, SizeT = 0)
    {
        if (problem->d_marks[edge]) return;
        const VertexId pf = LoadFresh(problem->d_component_ids + problem->d_froms[edge]);
        const VertexId pt = LoadFresh(problem->d_component_ids + problem->d_tos[edge]);
        if (pf == pt) {
            problem->d_marks[edge] = 1;
        } else {
            const VertexId hi = pf > pt ? pf : pt, lo = pf > pt ? pt : pf;
            StoreFresh(problem->d_component_ids + lo, hi);
            StoreFresh(problem->d_edge_flag, 0);
        }
    }
};

// Pointer jumping follows up to kJumpHops parents per sweep instead of the reference's one (parent[v] = parent[parent[v]],
// cc_functor.cuh:230-262): every value read on the way is an ancestor of v (ids only ever decrease towards the root), so the
// fixed point -- every vertex points at its root -- is the same; a round of sweeps converges in 2-3 launches instead of 5-7.
constexpr int kJumpHops = 8;
template <typename VertexId, typename DataSlice>
__device__ __forceinline__ VertexId JumpFrom(VertexId parent, DataSlice *problem)
{
#pragma unroll 1
    for (int hop = 0; hop < kJumpHops; ++hop) {
        const VertexId grand = LoadFresh(problem->d_component_ids + parent);
        if (grand == parent) break;
        parent = grand;
    }
    return parent;
}

template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct PtrJumpFunctor {
    typedef typename ProblemData::DataSlice DataSlice;
    static __device__ __forceinline__ bool CondFilter(VertexId, DataSlice *, Value = 0, SizeT = 0) { return true; }
    static __device__ __forceinline__ void ApplyFilter(VertexId node, DataSlice *problem, Value = 0, SizeT = 0)
    {
        const VertexId parent = LoadFresh(problem->d_component_ids + node);
        const VertexId grand = JumpFrom(parent, problem);
        if (parent != grand) {
            StoreFresh(problem->d_vertex_flag, 0);
            StoreFresh(problem->d_component_ids + node, grand);
        }
    }
};

// the same for vertices whose mask is 0; one already at a root gets mask -1      -- cc_functor.cuh:276-313
template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct PtrJumpMaskFunctor {
    typedef typename ProblemData::DataSlice DataSlice;
    static __device__ __forceinline__ bool CondFilter(VertexId, DataSlice *, Value = 0, SizeT = 0) { return true; }
    static __device__ __forceinline__ void ApplyFilter(VertexId node, DataSlice *problem, Value = 0, SizeT = 0)
    {
        if (problem->d_masks[node] != 0) return;
        const VertexId parent = LoadFresh(problem->d_component_ids + node);
        const VertexId grand = JumpFrom(parent, problem);
        if (parent != grand) {
            StoreFresh(problem->d_vertex_flag, 0);
            StoreFresh(problem->d_component_ids + node, grand);
        } else {
            problem->d_masks[node] = -1;
        }
    }
};

// one jump for vertices whose mask is 1      -- cc_functor.cuh:327-352
template <typename VertexId, typename SizeT, typename Value, typename ProblemData>
struct PtrJumpUnmaskFunctor {
    typedef typename ProblemData::DataSlice DataSlice;
    static __device__ __forceinline__ bool CondFilter(VertexId, DataSlice *, Value = 0, SizeT = 0) { return true; }
    static __device__ __forceinline__ void ApplyFilter(VertexId node, DataSlice *problem, Value = 0, SizeT = 0)
    {
        if (problem->d_masks[node] != 1) return;
        const VertexId parent = LoadFresh(problem->d_component_ids + node);
        const VertexId grand = LoadFresh(problem->d_component_ids + parent);
        StoreFresh(problem->d_component_ids + node, grand);
    }
};

}  // namespace cc
}  // namespace app
}  // namespace gunrock
