// oprtr/advance/kernel.hpp -- the ADVANCE (edge-expand) operator for gfx950.
//
// Public surface kept from the reference (gunrock/oprtr/advance/kernel.cuh:101-129,
// kernel_policy.cuh:43-174): namespace gunrock::oprtr::advance, enums MODE / TYPE, a KernelPolicy,
// and LaunchKernel<KernelPolicy, ProblemData, Functor>() that calls the user functor's
// CondEdge / ApplyEdge for every out-edge of every input-frontier vertex and enqueues the accepted
// destinations.
//
// Implementation is new (no moderngpu, no device-wide scan / sorted search per call):
//   * the input frontier already carries the exclusive degree prefix (util/frontier.hpp), so the
//     reference's GetEdgeCounts + mgpu::Scan + MarkPartitionSizes + mgpu::SortedSearch + D2H length read
//     (advance/kernel.cuh:300-368) disappear;
//   * each workgroup owns a CONTIGUOUS range of edge slots of equal size (perfect load balance even
//     when one R-MAT hub owns a million slots -- rows are split across tiles and workgroups, like
//     RelaxPartitionedEdges2's per-slot search, edge_map_partitioned/kernel.cuh:369-392);
//   * per tile the covering frontier slice (degree prefix, row start, vertex id) is staged in LDS with
//     coalesced loads; every staged entry marks the slot where its row begins and an inclusive max-scan
//     over the marks on the DPP path gives each slot its owner (ExpandTiles below; no search per slot), so
//     consecutive lanes read consecutive column_indices entries (256 B per wave-instruction);
//   * accepted destinations go through FrontierWriter (LDS staging, one packed global atomic per flush).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <climits>
#include <limits>
#include <type_traits>

#include <gunrock/oprtr/advance/binned.hpp>
#include <gunrock/oprtr/advance/functor_hooks.hpp>
#include <gunrock/oprtr/advance/sweep_chain.hpp>
#include <gunrock/oprtr/frontier_writer.hpp>
#include <gunrock/util/device_intrinsics.hpp>
#include <gunrock/util/error_utils.hpp>
#include <gunrock/util/frontier.hpp>
#include <gunrock/util/kernel_runtime_stats.hpp>

namespace gunrock {
namespace oprtr {
namespace advance {

// Names kept from advance/kernel_policy.cuh:43-75.  Only the forward modes are in scope.
enum MODE { TWC_FORWARD, TWC_BACKWARD, LB_BACKWARD, LB };
enum TYPE { V2V, V2E, E2V, E2E };
// Neighbour-list reduction of an advance (names from advance/kernel_policy.cuh:58-79): per input-frontier vertex, the
// REDUCE_OP of a value taken per out-edge -- at the edge's destination (VERTEX) or at the edge itself (EDGE).
enum REDUCE_OP { NONE, PLUS, MINUS, MULTIPLIES, MODULUS, BIT_OR, BIT_AND, BIT_XOR, MAXIMUM, MINIMUM };
enum REDUCE_TYPE { EMPTY, VERTEX, EDGE };

// Tuning surface reduced to what matters on CDNA4 (the reference's 14-integer policies are CUDA
// occupancy detail).
template <int _THREADS, int _ITEMS_PER_THREAD, int _MIN_BLOCKS_PER_CU, MODE _ADVANCE_MODE = LB>
struct KernelPolicy {
    static constexpr int THREADS = _THREADS;
    static constexpr int ITEMS = _ITEMS_PER_THREAD;
    static constexpr int TILE = THREADS * ITEMS;          // edge slots per tile
    static constexpr int MIN_BLOCKS = _MIN_BLOCKS_PER_CU;
    static constexpr int STAGE_CAPACITY = 2 * TILE;       // FrontierWriter staging entries
    static constexpr MODE ADVANCE_MODE = _ADVANCE_MODE;
};

template <typename VertexId, typename SizeT>
struct AdvanceArgs {
    util::Frontier<VertexId, SizeT> in;
    util::Frontier<VertexId, SizeT> out;
    SizeT in_len;                 // vertices in the input frontier
    SizeT in_edges;               // sum of their degrees (= scan total)
    const SizeT *d_row_offsets;
    const VertexId *d_column_indices;
    unsigned long long *d_tail_out;    // packed tail of the output frontier
    unsigned long long *d_tail_clear;  // ring slot to zero for the step after next
    int *d_overflow;
    BinPool<VertexId> bins;            // BINNED advance only (binned.hpp): where phase 1 hands its survivors
    unsigned long long *d_duty = nullptr;     // INSTRUMENT: this launch's runtime-stamp words (util/kernel_runtime_stats.hpp)
    const void *d_value_to_reduce = nullptr;  // reducing advance only (LaunchReduce): values indexed by vertex / by edge ...
    void *d_reduced_value = nullptr;          // ... and the per-frontier-entry results
};

// Frontier-array load.  FRESH = the arrays were written earlier in the SAME launch (multi-level tail kernel): read
// them at agent scope (global_load sc1, served by L2) so a line this CU's L1 cached on an earlier level cannot come
// back stale.
template <bool FRESH, typename T>
__device__ __forceinline__ T LoadQueue(const T *p)
{
    if (FRESH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

struct NoWriterStorage {};

// ---- reduction policy of an advance (reference: R_TYPE / R_OP of advance::LaunchKernel + moderngpu SegReduceCsr over the
//      scanned edge slots, advance/kernel.cuh:733-761, edge_map_partitioned/kernel.cuh:417-480) ----
struct NoReduce {
    static constexpr bool ENABLED = false;
};

template <REDUCE_OP OP, typename T>
struct ReduceOps {
    static __device__ __forceinline__ T Identity()
    {
        if (OP == MULTIPLIES) return static_cast<T>(1);
        if (OP == MAXIMUM) return std::numeric_limits<T>::lowest();
        if (OP == MINIMUM) return std::numeric_limits<T>::max();
        if (OP == BIT_AND) return static_cast<T>(~0ull);
        return static_cast<T>(0);  // PLUS, BIT_OR, BIT_XOR (and the operators without a reduction meaning)
    }
    static __device__ __forceinline__ T Combine(T a, T b)
    {
        if constexpr (OP == MULTIPLIES) return a * b;
        else if constexpr (OP == MAXIMUM) return a > b ? a : b;
        else if constexpr (OP == MINIMUM) return a < b ? a : b;
        else if constexpr (OP == BIT_OR || OP == BIT_AND || OP == BIT_XOR) {
            static_assert(!(OP == BIT_OR || OP == BIT_AND || OP == BIT_XOR) || std::is_integral<T>::value, "bitwise reductions need an integer type");
            if constexpr (OP == BIT_OR) return a | b;
            else if constexpr (OP == BIT_AND) return a & b;
            else return a ^ b;
        } else return a + b;
    }
    // combine `v` into *p (another workgroup or wave may hold the other part of the same neighbour list)
    static __device__ __forceinline__ void AtomicCombine(T *p, T v)
    {
        if constexpr (OP == PLUS || OP == NONE || OP == MINUS || OP == MODULUS) {
            atomicAdd(p, v);
        } else if constexpr (std::is_integral<T>::value && sizeof(T) == 8 && (OP == MAXIMUM || OP == MINIMUM)) {
            if constexpr (OP == MAXIMUM) atomicMax(reinterpret_cast<unsigned long long *>(p), static_cast<unsigned long long>(v));
            else atomicMin(reinterpret_cast<unsigned long long *>(p), static_cast<unsigned long long>(v));
        } else if constexpr (std::is_integral<T>::value && sizeof(T) == 4 && (OP == MAXIMUM || OP == MINIMUM || OP == BIT_OR || OP == BIT_AND || OP == BIT_XOR)) {
            if constexpr (OP == MAXIMUM) atomicMax(p, v);
            else if constexpr (OP == MINIMUM) atomicMin(p, v);
            else if constexpr (OP == BIT_OR) atomicOr(p, v);
            else if constexpr (OP == BIT_AND) atomicAnd(p, v);
            else atomicXor(p, v);
        } else {  // compare-and-swap loop on the value's bits
            static_assert(sizeof(T) == 4, "32-bit values");
            unsigned *q = reinterpret_cast<unsigned *>(p);
            unsigned seen = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                T cur;
                __builtin_memcpy(&cur, &seen, 4);
                const T want = Combine(cur, v);
                unsigned bits;
                __builtin_memcpy(&bits, &want, 4);
                if (bits == seen) break;
                const unsigned old = atomicCAS(q, seen, bits);
                if (old == seen) break;
                seen = old;
            }
        }
    }
};

// BY_VERTEX: the result of a frontier entry lands at d_reduced_value[its vertex id] instead of [its position in the frontier]
// (the reference indexes by position, advance/kernel.cuh:738; by vertex saves the scatter pass a vertex-indexed consumer needs)
template <REDUCE_TYPE _R_TYPE, REDUCE_OP _R_OP, typename _Value, bool _BY_VERTEX = false>
struct Reduce {
    static constexpr bool ENABLED = true;
    static constexpr bool BY_VERTEX = _BY_VERTEX;
    static constexpr REDUCE_TYPE R_TYPE = _R_TYPE;
    static constexpr REDUCE_OP R_OP = _R_OP;
    typedef _Value Value;
    typedef ReduceOps<_R_OP, _Value> Ops;
};

template <typename KernelPolicy, typename VertexId, typename SizeT, bool WITH_WRITER = true>
struct AdvanceShared {
    typedef FrontierWriter<KernelPolicy::THREADS, KernelPolicy::STAGE_CAPACITY, VertexId, SizeT> Writer;
    static constexpr int WAVES = KernelPolicy::THREADS / util::kWaveSize;
    SizeT scan[KernelPolicy::THREADS];       // stage: degree prefix relative to the tile's first slot
    SizeT row[KernelPolicy::THREADS];        // stage: first edge of the vertex
    VertexId vertex[KernelPolicy::THREADS];  // stage: vertex id
    unsigned source_data[KernelPolicy::THREADS];  // stage: Functor::SourceData of the vertex (functor_hooks.hpp, staged hooks)
    unsigned own[KernelPolicy::TILE];        // slot -> (tile tag << IDX_BITS | staged entry whose row starts at this slot)
    typename std::conditional<WITH_WRITER, typename Writer::Storage, NoWriterStorage>::type writer;
    int owner_count[2][WAVES];
    unsigned long long level_tail;        // tail kernel: broadcast of the level's packed tail
    unsigned long long wave_sum[WAVES];   // COUNT_ONLY reduction
    BinnerStorage binner;                 // BINNED: the workgroup's open chunks
};

constexpr int ILog2(int x) { return x <= 1 ? 0 : 1 + ILog2(x >> 1); }

// Per-kernel state of the slot -> owner marks (see ExpandTiles): zero the marks once, then every tile uses a fresh tag.
// Whole workgroup; a barrier must follow before the first ExpandTiles.
template <typename KernelPolicy, typename Shared>
__device__ __forceinline__ void InitOwnerMarks(Shared &sh, unsigned &tile_tag)
{
#pragma unroll
    for (int k = 0; k < KernelPolicy::ITEMS; ++k) sh.own[k * KernelPolicy::THREADS + threadIdx.x] = 0u;
    tile_tag = 0u;
}

// Expand the edge-slot tiles [tile_begin, tile_end) of the input frontier.  Whole workgroup calls; requires the writer
// initialised, InitOwnerMarks done and a barrier since.  On return all appends are complete and a barrier has passed (count
// is stable); nothing has been flushed beyond what overflow protection forced.
// OUT_WITH_DEGREES: the output is a full frontier (vertex, row start, degree prefix) ready for the next advance
// (BFS); false = ids only, for outputs that pass through a filter / priority-queue split first (SSSP).
// COUNT_ONLY: accepted destinations are only counted (into `accepted`), nothing is enqueued -- for a level whose output
// frontier will be consumed as a bitmap (the bottom-up direction) and needs neither ids nor degrees.
// BINNED (with COUNT_ONLY): phase 1 of the destination-binned advance (binned.hpp) -- edges that pass ScreenEdge are handed to
// the bin of their destination's owner XCD; CondEdge / ApplyEdge run in phase 2.  The caller has initialised sh.binner.
//
// Slot -> owner without a search per slot.  The reference binary-searches the staged degree prefix for every output slot
// (edge_map_partitioned/kernel.cuh:369-392); on a low-degree frontier (R-MAT level 2: 6.6 M vertices of degree ~22) that is
// 6 dependent LDS round trips per edge and it bounded the kernel.  Here the thread that stages frontier entry j also drops
// the mark (tag | j) at the slot where j's row begins; a wave owns ITEMS x 64 CONSECUTIVE slots, finds the owner of its
// first slot with one (wave-uniform) search and resolves the rest with an inclusive max-scan over the marks on the DPP
// path.  Tags make stale marks of earlier tiles lose every max, so the marks are never cleared.
// Reducer = Reduce<R_TYPE, R_OP, Value> (with COUNT_ONLY): besides CondEdge / ApplyEdge, the edge slots of every frontier entry
// are reduced into a.d_reduced_value[entry] (pre-set to the operator's identity): a wave-segmented scan over each run of
// consecutive slots with the same owner, then ONE plain store when the run is the entry's whole neighbour list, else one atomic.
template <typename KernelPolicy, typename ProblemData, typename Functor, bool OUT_WITH_DEGREES, bool FRESH, bool COUNT_ONLY = false,
          bool BINNED = false, typename Reducer = NoReduce, typename Shared>
__device__ __forceinline__ void ExpandTiles(
    const AdvanceArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> &a, typename ProblemData::DataSlice &slice,
    const long long tile_begin, const long long tile_end, Shared &sh, unsigned &accepted, unsigned &tile_tag)
{
    typedef typename ProblemData::VertexId VertexId;
    typedef typename ProblemData::SizeT SizeT;
    constexpr int THREADS = KernelPolicy::THREADS;
    constexpr int TILE = KernelPolicy::TILE;
    constexpr int ITEMS = KernelPolicy::ITEMS;
    constexpr int IDX_BITS = ILog2(THREADS);
    static_assert((1 << IDX_BITS) == THREADS, "workgroup size must be a power of two");
    constexpr unsigned MAX_TAG = (1u << (32 - IDX_BITS)) - 1u;
    typedef FrontierWriter<THREADS, KernelPolicy::STAGE_CAPACITY, VertexId, SizeT> Writer;
    const int tid = threadIdx.x;
    const unsigned lane = util::LaneId();
    const int wave_base = __builtin_amdgcn_readfirstlane(tid / util::kWaveSize) * (util::kWaveSize * ITEMS);  // first slot of this wave
    const long long total = a.in_edges;

    // Frontier cursor: largest i with scan[i] <= first slot of this range (zero-degree vertices are never enqueued,
    // so the prefix is strictly increasing and the owner is unique).
    SizeT cursor = 0;
    if (tile_begin > 0) {  // the range that starts at slot 0 starts at frontier entry 0: no search (every level of the tail kernel)
        const SizeT first_slot = static_cast<SizeT>(tile_begin * TILE);
        // 64-ary search, one probe per lane and round: a frontier of 3 M entries takes 4 dependent round trips instead of the
        // 22 of a binary search (every workgroup pays this chain before its first tile; it was ~10 us of a 46 us launch).
        SizeT lo = 0, hi = a.in_len;  // invariant: scan[lo] <= first_slot < scan[hi] (scan[in_len] = total)
        while (hi - lo > 1) {
            const SizeT step = (hi - lo + util::kWaveSize - 1) / util::kWaveSize;
            const long long idx = static_cast<long long>(lo) + static_cast<long long>(lane) * step;
            const bool le = idx < hi && LoadQueue<FRESH>(a.in.scan + idx) <= first_slot;
            const int last = __popcll(__ballot(le)) - 1;  // the prefix is increasing: the lanes that hold are lanes 0..last
            const long long next_hi = static_cast<long long>(lo) + static_cast<long long>(last + 1) * step;
            lo = static_cast<SizeT>(lo + static_cast<long long>(last) * step);
            if (next_hi < hi) hi = static_cast<SizeT>(next_hi);
        }
        cursor = lo;
    }

    // A tile = up to TILE consecutive edge slots whose rows fit the stage (THREADS frontier entries): when more rows than that
    // begin inside the TILE slots (rows of degree < ITEMS on average), the tile ends where the last staged row begins.
    // The stage of the NEXT tile is fetched while this one expands (row start and vertex together with the prefix: one round
    // trip, not two).
    const long long slot_end = (tile_end * TILE < total) ? tile_end * TILE : total;
    long long slot_at = tile_begin * TILE;
    SizeT p_scan = INT_MAX, p_row = 0;
    VertexId p_v = 0;
    constexpr bool STAGED = HasSourceData<Functor, VertexId, typename ProblemData::DataSlice>::value;
    typedef typename EdgeStateOf<Functor, STAGED>::type EdgeState;
    unsigned p_source = 0u;
    if (slot_at < slot_end && cursor + tid < a.in_len) {
        p_scan = LoadQueue<FRESH>(a.in.scan + cursor + tid);
        p_row = LoadQueue<FRESH>(a.in.row_start + cursor + tid);
        p_v = LoadQueue<FRESH>(a.in.v + cursor + tid);
        if constexpr (STAGED) p_source = Functor::SourceData(p_v, &slice);
    }

    while (slot_at < slot_end) {
        const SizeT slot0 = static_cast<SizeT>(slot_at);
        const int limit = (slot_end - slot_at < TILE) ? static_cast<int>(slot_end - slot_at) : TILE;
        if (++tile_tag > MAX_TAG) {  // (uniform) tags exhausted: forget every mark and start over
            __syncthreads();
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) sh.own[k * THREADS + tid] = 0u;
            tile_tag = 1u;
            __syncthreads();
        }
        const unsigned tag = tile_tag << IDX_BITS;

        // Appends of the previous tile are complete (barrier at the end of the loop body / after the
        // cursor search); every thread reads the count here, before this tile's first barrier, and the
        // next Append comes after it.
        int pending = 0;
        if constexpr (!COUNT_ONLY) pending = Writer::Count(sh.writer);

        // ---- stage the covering frontier slice ----
        // sh.scan holds the prefix relative to the tile (<= 0 for entry 0, whose row may have begun in an earlier tile;
        // INT_MAX past the frontier).  Entries with a prefix below `limit` are candidates to own slots of this tile (a prefix of
        // the stage, because the degree prefix is increasing); every one but entry 0 marks the slot where its row begins.
        {
            const SizeT rel = (p_scan == INT_MAX) ? INT_MAX : p_scan - slot0;
            sh.scan[tid] = rel;
            if (rel < limit) {
                sh.row[tid] = p_row;
                sh.vertex[tid] = p_v;
                if constexpr (STAGED) sh.source_data[tid] = p_source;
                if (rel > 0) sh.own[rel] = tag | static_cast<unsigned>(tid);
            }
            const unsigned long long in_tile = __ballot(rel < limit);
            if (lane == 0) sh.owner_count[tile_tag & 1][tid / util::kWaveSize] = __popcll(in_tile);
        }
        if (FRESH) __syncthreads();
        else util::LdsBarrier();  // (orders LDS only: the previous tile's global stores stay in flight)
        int owners = 0;
#pragma unroll
        for (int w = 0; w < THREADS / util::kWaveSize; ++w) owners += sh.owner_count[tile_tag & 1][w];
        int slots = limit;
        if (owners == THREADS) {  // the stage is full of rows that begin inside the tile: cut the tile at the last one
            slots = sh.scan[THREADS - 1];  // >= THREADS - 1 because degrees are >= 1: progress is guaranteed
            owners = THREADS - 1;
        }
        // where the next tile starts: at this tile's last owner, or at the staged entry right behind it when that entry's row
        // begins exactly at the tile's end
        int advance = owners - 1;
        if (sh.scan[owners] == slots) advance = owners;  // (owners < THREADS here)
        {  // next tile's stage: in flight while this tile expands
            const SizeT idx = cursor + advance + tid;
            p_scan = INT_MAX;
            if (slot_at + slots < slot_end && idx < a.in_len) {
                p_scan = LoadQueue<FRESH>(a.in.scan + idx);
                p_row = LoadQueue<FRESH>(a.in.row_start + idx);
                p_v = LoadQueue<FRESH>(a.in.v + idx);
            }
        }
        if constexpr (!COUNT_ONLY) {
            if (pending > KernelPolicy::STAGE_CAPACITY - TILE) {
                if (OUT_WITH_DEGREES) Writer::template Flush<true>(sh.writer, pending, a.out, a.d_tail_out, a.d_overflow, a.d_row_offsets);
                else Writer::FlushIds(sh.writer, pending, a.out.v, a.out.capacity, a.d_tail_out, a.d_overflow);
            }
        }

        // ---- slot -> owner: one wave-uniform search for the wave's first slot, then a max-scan over the marks ----
        SizeT edge[ITEMS];
        VertexId src[ITEMS];
        VertexId dst[ITEMS];
        bool live[ITEMS];
        int own[Reducer::ENABLED ? ITEMS : 1];  // staged entry that owns the slot (reducing advance only)
        unsigned source[STAGED ? ITEMS : 1];
        EdgeState state[STAGED ? ITEMS : 1];
        {
            int lo = 0, hi = owners;  // sh.scan[lo] <= wave_base < sh.scan[hi] (hi == owners: past the slice)
            if (wave_base > 0) {
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (sh.scan[mid] <= wave_base) lo = mid; else hi = mid;
                }
            }
            unsigned carry = tag | static_cast<unsigned>(lo);
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) {
                const int slot = wave_base + k * util::kWaveSize + static_cast<int>(lane);
                live[k] = slot < slots;
                unsigned m = sh.own[slot];
                if (lane == 0) m = max(m, carry);
                m = util::WaveInclusiveMaxDpp(m);
                carry = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(m), util::kWaveSize - 1));
                const int owner = static_cast<int>(m & static_cast<unsigned>(THREADS - 1));
                edge[k] = sh.row[owner] + (slot - sh.scan[owner]);
                src[k] = sh.vertex[owner];
                if constexpr (STAGED) source[k] = sh.source_data[owner];
                if constexpr (Reducer::ENABLED) own[k] = owner;
            }
        }
        // ---- expand, phase by phase so each thread keeps ITEMS independent memory operations in flight ----
        // consecutive lanes hold consecutive slots => one wave-instruction reads 256 contiguous bytes of column_indices
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {  // (a dead slot reads entry 0: no branch around the load)
            if (!live[k]) { edge[k] = 0; src[k] = 0; }
            dst[k] = a.d_column_indices[edge[k]];
        }
        // side-effect-free screen, evaluated for EVERY slot (dead ones with vertex 0, edge 0) and combined without a branch, so
        // the status loads of a tile are in flight together: `live && Screen()` would wait for each load before the next
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            bool pass;
            if constexpr (STAGED)
                pass = Functor::ScreenEdge(src[k], dst[k], &slice, edge[k], slot0 + wave_base + k * util::kWaveSize + static_cast<int>(lane),
                                           source[k], state[k]);
            else
                pass = ScreenEdge<Functor>(src[k], dst[k], &slice, edge[k], slot0 + wave_base + k * util::kWaveSize + static_cast<int>(lane));
            live[k] = live[k] & pass;
        }
        if constexpr (STAGED) {  // the next tile's source data: its vertex ids (fetched above) have arrived with the screen's loads
            p_source = 0u;
            if (p_scan != INT_MAX) p_source = Functor::SourceData(p_v, &slice);
        }
        int mine = 0;
        if constexpr (BINNED) {
            static_assert(!BINNED || COUNT_ONLY, "a binned advance enqueues nothing itself");
            Binner<THREADS, ITEMS, VertexId, ProblemData::MARK_PREDECESSORS>::Put(sh.binner, a.bins, src, dst, live);
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) mine += live[k] ? 1 : 0;
        } else {
            if constexpr (STAGED) {
                // survivors pay the (atomic) claim: all of them issued, then all of them examined; what the screen computed
                // (state) and the staged source data come along, so the claim loads nothing
                typedef decltype(Functor::IssueEdge(src[0], dst[0], &slice, edge[0], slot0, source[0], state[0])) Token;
                Token token[ITEMS];
#pragma unroll
                for (int k = 0; k < ITEMS; ++k) {
                    token[k] = Token();
                    if (live[k]) token[k] = Functor::IssueEdge(src[k], dst[k], &slice, edge[k],
                                                               slot0 + wave_base + k * util::kWaveSize + static_cast<int>(lane), source[k], state[k]);
                }
#pragma unroll
                for (int k = 0; k < ITEMS; ++k)
                    live[k] = live[k] && Functor::ResolveEdge(token[k], src[k], dst[k], &slice, edge[k],
                                                              slot0 + wave_base + k * util::kWaveSize + static_cast<int>(lane), state[k]);
            } else if constexpr (HasIssueEdge<Functor, VertexId, typename ProblemData::DataSlice>::value) {
                // survivors pay the (atomic) claim: all of them issued, then all of them examined
                typedef decltype(Functor::IssueEdge(src[0], dst[0], &slice, edge[0], slot0)) Token;
                Token token[ITEMS];
#pragma unroll
                for (int k = 0; k < ITEMS; ++k) {
                    token[k] = Token();
                    if (live[k]) token[k] = Functor::IssueEdge(src[k], dst[k], &slice, edge[k],
                                                               slot0 + wave_base + k * util::kWaveSize + static_cast<int>(lane));
                }
#pragma unroll
                for (int k = 0; k < ITEMS; ++k)
                    live[k] = live[k] && Functor::ResolveEdge(token[k], src[k], dst[k], &slice, edge[k],
                                                              slot0 + wave_base + k * util::kWaveSize + static_cast<int>(lane));
            } else {
#pragma unroll
                for (int k = 0; k < ITEMS; ++k)  // survivors pay the (atomic) claim
                    live[k] = live[k] && Functor::CondEdge(src[k], dst[k], &slice, edge[k],
                                                           slot0 + wave_base + k * util::kWaveSize + static_cast<int>(lane));
            }
            if constexpr (HasApplyEdgeWave<Functor, VertexId, typename ProblemData::DataSlice>::value) {
#pragma unroll
                for (int k = 0; k < ITEMS; ++k) {  // every lane calls: the functor works across the wave
                    Functor::ApplyEdgeWave(src[k], dst[k], live[k], &slice, edge[k],
                                           slot0 + wave_base + k * util::kWaveSize + static_cast<int>(lane));
                    mine += live[k] ? 1 : 0;
                }
            } else {
#pragma unroll
                for (int k = 0; k < ITEMS; ++k) {
                    if (live[k]) {
                        Functor::ApplyEdge(src[k], dst[k], &slice, edge[k],
                                           slot0 + wave_base + k * util::kWaveSize + static_cast<int>(lane));
                        ++mine;
                    }
                }
            }
        }
        if constexpr (Reducer::ENABLED) {
            typedef typename Reducer::Value RValue;
            typedef typename Reducer::Ops Ops;
            const RValue *values = static_cast<const RValue *>(a.d_value_to_reduce);
            RValue *reduced = static_cast<RValue *>(a.d_reduced_value);
            RValue val[ITEMS];
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) {  // all value loads in flight together (rejected slots read entry 0, then take the identity)
                RValue got;
                if constexpr (HasReduceValue<Functor, VertexId, typename ProblemData::DataSlice>::value)
                    got = Functor::ReduceValue(live[k] ? src[k] : static_cast<VertexId>(0), live[k] ? dst[k] : static_cast<VertexId>(0), &slice,
                                               live[k] ? edge[k] : static_cast<SizeT>(0), 0);
                else
                    got = values[live[k] ? (Reducer::R_TYPE == VERTEX ? static_cast<SizeT>(dst[k]) : edge[k]) : static_cast<SizeT>(0)];
                val[k] = live[k] ? got : Ops::Identity();
            }
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) {
                const int row_first = wave_base + k * util::kWaveSize;
                const int slot = row_first + static_cast<int>(lane);
                const int key = slot < slots ? own[k] : -1;  // owners are non-decreasing along the slots: equal keys are one run
                // segment heads: the key changes (dead slots, key -1, form their own segment and report nothing)
                const int prev_key = static_cast<int>(util::DppFrom<0x138, 0xF>(static_cast<unsigned>(key) ^ 1u, static_cast<unsigned>(key)));  // wave_shr:1
                const bool head = lane == 0 || prev_key != key;
                const RValue v = util::WaveSegmentedScanDpp<Ops, RValue>(head, val[k]);
                const int next_key = __shfl_down(key, 1, util::kWaveSize);
                if (key >= 0 && (lane == util::kWaveSize - 1 || next_key != key)) {  // last slot of the run
                    const int row_begin = sh.scan[key];
                    const int row_end = key + 1 < THREADS ? sh.scan[key + 1] : INT_MAX;  // (INT_MAX past the frontier too: the last entry's
                                                                                        //  list ends with the slots)
                    const bool ends_here = row_end - 1 == slot || (row_end == INT_MAX && key + 1 < THREADS && slot == slots - 1 &&
                                                                  slot_at + slots >= total);
                    const long long at = Reducer::BY_VERTEX ? static_cast<long long>(sh.vertex[key]) : static_cast<long long>(cursor) + key;
                    if (row_begin >= row_first && ends_here) reduced[at] = v;   // the whole neighbour list sat in this run
                    else Ops::AtomicCombine(reduced + at, v);
                }
            }
        }
        accepted += static_cast<unsigned>(mine);
        if constexpr (!COUNT_ONLY) {  // one LDS reservation per wave per tile
            int pos = Writer::Reserve(sh.writer, mine);
#pragma unroll
            for (int k = 0; k < ITEMS; ++k)
                if (live[k]) sh.writer.buf[pos++] = dst[k];
        }

        if (FRESH) __syncthreads();  // every wave is done with the staged slice (and its appends are complete)
        else util::LdsBarrier();     // same, without waiting for this tile's global stores (labels, binned pairs)
        cursor += advance;
        slot_at += slots;
        if constexpr (BINNED)  // chunks that filled up are rotated; the next tile's staging barrier orders this before its Put
            Binner<THREADS, ITEMS, VertexId, ProblemData::MARK_PREDECESSORS>::EndTile(sh.binner, a.bins);
    }
}

template <typename KernelPolicy, typename ProblemData, typename Functor, bool OUT_WITH_DEGREES = true, bool COUNT_ONLY = false>
__global__ __launch_bounds__(KernelPolicy::THREADS) void LoadBalancedKernel(
    AdvanceArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> a,
    typename ProblemData::DataSlice slice)
{
    typedef typename ProblemData::VertexId VertexId;
    typedef typename ProblemData::SizeT SizeT;
    typedef AdvanceShared<KernelPolicy, VertexId, SizeT, !COUNT_ONLY> Shared;
    typedef typename Shared::Writer Writer;
    __shared__ Shared sh;
    util::DutyStamp duty(a.d_duty);

    if (blockIdx.x == 0 && threadIdx.x == 0 && a.d_tail_clear) *a.d_tail_clear = 0ull;
    if constexpr (!COUNT_ONLY) Writer::Init(sh.writer);
    unsigned tile_tag;
    InitOwnerMarks<KernelPolicy>(sh, tile_tag);

    const long long tiles = (static_cast<long long>(a.in_edges) + KernelPolicy::TILE - 1) / KernelPolicy::TILE;
    const long long per_block = (tiles + gridDim.x - 1) / gridDim.x;
    const long long tile_begin = static_cast<long long>(blockIdx.x) * per_block;
    const long long tile_end = (tile_begin + per_block < tiles) ? tile_begin + per_block : tiles;
    if (tile_begin >= tile_end) return;  // workgroup-uniform, nothing staged
    __syncthreads();                     // writer count initialised

    unsigned accepted = 0;
    ExpandTiles<KernelPolicy, ProblemData, Functor, OUT_WITH_DEGREES, false, COUNT_ONLY>(a, slice, tile_begin, tile_end, sh, accepted, tile_tag);

    if constexpr (COUNT_ONLY) {  // workgroup total -> one atomic on the packed tail (edge half stays 0)
        unsigned long long sum = util::WaveSum(static_cast<unsigned long long>(accepted));
        if ((threadIdx.x & (util::kWaveSize - 1)) == 0) sh.wave_sum[threadIdx.x / util::kWaveSize] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long total = 0;
#pragma unroll
            for (int w = 0; w < KernelPolicy::THREADS / util::kWaveSize; ++w) total += sh.wave_sum[w];
            if (total && a.d_tail_out) atomicAdd(a.d_tail_out, total);  // (nullptr: the caller does not want the count)
        }
    } else {
        // final flush (ExpandTiles ended on a barrier: all appends complete, count is stable)
        const int rest = Writer::Count(sh.writer);
        __syncthreads();
        if (OUT_WITH_DEGREES) Writer::template Flush<true>(sh.writer, rest, a.out, a.d_tail_out, a.d_overflow, a.d_row_offsets);
        else Writer::FlushIds(sh.writer, rest, a.out.v, a.out.capacity, a.d_tail_out, a.d_overflow);
    }
}

// ---- phase 1 of the destination-binned advance (binned.hpp): expand + ScreenEdge, survivors go to their owner XCD's bin ----
template <typename KernelPolicy, typename ProblemData, typename Functor>
__global__ __launch_bounds__(KernelPolicy::THREADS) void BinnedExpandKernel(
    AdvanceArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> a,
    typename ProblemData::DataSlice slice)
{
    typedef typename ProblemData::VertexId VertexId;
    typedef typename ProblemData::SizeT SizeT;
    typedef AdvanceShared<KernelPolicy, VertexId, SizeT, false> Shared;
    typedef Binner<KernelPolicy::THREADS, KernelPolicy::ITEMS, VertexId, ProblemData::MARK_PREDECESSORS> Bins;
    __shared__ Shared sh;
    util::DutyStamp duty(a.d_duty);
    unsigned tile_tag;
    InitOwnerMarks<KernelPolicy>(sh, tile_tag);

    if (blockIdx.x == 0 && threadIdx.x == 0 && a.d_tail_clear) *a.d_tail_clear = 0ull;
    const long long tiles = (static_cast<long long>(a.in_edges) + KernelPolicy::TILE - 1) / KernelPolicy::TILE;
    const long long per_block = (tiles + gridDim.x - 1) / gridDim.x;
    const long long tile_begin = static_cast<long long>(blockIdx.x) * per_block;
    const long long tile_end = (tile_begin + per_block < tiles) ? tile_begin + per_block : tiles;
    if (tile_begin >= tile_end) return;  // workgroup-uniform: no chunk taken
    BinnerStorage &bins = sh.binner;
    Bins::Init(bins, a.bins);
    __syncthreads();

    unsigned accepted = 0;
    ExpandTiles<KernelPolicy, ProblemData, Functor, false, false, true, true>(a, slice, tile_begin, tile_end, sh, accepted, tile_tag);
    Bins::Finish(bins, a.bins);  // (ExpandTiles ended on a barrier + EndTile by the same threads)
}

template <typename KernelPolicy, typename ProblemData, typename Functor>
hipError_t LaunchBinned(const AdvanceArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> &args,
                        const typename ProblemData::DataSlice &slice, const typename ProblemData::DataSlice &apply_slice,
                        int max_grid_size, int apply_grid, hipStream_t stream)
{
    if (args.in_len <= 0 || args.in_edges <= 0) return hipSuccess;
    const long long tiles = (static_cast<long long>(args.in_edges) + KernelPolicy::TILE - 1) / KernelPolicy::TILE;
    long long grid = tiles < max_grid_size ? tiles : max_grid_size;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((BinnedExpandKernel<KernelPolicy, ProblemData, Functor>), dim3(static_cast<unsigned>(grid)),
                       dim3(KernelPolicy::THREADS), 0, stream, args, slice);
    hipError_t rc = util::GRError("advance::BinnedExpandKernel launch failed", __FILE__, __LINE__);
    if (rc) return rc;
    hipLaunchKernelGGL((BinnedApplyKernel<256, ProblemData, Functor, ProblemData::MARK_PREDECESSORS>), dim3(apply_grid), dim3(256), 0,
                       stream, args.bins, apply_slice);
    return util::GRError("advance::BinnedApplyKernel launch failed", __FILE__, __LINE__);
}

// ---- multi-level tail: ONE workgroup runs consecutive BSP levels while the frontier stays small ----
// A level of a few thousand edges costs ~10 us as a kernel plus a ~20 us host round trip for its length; the last
// levels of an R-MAT search (and every level of a road network) are that small.  Here a single workgroup keeps
// expanding level after level: the next frontier comes back through the same FrontierWriter, the level's packed tail
// is read at agent scope, and the host is involved again only when the frontier is empty, outgrows `edge_limit`, or
// `max_levels` levels have run.  Queue arrays written on one level are read on the next with L1-bypassing loads after
// every wave has drained its stores (s_waitcnt vmcnt(0)) and the workgroup has met at a barrier.
template <typename VertexId, typename SizeT>
struct TailArgs {
    util::Frontier<VertexId, SizeT> queue[2];
    int selector;                      // queue[selector] is the input of the first level
    long long first_iteration;         // BSP iteration number of the first level
    unsigned long long *d_tail;        // ring of 4 packed tails; slot (iteration & 3) holds the input frontier's
    SizeT edge_limit;                  // leave when a level has more edge slots than this
    int max_levels;
    int *d_levels_done;                // out: levels executed
    unsigned long long *d_level_sums;  // out: [0] sum of frontier lengths, [1] sum of frontier edge counts over those levels
    const SizeT *d_row_offsets;
    const VertexId *d_column_indices;
    int *d_overflow;
};

template <typename KernelPolicy, typename ProblemData, typename Functor>
__global__ __launch_bounds__(KernelPolicy::THREADS) void TailLevelsKernel(
    TailArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> t, typename ProblemData::DataSlice slice)
{
    typedef typename ProblemData::VertexId VertexId;
    typedef typename ProblemData::SizeT SizeT;
    typedef AdvanceShared<KernelPolicy, VertexId, SizeT> Shared;
    typedef typename Shared::Writer Writer;
    __shared__ Shared sh;

    Writer::Init(sh.writer);
    unsigned tile_tag;
    InitOwnerMarks<KernelPolicy>(sh, tile_tag);
    int selector = t.selector;
    long long iteration = t.first_iteration;
    int done = 0;
    unsigned long long sum_len = 0, sum_edges = 0;
    for (;;) {
        if (threadIdx.x == 0) {
            sh.level_tail = __hip_atomic_load(t.d_tail + (iteration & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(t.d_tail + ((iteration + 2) & 3), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const unsigned long long tail = sh.level_tail;
        const SizeT len = static_cast<SizeT>(util::TailCount(tail));
        const SizeT edges = static_cast<SizeT>(util::TailEdges(tail));
        if (len == 0 || edges > t.edge_limit || done >= t.max_levels) break;  // workgroup-uniform

        AdvanceArgs<VertexId, SizeT> a;
        a.in = t.queue[selector];
        a.out = t.queue[selector ^ 1];
        a.in_len = len;
        a.in_edges = edges;
        a.d_row_offsets = t.d_row_offsets;
        a.d_column_indices = t.d_column_indices;
        a.d_tail_out = t.d_tail + ((iteration + 1) & 3);
        a.d_tail_clear = nullptr;
        a.d_overflow = t.d_overflow;
        slice.iteration = static_cast<VertexId>(iteration);

        const long long tiles = (static_cast<long long>(edges) + KernelPolicy::TILE - 1) / KernelPolicy::TILE;
        unsigned accepted = 0;
        ExpandTiles<KernelPolicy, ProblemData, Functor, true, true>(a, slice, 0, tiles, sh, accepted, tile_tag);
        const int rest = Writer::Count(sh.writer);
        __syncthreads();
        Writer::template Flush<true>(sh.writer, rest, a.out, a.d_tail_out, a.d_overflow, a.d_row_offsets);
        // every wave's queue stores (and the flush's tail atomic) must have reached L2 before the next level reads them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        selector ^= 1;
        ++iteration;
        ++done;
        sum_len += len;
        sum_edges += edges;
    }
    if (threadIdx.x == 0) {
        *t.d_levels_done = done;
        t.d_level_sums[0] = sum_len;
        t.d_level_sums[1] = sum_edges;
    }
}

// Host-side launch.  Mirrors advance::LaunchKernel (advance/kernel.cuh:101-129) at the distilled level of
// input/output frontier, graph, problem data, traversal type.
template <typename KernelPolicy, typename ProblemData, typename Functor, bool OUT_WITH_DEGREES = true, bool COUNT_ONLY = false>
hipError_t LaunchKernel(const AdvanceArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> &args,
                        const typename ProblemData::DataSlice &slice, int max_grid_size, hipStream_t stream,
                        TYPE /*ADVANCE_TYPE: only V2V is on the BFS/SSSP path*/ = V2V)
{
    if (args.in_len <= 0 || args.in_edges <= 0) return hipSuccess;
    const long long tiles = (static_cast<long long>(args.in_edges) + KernelPolicy::TILE - 1) / KernelPolicy::TILE;
    if (max_grid_size <= 0)  // one resident wave of workgroups, each with a contiguous share of the tiles
        max_grid_size = util::ResidentGrid(LoadBalancedKernel<KernelPolicy, ProblemData, Functor, OUT_WITH_DEGREES, COUNT_ONLY>, KernelPolicy::THREADS);
    long long grid = tiles < max_grid_size ? tiles : max_grid_size;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((LoadBalancedKernel<KernelPolicy, ProblemData, Functor, OUT_WITH_DEGREES, COUNT_ONLY>), dim3(static_cast<unsigned>(grid)),
                       dim3(KernelPolicy::THREADS), 0, stream, args, slice);
    return util::GRError("advance::LoadBalancedKernel launch failed", __FILE__, __LINE__);
}

// ---- reducing advance: CondEdge / ApplyEdge as usual, plus the per-frontier-entry reduction of a value per edge ----
template <typename KernelPolicy, typename ProblemData, typename Functor, typename Reducer>
__global__ __launch_bounds__(KernelPolicy::THREADS) void ReduceKernel(
    AdvanceArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> a, typename ProblemData::DataSlice slice)
{
    typedef typename ProblemData::VertexId VertexId;
    typedef typename ProblemData::SizeT SizeT;
    typedef AdvanceShared<KernelPolicy, VertexId, SizeT, false> Shared;
    __shared__ Shared sh;
    unsigned tile_tag;
    InitOwnerMarks<KernelPolicy>(sh, tile_tag);
    const long long tiles = (static_cast<long long>(a.in_edges) + KernelPolicy::TILE - 1) / KernelPolicy::TILE;
    const long long per_block = (tiles + gridDim.x - 1) / gridDim.x;
    const long long tile_begin = static_cast<long long>(blockIdx.x) * per_block;
    const long long tile_end = (tile_begin + per_block < tiles) ? tile_begin + per_block : tiles;
    if (tile_begin >= tile_end) return;
    __syncthreads();
    unsigned accepted = 0;
    ExpandTiles<KernelPolicy, ProblemData, Functor, false, false, true, false, Reducer>(a, slice, tile_begin, tile_end, sh, accepted, tile_tag);
}

template <typename T>
__global__ void FillKernel(T *d_out, T value, long long n)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) d_out[i] = value;
}

template <REDUCE_OP OP, typename T>
inline T HostIdentity()
{
    if (OP == MULTIPLIES) return static_cast<T>(1);
    if (OP == MAXIMUM) return std::numeric_limits<T>::lowest();
    if (OP == MINIMUM) return std::numeric_limits<T>::max();
    if (OP == BIT_AND) return static_cast<T>(~0ull);
    return static_cast<T>(0);
}

// advance::LaunchKernel with R_TYPE / R_OP (advance/kernel.cuh:101-129): d_reduced_value[i] = R_OP over the out-edges e = (v, u)
// of input-frontier entry i = (v) that pass CondEdge of d_value_to_reduce[u] (VERTEX) or d_value_to_reduce[e] (EDGE); entries
// whose every edge fails get the operator's identity.  ApplyEdge runs for the passing edges as in a plain advance; nothing is
// enqueued.  One launch (plus the identity fill): no device-wide segmented-reduce pass over a materialised edge-value array.
// BY_VERTEX: results indexed by vertex id; `out_len` = entries of d_reduced_value to pre-set (0 = the frontier length).
// prefill = false: the caller guarantees that the output entries of this frontier already hold the operator's identity (BC's
// dependency sums: zero since Reset, every vertex is reduced into once) -- the other entries of the array are left alone.
template <typename KernelPolicy, typename ProblemData, typename Functor, REDUCE_TYPE R_TYPE, REDUCE_OP R_OP, typename Value, bool BY_VERTEX = false>
hipError_t LaunchReduce(AdvanceArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> args,
                        const typename ProblemData::DataSlice &slice, const Value *d_value_to_reduce, Value *d_reduced_value,
                        int max_grid_size, hipStream_t stream, long long out_len = 0, bool prefill = true)
{
    static_assert(R_TYPE != EMPTY && R_OP != NONE, "a reducing advance needs a reduction");
    if (out_len <= 0) out_len = args.in_len;
    if (out_len <= 0) return hipSuccess;
    hipError_t rc = hipSuccess;
    if (prefill) {
        hipLaunchKernelGGL((FillKernel<Value>), dim3(static_cast<unsigned>((out_len + 1023) / 1024 < 2048 ? (out_len + 1023) / 1024 : 2048)),
                           dim3(256), 0, stream, d_reduced_value, HostIdentity<R_OP, Value>(), out_len);
        rc = util::GRError("advance::FillKernel launch failed", __FILE__, __LINE__);
    }
    if (rc || args.in_edges <= 0 || args.in_len <= 0) return rc;
    args.d_value_to_reduce = d_value_to_reduce;
    args.d_reduced_value = d_reduced_value;
    typedef Reduce<R_TYPE, R_OP, Value, BY_VERTEX> Reducer;
    const long long tiles = (static_cast<long long>(args.in_edges) + KernelPolicy::TILE - 1) / KernelPolicy::TILE;
    if (max_grid_size <= 0) max_grid_size = util::ResidentGrid(ReduceKernel<KernelPolicy, ProblemData, Functor, Reducer>, KernelPolicy::THREADS);
    long long grid = tiles < max_grid_size ? tiles : max_grid_size;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((ReduceKernel<KernelPolicy, ProblemData, Functor, Reducer>), dim3(static_cast<unsigned>(grid)),
                       dim3(KernelPolicy::THREADS), 0, stream, args, slice);
    return util::GRError("advance::ReduceKernel launch failed", __FILE__, __LINE__);
}

// ---- persistent multi-workgroup levels: the same idea for MID-SIZE frontiers (8 K .. ~1 M edge slots) ----
// A 2048 x 2048 grid graph runs 2049 levels of ~16 K edges: as separate launches each costs ~20 us of kernel plus ~40 us of
// host round trip (62 ms per search).  Here `gridDim.x` resident workgroups (one per CU at most) stay in the kernel, split
// every level's edge slots evenly, and meet at a grid barrier between levels.  The barrier is the counter form of the
// programming guide's inter-workgroup recipe: every workgroup drains its stores, one lane issues an agent-scope release,
// adds to a monotonic counter, polls it with relaxed agent loads + s_sleep, then issues one agent-scope acquire.  Spins are
// bounded: a workgroup that waits too long raises `*d_timeout` and everybody leaves (the enactor reports the failure).
// Residency is what makes the barrier safe: the launch code caps the grid at the occupancy of this kernel and at the CU
// count, and nothing else is running on the stream's device at that time.
struct GridBarrierState {
    unsigned *d_counter;   // zeroed by the host before every launch
    int *d_timeout;        // = d_counter + 1; zeroed together with it
};

__device__ __forceinline__ bool GridBarrier(const GridBarrierState &b, unsigned &epoch)
{
    __shared__ int s_ok;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its stores have left the CU
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        epoch += 1;
        const unsigned target = epoch * gridDim.x;
        __hip_atomic_fetch_add(b.d_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(b.d_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 4000000u || __hip_atomic_load(b.d_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(b.d_timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (__hip_atomic_load(b.d_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ok = 0;
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

template <typename VertexId, typename SizeT>
struct PersistentArgs {
    TailArgs<VertexId, SizeT> t;
    GridBarrierState barrier;
    SizeT solo_edges;          // once a level has at most this many edge slots, workgroup 0 carries on ALONE (no more grid
                               // barriers: the single-workgroup tail kernel inside this launch) until a level outgrows it
    long long unexplored_edges;  // direction-optimizing: leave when edges * switch_factor > unexplored (0 factor = never)
    double switch_factor;
    SizeT leave_below = 0;     // > 0: after at least one level, leave when a level has fewer edge slots than this (the host has
                               // a cheaper kernel for small levels: the TWC workgroup, twc.hpp)
};

template <typename KernelPolicy, typename ProblemData, typename Functor>
__device__ __forceinline__ void PersistentLevelsBody(const PersistentArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> &p,
                                                     typename ProblemData::DataSlice &slice, const long long first_iteration)
{
    typedef typename ProblemData::VertexId VertexId;
    typedef typename ProblemData::SizeT SizeT;
    typedef AdvanceShared<KernelPolicy, VertexId, SizeT> Shared;
    typedef typename Shared::Writer Writer;
    __shared__ Shared sh;
    const TailArgs<VertexId, SizeT> &t = p.t;

    Writer::Init(sh.writer);
    unsigned tile_tag;
    InitOwnerMarks<KernelPolicy>(sh, tile_tag);
    int selector = t.selector;
    long long iteration = first_iteration;  // (a parameter, not p.t.first_iteration: writing into the by-value argument block
                                            //  would move the whole block to scratch memory)
    long long unexplored = p.unexplored_edges;
    int done = 0;
    unsigned epoch = 0;
    bool solo = (gridDim.x == 1);
    unsigned long long sum_len = 0, sum_edges = 0;
    for (;;) {
        if (threadIdx.x == 0) {
            sh.level_tail = __hip_atomic_load(t.d_tail + (iteration & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (blockIdx.x == 0)
                __hip_atomic_store(t.d_tail + ((iteration + 2) & 3), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const unsigned long long tail = sh.level_tail;
        const SizeT len = static_cast<SizeT>(util::TailCount(tail));
        const SizeT edges = static_cast<SizeT>(util::TailEdges(tail));
        // every workgroup evaluates the same values => the same decision
        if (len == 0 || edges > t.edge_limit || done >= t.max_levels) break;
        if (p.switch_factor > 0 && static_cast<double>(edges) * p.switch_factor > static_cast<double>(unexplored)) break;
        if (p.leave_below > 0 && done > 0 && edges < p.leave_below) break;
        if (!solo && edges <= p.solo_edges) {
            if (blockIdx.x != 0) return;  // (the others are past the last barrier they take part in)
            solo = true;
        } else if (solo && gridDim.x > 1 && edges > p.solo_edges)
            break;  // outgrew one workgroup and the others are gone: back to the host

        AdvanceArgs<VertexId, SizeT> a;
        a.in = t.queue[selector];
        a.out = t.queue[selector ^ 1];
        a.in_len = len;
        a.in_edges = edges;
        a.d_row_offsets = t.d_row_offsets;
        a.d_column_indices = t.d_column_indices;
        a.d_tail_out = t.d_tail + ((iteration + 1) & 3);
        a.d_tail_clear = nullptr;
        a.d_overflow = t.d_overflow;
        slice.iteration = static_cast<VertexId>(iteration);

        const long long tiles = (static_cast<long long>(edges) + KernelPolicy::TILE - 1) / KernelPolicy::TILE;
        const long long per_block = solo ? tiles : (tiles + gridDim.x - 1) / gridDim.x;
        const long long tile_begin = static_cast<long long>(blockIdx.x) * per_block;
        const long long tile_end = (tile_begin + per_block < tiles) ? tile_begin + per_block : tiles;
        if (tile_begin < tile_end) {  // workgroup-uniform
            unsigned accepted = 0;
            ExpandTiles<KernelPolicy, ProblemData, Functor, true, true>(a, slice, tile_begin, tile_end, sh, accepted, tile_tag);
            const int rest = Writer::Count(sh.writer);
            __syncthreads();
            Writer::template Flush<true>(sh.writer, rest, a.out, a.d_tail_out, a.d_overflow, a.d_row_offsets);
        }
        if (solo) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this workgroup's stores and atomics have left the CU
            __syncthreads();
        } else if (!GridBarrier(p.barrier, epoch))
            break;  // (timeout: every workgroup sees the flag and leaves)
        selector ^= 1;
        ++iteration;
        ++done;
        sum_len += len;
        sum_edges += edges;
        unexplored -= edges;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *t.d_levels_done = done;
        t.d_level_sums[0] = sum_len;
        t.d_level_sums[1] = sum_edges;
    }
}

template <typename KernelPolicy, typename ProblemData, typename Functor>
__global__ __launch_bounds__(KernelPolicy::THREADS) void PersistentLevelsKernel(
    PersistentArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> p, typename ProblemData::DataSlice slice)
{
    PersistentLevelsBody<KernelPolicy, ProblemData, Functor>(p, slice, p.t.first_iteration);
}

// The same levels queued BEHIND a chain of bottom-up sweeps, before the host knows how the chain ended (sweep_chain.hpp): they
// run only when the chain ended with "return to top-down" right after a sweep that emitted its finds as a queue (and the
// queue is intact); the BSP level they start at follows from where the chain ended.  d_gate[0] = 1 when they ran (the label
// pass queued behind reads it), d_gate[1] = sweeps of the chain that ran.
template <typename KernelPolicy, typename ProblemData, typename Functor>
__global__ __launch_bounds__(KernelPolicy::THREADS) void ChainedPersistentLevelsKernel(
    PersistentArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> p, typename ProblemData::DataSlice slice,
    SweepChain chain, int sweeps, long long first_level, const int *d_queue_invalid, int *d_gate)
{
    const ChainEnd end = ChainOutcome(chain, sweeps, util::LaneId());
    const bool run = end.action == kSweepSwitch && end.end >= 2 && end.prev_action == kSweepSparseEmit && *d_queue_invalid == 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d_gate[0] = run ? 1 : 0;
        d_gate[1] = end.end - 1;
        if (!run) {
            *p.t.d_levels_done = 0;
            p.t.d_level_sums[0] = 0ull;
            p.t.d_level_sums[1] = 0ull;
        }
    }
    if (!run) return;  // (the same in every workgroup)
    PersistentLevelsBody<KernelPolicy, ProblemData, Functor>(p, slice, first_level + (end.end - 1));
}

// cooperative: launch through hipLaunchCooperativeKernel, whose launch-time check rejects a grid beyond the occupancy query
// (+15-19 us of host time per launch on MI355X; residency itself is the same as a plain launch's, so it is off by default and
// the barrier's timeout word stays the run-time safety net against CUs taken by another stream or process).
template <typename KernelPolicy, typename ProblemData, typename Functor>
hipError_t LaunchPersistentLevels(const PersistentArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> &args,
                                  const typename ProblemData::DataSlice &slice, int cu_count, int grid_hint, hipStream_t stream,
                                  bool cooperative = false)
{
    int grid = util::ResidentGrid(PersistentLevelsKernel<KernelPolicy, ProblemData, Functor>, KernelPolicy::THREADS);
    if (grid > cu_count) grid = cu_count;  // one workgroup per CU: every one of them is resident
    if (grid_hint > 0 && grid_hint < grid) grid = grid_hint;  // small levels: fewer workgroups = cheaper barrier
    if (grid < 1) grid = 1;
    // contract: barrier counter and timeout word (adjacent: WorkProgress slot 7) are ZERO at launch -- WorkProgress::Reset
    // zeroes them at the start of an Enact and every read-back (PublishKernel) re-arms them after mirroring
    if (cooperative) {
        PersistentArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> a = args;
        typename ProblemData::DataSlice s = slice;
        void *params[] = {&a, &s};
        return util::GRError(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(&PersistentLevelsKernel<KernelPolicy, ProblemData, Functor>),
                                                        dim3(grid), dim3(KernelPolicy::THREADS), params, 0, stream),
                             "advance::PersistentLevelsKernel cooperative launch failed", __FILE__, __LINE__);
    }
    hipLaunchKernelGGL((PersistentLevelsKernel<KernelPolicy, ProblemData, Functor>), dim3(grid), dim3(KernelPolicy::THREADS), 0,
                       stream, args, slice);
    return util::GRError("advance::PersistentLevelsKernel launch failed", __FILE__, __LINE__);
}

template <typename KernelPolicy, typename ProblemData, typename Functor>
hipError_t LaunchChainedPersistentLevels(const PersistentArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> &args,
                                         const typename ProblemData::DataSlice &slice, int cu_count, const SweepChain &chain, int sweeps,
                                         long long first_level, const int *d_queue_invalid, int *d_gate, hipStream_t stream)
{
    int grid = util::ResidentGrid(ChainedPersistentLevelsKernel<KernelPolicy, ProblemData, Functor>, KernelPolicy::THREADS);
    if (grid > cu_count) grid = cu_count;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((ChainedPersistentLevelsKernel<KernelPolicy, ProblemData, Functor>), dim3(grid), dim3(KernelPolicy::THREADS), 0, stream, args,
                       slice, chain, sweeps, first_level, d_queue_invalid, d_gate);
    return util::GRError("advance::ChainedPersistentLevelsKernel launch failed", __FILE__, __LINE__);
}

template <typename KernelPolicy, typename ProblemData, typename Functor>
hipError_t LaunchTailLevels(const TailArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> &args,
                            const typename ProblemData::DataSlice &slice, hipStream_t stream)
{
    hipLaunchKernelGGL((TailLevelsKernel<KernelPolicy, ProblemData, Functor>), dim3(1), dim3(KernelPolicy::THREADS), 0, stream,
                       args, slice);
    return util::GRError("advance::TailLevelsKernel launch failed", __FILE__, __LINE__);
}

}  // namespace advance
}  // namespace oprtr
}  // namespace gunrock
