// oprtr/advance/bottom_up.hpp -- backward (bottom-up) advance and the queue <-> bitmap conversions.
//
// Role of the reference's backward edge-map operators used by its direction-optimizing BFS
// (gunrock/oprtr/edge_map_backward/{kernel,cta}.cuh, edge_map_partitioned_backward/kernel.cuh, dispatched
// from oprtr/advance/kernel.cuh:164-292; enactor app/dobfs/dobfs_enactor.cuh): every UNVISITED vertex looks
// through its in-neighbours for one that is in the current frontier and adopts it as parent.
//
// gfx950 design:
//   * frontiers are bitmaps here (n/8 bytes: 2 MiB at scale-24, L2-resident on every XCD);
//   * one wave owns 64 consecutive vertices = one aligned 64-bit word of every bitmap, so the visited /
//     next-frontier words are written whole by their owner -- no atomics anywhere in the sweep;
//   * a lane first probes up to PROBE in-edges with all loads in flight (R-MAT adjacency lists are sorted
//     and hubs have small ids, so parents sit at the front), keeps going alone for a bounded number of
//     edges, and hands long unlucky rows to the whole wave (64 edges per step, ballot early-exit) so a
//     single long row cannot stall 63 idle lanes;
//   * found vertices are counted per workgroup and added to the step's packed tail (same word the forward advance
//     produces; the edge half stays 0 -- degrees are not needed while the search runs bottom-up, and the
//     bitmap -> queue conversion recomputes them exactly when it returns to top-down).
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/oprtr/advance/sweep_chain.hpp>
#include <gunrock/oprtr/frontier_writer.hpp>
#include <gunrock/util/device_intrinsics.hpp>
#include <gunrock/util/error_utils.hpp>
#include <gunrock/util/frontier.hpp>
#include <gunrock/util/kernel_runtime_stats.hpp>

namespace gunrock {
namespace oprtr {
namespace advance {

// ---- queue -> bitmap (the bitmap must be zero before) ----
template <typename VertexId, typename SizeT>
__global__ void QueueToBitmapKernel(const VertexId *d_queue, SizeT length, unsigned *d_bitmap)
{
    const SizeT stride = static_cast<SizeT>(gridDim.x) * blockDim.x;
    for (SizeT i = static_cast<SizeT>(blockIdx.x) * blockDim.x + threadIdx.x; i < length; i += stride) {
        const VertexId v = d_queue[i];
        atomicOr(d_bitmap + (static_cast<unsigned>(v) >> 5), 1u << (v & 31));
    }
}

// ---- frontier bitmap = visited now XOR visited before the last top-down level (no atomics, 3 x n/8 bytes) ----
// 218 K scattered atomicOr for the level-1 frontier of a scale-24 search cost 144 us (each is a 64-byte memory-side
// request); two streaming bitmap reads cost ~3 us.
__attribute__((unused)) static __global__ void BitmapDiffKernel(const unsigned long long *d_now, const unsigned long long *d_before,
                                        unsigned long long *d_out, long long words)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long w = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; w < words; w += stride)
        d_out[w] = d_now[w] ^ d_before[w];
}

__attribute__((unused)) static __global__ void BitmapCopyKernel(const unsigned long long *d_from, unsigned long long *d_to, long long words)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long w = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; w < words; w += stride) d_to[w] = d_from[w];
}

// ---- "fresh" byte flags -> bitmaps + labels (the closing pass of an atomic-free top-down level) ----
// A count-only top-down level marks every unvisited destination with a plain one-byte store (duplicates are harmless,
// nothing waits for a returned value; the same level with atomicOr claims was bound by the memory-side atomic rate).
// This pass owns 64 consecutive vertices per wave: it turns the bytes into the next-frontier word, ORs them into the
// visited word, writes the labels in vertex order, clears the bytes and counts the discoveries.
template <typename VertexId>
__global__ void FreshToBitmapKernel(unsigned char *d_fresh, long long nodes, unsigned long long *d_visited,
                                    const unsigned long long *d_visited_before, unsigned long long *d_frontier_out,
                                    VertexId *d_labels, VertexId label, unsigned long long *d_tail_out,
                                    unsigned long long *d_wide, const unsigned long long *d_merge = nullptr,
                                    unsigned long long *d_tail_clear = nullptr)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && d_tail_clear) *d_tail_clear = 0ull;  // ring slot of the step after next
    // One wave step = 1024 vertices: every lane loads its 16 flag bytes with ONE 16-byte load (the byte-per-lane version
    // issued 16x the load instructions and ran at 107 us for a 16 MiB map), squeezes them into 16 bits, and four
    // neighbouring lanes assemble one 64-bit bitmap word.  d_fresh is padded to a multiple of 1024 bytes (bfs_problem.hpp).
    const unsigned lane = util::LaneId();
    const long long words = (nodes + 63) / 64;
    const long long steps = (words + 15) / 16;
    const long long wave0 = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) / util::kWaveSize;
    const long long nwaves = static_cast<long long>(gridDim.x) * blockDim.x / util::kWaveSize;
    const unsigned quad = lane & 3u;
    unsigned count = 0;
    for (long long step = wave0; step < steps; step += nwaves) {
        const long long my_word = step * 16 + (lane >> 2);
        const bool in_range = my_word < words;
        uint4 *my_flags = reinterpret_cast<uint4 *>(d_fresh) + (step * 64 + lane);
        const uint4 f = *my_flags;
        // "seen" is the bitmap as it was BEFORE the level (the four lanes of a quad read the same word: one request)
        const unsigned long long seen = in_range ? d_visited_before[my_word] : ~0ull;
        // bytes are 0 or 1: (w * 0x01020408) >> 24 gathers the four low bits of a dword's bytes
        const unsigned m16 = ((f.x * 0x01020408u) >> 24 & 0xFu) | ((f.y * 0x01020408u) >> 24 & 0xFu) << 4 |
                             ((f.z * 0x01020408u) >> 24 & 0xFu) << 8 | ((f.w * 0x01020408u) >> 24 & 0xFu) << 12;
        if (m16) *my_flags = make_uint4(0, 0, 0, 0);
        unsigned long long word = static_cast<unsigned long long>(m16) << (16 * quad);
        word |= __shfl_xor(word, 1, util::kWaveSize);
        word |= __shfl_xor(word, 2, util::kWaveSize);
        const unsigned long long mask = word & ~seen;
        unsigned mine = static_cast<unsigned>(mask >> (16 * quad)) & 0xFFFFu;
        const long long v0 = (step * 64 + lane) * 16;
        if (d_labels) {  // (kernel argument: uniform) nullptr = labels are deferred: the level's bitmap is kept and
                         // EmitLabelsKernel (bfs_problem.hpp) writes all labels of the search in one coalesced pass
            while (mine) {
                const int b = __builtin_ctz(mine);
                mine &= mine - 1;
                d_labels[v0 + b] = label;
            }
        }
        if (quad == 0 && in_range) {
            d_frontier_out[my_word] = d_merge ? (mask | d_merge[my_word]) : mask;  // (d_merge: this level's head-pass finds)
            d_visited[my_word] = seen | mask;  // authoritative
            count += static_cast<unsigned>(__popcll(mask));
        }
    }
    // workgroup total -> one atomic on this workgroup's line of the wide tail (or on d_tail_out when there is none)
    __shared__ unsigned long long s_wave_total[16];
    unsigned long long total = util::WaveSum(static_cast<unsigned long long>(count));
    if (lane == 0) s_wave_total[threadIdx.x / util::kWaveSize] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long sum = 0;
        for (unsigned w = 0; w < blockDim.x / util::kWaveSize; ++w) sum += s_wave_total[w];
        unsigned long long *slot = util::WideTailSlot(d_wide);
        if (sum) atomicAdd(slot ? slot : d_tail_out, sum);
    }
}

// Frontier membership tests for the bottom-up sweep.
// BitmapLookup: one bitmap indexed by vertex id (single GPU).
// StripedBitmapLookup: vertex-cut over P ranks, owner = v mod P, local id = v div P (the reference's only multi-GPU
// vestige, problem_base.cuh:185-210); the all-gathered bitmaps of the P ranks sit back to back, words_per_rank apart.
template <typename VertexId>
struct BitmapLookup {
    const unsigned *bits;
    __device__ __forceinline__ bool operator()(VertexId u) const
    {
        // plain (L1-cached) probe: the in-neighbours asked about first are the high-degree ones, whose frontier words stay in
        // L1 -- an L1-bypassing (sc1) probe measured 15 % slower over the whole search (0.398 vs 0.346 ms per scale-24 search)
        return (bits[static_cast<unsigned>(u) >> 5] >> (u & 31)) & 1u;
    }
};
// Frontier bitmaps of `parts` ranks, one after the other (the all-gather's layout); vertex u is bit (u / parts) of rank
// (u mod parts).  The quotient comes from a multiplication by a precomputed reciprocal: gfx950 has no integer divide, a
// runtime-divisor u / parts is a ~35-instruction sequence, and this runs once per probed in-edge of a bottom-up sweep.
// Exact for 0 <= u < 2^31 and parts <= 2^15 (magic = floor(2^shift / parts) + 1, shift = 32 + ceil(log2 parts):
// u * (magic * parts - 2^shift) < 2^shift, and u * magic < 2^64).
template <typename VertexId>
struct StripedBitmapLookup {
    const unsigned *bits;
    unsigned parts;
    unsigned words_per_rank;
    unsigned long long magic;
    unsigned shift;
    __host__ __device__ StripedBitmapLookup(const unsigned *bits_, unsigned parts_, unsigned words_per_rank_)
        : bits(bits_), parts(parts_), words_per_rank(words_per_rank_)
    {
        unsigned s = 0;
        while ((1u << s) < parts) ++s;
        shift = 32 + s;
        magic = (1ull << shift) / parts + 1ull;
    }
    __device__ __forceinline__ bool operator()(VertexId u) const
    {
        const unsigned local = static_cast<unsigned>((static_cast<unsigned long long>(static_cast<unsigned>(u)) * magic) >> shift);
        const unsigned owner = static_cast<unsigned>(u) - local * parts;
        return (bits[owner * words_per_rank + (local >> 5)] >> (local & 31)) & 1u;
    }
};

template <typename VertexId, typename SizeT>
struct BottomUpArgs {
    SizeT nodes;
    const SizeT *d_inv_row_offsets;
    const VertexId *d_inv_column_indices;
    const int2 *d_inv_heads;                // first two in-neighbours per vertex (-1 padded)
    unsigned long long *d_frontier_out;     // next frontier bitmap, every word is written
    unsigned long long *d_visited;          // visited bitmap, owner-updated
    unsigned long long *d_tail_out;
    unsigned long long *d_tail_clear;
    int head_skip = 2;                      // row entries the heads stand for: 2 = the first two (positional heads), 0 = ranked heads
    // compacted heads: vertices without in-edges (bit set in d_never) have no head entry; a vertex's entry sits at
    // d_head_base[word] + (number of vertices WITH in-edges before it in its 64-vertex word).  nullptr: indexed by vertex id.
    const unsigned long long *d_never = nullptr;
    const unsigned *d_head_base = nullptr;
    // the compacting sweep can also emit its finds as the next top-down queue (vertex, row start, degree prefix), so a
    // switch back to top-down needs no bitmap -> queue pass: forward row offsets, the ring slot that receives the packed
    // tail, and a flag raised when a workgroup's staging buffer overflowed (the queue is then unusable, the bitmap is not)
    util::Frontier<VertexId, SizeT> queue_out;
    const SizeT *d_fwd_row_offsets = nullptr;
    unsigned long long *d_queue_tail = nullptr;
    int *d_queue_invalid = nullptr;
    int *d_overflow = nullptr;
    int heads_only = 0;                     // 1: probe the adjacency heads and stop (no CSR walk): a cheap first cut of a level
    unsigned long long *d_wide = nullptr;   // when set, workgroup counts go to WorkProgress's wide tail instead of d_tail_out
    unsigned long long *d_duty = nullptr;   // INSTRUMENT: this launch's runtime-stamp words (util/kernel_runtime_stats.hpp)
};

// 64-vertex bitmap words one wave takes per step of the bottom-up sweep (launch code sizes the grid from it)
#ifndef GRX_BU_STEP_WORDS
#define GRX_BU_STEP_WORDS 8
#endif
constexpr int kBottomUpStepWords = GRX_BU_STEP_WORDS;
// 64-vertex bitmap words per chunk of the compacting sweep (one word per lane: at most 64).  A pass over a chunk costs the same
// chain of dependent round trips whether it carries 64 unvisited vertices or 6, so the chunk should hold about a wave's worth:
// with 16 words a late level (a dozen open vertices per 1024) ran three nearly empty passes per wave.
#ifndef GRX_BU_SPARSE_CHUNK_WORDS
#define GRX_BU_SPARSE_CHUNK_WORDS 64
#endif
constexpr int kSparseChunkWords = GRX_BU_SPARSE_CHUNK_WORDS;

struct __attribute__((packed, aligned(4))) Quad {
    int v[4];
};

// first two in-neighbours of every vertex, -1 padded: the "adjacency head" (8 bytes per vertex, built once per
// inverse graph).  Consecutive vertices have consecutive heads, so a wave's 64 lanes read 512 contiguous bytes,
// where going to the CSR row costs a 64-byte line per vertex to use 4-8 bytes of it.
// Which two?  The two in-neighbours of LARGEST degree among the first kHeadScan of the row: a bottom-up level asks "is any
// in-neighbour in the frontier", and on a scale-free graph the well-connected neighbours are the ones discovered early, so
// probing them first ends most searches at the heads (the first two by id were in the frontier far less often and sent
// the vertex on to its CSR row: a 64-byte line for the offsets and another for the columns).  One wave per vertex.
constexpr int kHeadScan = 512;

template <typename VertexId, typename SizeT>
__global__ void BuildHeadsKernel(const SizeT *d_row_offsets, const VertexId *d_column_indices, long long nodes, int2 *d_heads,
                                 const SizeT *d_degree_offsets,  // row offsets the neighbour ids index (nullptr: rank by position)
                                 const unsigned long long *d_never = nullptr, const unsigned *d_head_base = nullptr)
{
    const unsigned lane = util::LaneId();
    const long long nwaves = static_cast<long long>(gridDim.x) * blockDim.x / util::kWaveSize;
    for (long long v = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) / util::kWaveSize; v < nodes; v += nwaves) {
        const SizeT b = d_row_offsets[v], e = d_row_offsets[v + 1];
        const SizeT len = (e - b < kHeadScan) ? e - b : kHeadScan;
        // key = (degree + 1) << 32 | id; 0 = nothing
        unsigned long long k1 = 0, k2 = 0;
        for (SizeT i = lane; i < len; i += util::kWaveSize) {
            const VertexId u = d_column_indices[b + i];
            // (no degree table -- the partitioned problem's columns are global ids -- : earlier position = larger key)
            const unsigned long long deg = d_degree_offsets ? static_cast<unsigned long long>(d_degree_offsets[u + 1] - d_degree_offsets[u])
                                                            : static_cast<unsigned long long>(kHeadScan - i);
            const unsigned long long k = ((deg + 1) << 32) | static_cast<unsigned>(u);
            if (k > k1) { k2 = k1; k1 = k; }
            else if (k > k2) k2 = k;
        }
        unsigned long long best1 = k1;
        for (int o = 32; o; o >>= 1) {
            const unsigned long long other = __shfl_xor(best1, o, util::kWaveSize);
            best1 = other > best1 ? other : best1;
        }
        unsigned long long best2 = (k1 == best1) ? k2 : k1;  // (ids are distinct in a deduplicated row; a duplicate costs nothing)
        for (int o = 32; o; o >>= 1) {
            const unsigned long long other = __shfl_xor(best2, o, util::kWaveSize);
            best2 = other > best2 ? other : best2;
        }
        if (lane == 0) {
            int2 h;
            h.x = best1 ? static_cast<int>(static_cast<unsigned>(best1)) : -1;
            h.y = best2 ? static_cast<int>(static_cast<unsigned>(best2)) : -1;
            if (!d_head_base) {
                d_heads[v] = h;
            } else if (e > b) {  // compacted layout: only vertices with in-edges own an entry
                const unsigned long long with_edges = ~d_never[v >> 6];
                d_heads[d_head_base[v >> 6] + __popcll(with_edges & ((1ull << (v & 63)) - 1ull))] = h;
            }
        }
    }
}

// The CSR walk of one pending vertex per lane (wave-collective: phase C uses the whole wave).  Returns the parent found in the
// current frontier or -1.
//   phase B: PROBE edges per round, up to SOLO_LIMIT edges per lane.  The round's in-neighbour ids were fetched during the
//            PREVIOUS round (software prefetch), so their frontier probes and the next round's id loads are in flight together:
//            one memory round trip per round instead of two (PMC: this loop is a pure latency chain).
//   phase C: rows still unresolved are swept by the whole wave, 256 in-edges per step.
template <int PROBE, int SOLO_LIMIT, typename VertexId, typename SizeT, typename Lookup>
__device__ __forceinline__ VertexId WalkRow(const BottomUpArgs<VertexId, SizeT> &a, const Lookup &in_frontier, bool active, VertexId v,
                                            unsigned lane, VertexId known_a = -1, VertexId known_b = -1, bool extent_given = false,
                                            SizeT given_begin = 0, SizeT given_end = 0)
{
    // extent_given: the caller fetched the row's extent already (one walk ahead: the fetch is the first of the walk's dependent
    // round trips, and the vertex of the NEXT walk is known while this one runs)
    // known_a / known_b: in-neighbours the caller has already probed (the adjacency heads, which are entries of this row): the
    // walk does not ask about them again -- two of the ~6-8 probes of a vertex that fails at this level
    SizeT pos = 0, end = 0;
    VertexId p_found = -1;
    if (extent_given) {
        pos = given_begin + a.head_skip;
        end = given_end;
    } else if (active) {
        pos = a.d_inv_row_offsets[v] + a.head_skip;
        end = a.d_inv_row_offsets[v + 1];
    }
    VertexId cur[PROBE];
    auto fetch = [&](VertexId (&dst)[PROBE], SizeT from, bool wanted) {
        if (wanted && from + PROBE <= end && (PROBE % 4) == 0) {
#pragma unroll
            for (int qd = 0; qd < PROBE / 4; ++qd) {  // 16-byte loads (rows are 4-byte aligned: gfx950 takes that)
                const Quad q = *reinterpret_cast<const Quad *>(a.d_inv_column_indices + from + 4 * qd);
#pragma unroll
                for (int k = 0; k < 4; ++k) dst[4 * qd + k] = q.v[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < PROBE; ++k)
                dst[k] = (wanted && from + k < end) ? a.d_inv_column_indices[from + k] : static_cast<VertexId>(-1);
        }
    };
    fetch(cur, pos, active && pos < end);
    for (int done = 0; done < SOLO_LIMIT; done += PROBE) {
        if (__ballot(active && p_found < 0 && pos < end) == 0) break;  // wave-uniform
        VertexId nxt[PROBE];
        fetch(nxt, pos + PROBE, active && p_found < 0 && pos + PROBE < end && done + PROBE < SOLO_LIMIT);
        bool fw[PROBE];
#pragma unroll
        for (int k = 0; k < PROBE; ++k)
            fw[k] = (active && p_found < 0 && cur[k] >= 0 && cur[k] != known_a && cur[k] != known_b) ? in_frontier(cur[k]) : false;
#pragma unroll
        for (int k = 0; k < PROBE; ++k)
            if (p_found < 0 && fw[k]) p_found = cur[k];
        pos += PROBE;
#pragma unroll
        for (int k = 0; k < PROBE; ++k) cur[k] = nxt[k];
    }
    unsigned long long todo = __ballot(active && p_found < 0 && pos < end);
    while (todo) {
        const int leader = __ffsll(static_cast<long long>(todo)) - 1;
        SizeT p = __shfl(pos, leader, util::kWaveSize);
        const SizeT e = __shfl(end, leader, util::kWaveSize);
        VertexId hit_parent = -1;
        for (; p < e && hit_parent < 0; p += 4 * util::kWaveSize) {  // 4 loads in flight per lane
            VertexId u[4];
            bool h[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const SizeT mine = p + static_cast<SizeT>(k * util::kWaveSize + lane);
                u[k] = (mine < e) ? a.d_inv_column_indices[mine] : static_cast<VertexId>(-1);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) h[k] = (u[k] >= 0) ? in_frontier(u[k]) : false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long long hm = __ballot(h[k]);
                if (hm && hit_parent < 0) hit_parent = __shfl(u[k], __ffsll(static_cast<long long>(hm)) - 1, util::kWaveSize);
            }
        }
        if (static_cast<int>(lane) == leader) {
            p_found = hit_parent;
            pos = end;
        }
        todo &= todo - 1;
    }
    return p_found;
}

// The dense sweep (whole workgroup).  BottomUpKernel is this body alone; BottomUpAutoKernel picks it or the compacting sweep on
// the device.
// HEADS_ONLY: the body without its row walks (the first cut of a "heads, then the rest" level): a third of the registers.
template <int THREADS, int PROBE, int SOLO_LIMIT, typename ProblemData, typename Lookup, bool HEADS_ONLY = false>
__device__ __forceinline__ void DenseSweep(const BottomUpArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> &a,
                                           typename ProblemData::DataSlice &slice, const Lookup &in_frontier)
{
    typedef typename ProblemData::VertexId VertexId;
    constexpr int WAVES = THREADS / util::kWaveSize;
    constexpr int STEP_WORDS = kBottomUpStepWords;  // bitmap words (x64 vertices) a wave takes per step; lanes 0..STEP_WORDS-1 own one word each
    __shared__ unsigned long long s_total[WAVES];
    util::DutyStamp duty(a.d_duty);

    const int tid = threadIdx.x;
    const unsigned lane = util::LaneId();
    if (blockIdx.x == 0 && tid == 0 && a.d_tail_clear) *a.d_tail_clear = 0ull;

    const long long words = (static_cast<long long>(a.nodes) + 63) / 64;
    const long long steps = (words + STEP_WORDS - 1) / STEP_WORDS;
    const long long wave0 = (static_cast<long long>(blockIdx.x) * THREADS + tid) / util::kWaveSize;
    const long long nwaves = static_cast<long long>(gridDim.x) * WAVES;
    const VertexId new_label = slice.iteration + 1;

    unsigned found_count = 0;
    unsigned found_edges = 0;
    for (long long step = wave0; step < steps; step += nwaves) {
        // lanes 0..STEP_WORDS-1 read the visited words of this step with one coalesced load
        const long long my_word = step * STEP_WORDS + lane;
        const bool owns_word = lane < STEP_WORDS && my_word < words;
        unsigned long long my_vis = ~0ull;
        if (owns_word) my_vis = a.d_visited[my_word];
        unsigned long long my_open = ~my_vis;
        if (owns_word && (my_word + 1) * 64 > a.nodes) {  // last word: bits past the vertex count are not vertices
            const int valid = static_cast<int>(a.nodes - my_word * 64);
            my_open &= (valid >= 64) ? ~0ull : ((1ull << valid) - 1ull);
        }
        unsigned long long my_found = 0;
        if (__ballot(owns_word && my_open != 0) != 0) {  // wave-uniform: the step has unvisited vertices
            // The sweep is latency-bound (PMC: waves wait ~90 % of their cycles on a chain of ~5 dependent loads per
            // word), so the loads of all STEP_WORDS words are issued together, BRANCH-FREE: a lane with nothing to ask
            // reads a harmless hot address instead of being masked off, which keeps the loads in one basic block where
            // the compiler leaves them all in flight.
            // ---- phase H1: adjacency heads (coalesced 8 bytes per lane and word)
            int2 head[STEP_WORDS];
            unsigned open_bits = 0;  // bit j: this lane's vertex of word j is unvisited
            if (a.d_head_base) {  // (kernel argument: wave-uniform) compacted heads: entry index from the never-mask ranks
                unsigned long long my_with_edges = 0;
                unsigned my_base = 0;
                if (owns_word) {
                    my_with_edges = ~a.d_never[my_word];
                    my_base = a.d_head_base[my_word];
                }
#pragma unroll
                for (int j = 0; j < STEP_WORDS; ++j) {
                    const unsigned long long open_mask = __shfl(my_open, j, util::kWaveSize);
                    const bool open = (open_mask >> lane) & 1ull;
                    open_bits |= static_cast<unsigned>(open) << j;
                    const unsigned long long we = __shfl(my_with_edges, j, util::kWaveSize);
                    const unsigned idx = __shfl(my_base, j, util::kWaveSize) + static_cast<unsigned>(__popcll(we & ((1ull << lane) - 1ull)));
                    head[j] = a.d_inv_heads[open ? idx : 0u];
                }
            } else {
#pragma unroll
                for (int j = 0; j < STEP_WORDS; ++j) {
                    const unsigned long long open_mask = __shfl(my_open, j, util::kWaveSize);
                    const bool open = (open_mask >> lane) & 1ull;
                    open_bits |= static_cast<unsigned>(open) << j;
                    const long long v = (step * STEP_WORDS + j) * 64 + lane;
                    head[j] = a.d_inv_heads[open ? v : 0];
                }
            }
            // ---- phase H2: first in-neighbour of every open vertex
            bool hit[STEP_WORDS];
#pragma unroll
            for (int j = 0; j < STEP_WORDS; ++j) {
                const bool ask = ((open_bits >> j) & 1u) && head[j].x >= 0;
                hit[j] = in_frontier(ask ? head[j].x : 0) && ask;
            }
            VertexId parent[STEP_WORDS];
            unsigned ask_y = 0;
#pragma unroll
            for (int j = 0; j < STEP_WORDS; ++j) {
                parent[j] = hit[j] ? head[j].x : static_cast<VertexId>(-1);
                if (((open_bits >> j) & 1u) && head[j].x >= 0 && !hit[j] && head[j].y >= 0) ask_y |= 1u << j;
            }
            // ---- phase H3: second in-neighbour, only where the first missed (skipped when no lane needs it)
            unsigned more_bits = 0;  // bit j: continue in the CSR row
            if (__ballot(ask_y != 0) != 0) {
#pragma unroll
                for (int j = 0; j < STEP_WORDS; ++j) {
                    const bool ask = (ask_y >> j) & 1u;
                    hit[j] = in_frontier(ask ? head[j].y : 0) && ask;
                }
#pragma unroll
                for (int j = 0; j < STEP_WORDS; ++j) {
                    if (hit[j]) parent[j] = head[j].y;
                    if (((ask_y >> j) & 1u) && !hit[j]) more_bits |= 1u << j;
                }
            }

            if (HEADS_ONLY || a.heads_only) more_bits = 0;
            // ---- rows longer than the head continue in their CSR row (from the third entry when the heads are the first two).  Only a few percent of the
            //      vertices get here, but nearly every WORD has one: walking the words one after another would put ~3
            //      dependent round trips per word back on the critical path.  Instead every lane takes ITS OWN next
            //      pending vertex (whatever word it sits in), so all pending vertices of the step advance together and
            //      the loop runs max-over-lanes(pending) times -- 1 or 2.
            // (the row extent of a lane's NEXT pending vertex is fetched while the current walk runs)
            typedef typename ProblemData::SizeT SizeT;
            SizeT next_begin = 0, next_end = 0;
            if (!HEADS_ONLY && more_bits != 0) {
                const long long nv = (step * STEP_WORDS + (__ffs(more_bits) - 1)) * 64 + lane;
                next_begin = a.d_inv_row_offsets[nv];
                next_end = a.d_inv_row_offsets[nv + 1];
            }
            while (!HEADS_ONLY && __ballot(more_bits != 0) != 0) {  // wave-uniform
                const bool active = more_bits != 0;
                const int jl = active ? (__ffs(more_bits) - 1) : 0;
                more_bits &= more_bits - 1;
                const VertexId v = static_cast<VertexId>((step * STEP_WORDS + jl) * 64 + lane);
                const SizeT row_begin = next_begin, row_end = next_end;
                if (more_bits != 0) {
                    const long long nv = (step * STEP_WORDS + (__ffs(more_bits) - 1)) * 64 + lane;
                    next_begin = a.d_inv_row_offsets[nv];
                    next_end = a.d_inv_row_offsets[nv + 1];
                }
                VertexId hx = -1, hy = -1;  // the heads of that vertex (already probed)
#pragma unroll
                for (int j = 0; j < STEP_WORDS; ++j)
                    if (jl == j) {
                        hx = head[j].x;
                        hy = head[j].y;
                    }
                const VertexId p_found = WalkRow<PROBE, SOLO_LIMIT>(a, in_frontier, active, v, lane, hx, hy, true, active ? row_begin : 0,
                                                                    active ? row_end : 0);
                const bool late = active && p_found >= 0;
                if (late) {
                    if (!slice.defer_labels) slice.d_labels[v] = new_label;
                    if (ProblemData::MARK_PREDECESSORS) slice.d_preds[v] = p_found;
                    found_count += 1;
                }
#pragma unroll
                for (int j = 0; j < STEP_WORDS; ++j) {  // fold the late discoveries into their words' result masks
                    const unsigned long long fm = __ballot(late && jl == j);
                    if (static_cast<int>(lane) == j) my_found |= fm;
                }
            }

            // ---- record the discoveries: labels (consecutive lanes = consecutive vertices) and the bitmap words
#pragma unroll
            for (int j = 0; j < STEP_WORDS; ++j) {
                const bool found = parent[j] >= 0;
                if (found) {
                    const VertexId v = static_cast<VertexId>((step * STEP_WORDS + j) * 64 + lane);
                    if (!slice.defer_labels) slice.d_labels[v] = new_label;
                    if (ProblemData::MARK_PREDECESSORS) slice.d_preds[v] = parent[j];
                    found_count += 1;
                }
                const unsigned long long fm = __ballot(found);
                if (static_cast<int>(lane) == j) my_found |= fm;
            }
        }
        if (owns_word) {  // coalesced write-back of the step's words
            a.d_frontier_out[my_word] = my_found;
            if (my_found) a.d_visited[my_word] = my_vis | my_found;
        }
    }

    // workgroup reduction of (vertices, edges) -> one packed atomic
    unsigned long long packed = util::PackTail(found_count, found_edges);
    packed = util::WaveSum(packed);
    if (lane == 0) s_total[tid / util::kWaveSize] = packed;
    __syncthreads();
    if (tid == 0) {
        unsigned long long sum = 0;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) sum += s_total[i];
        unsigned long long *slot = util::WideTailSlot(a.d_wide);
        if (sum) atomicAdd(slot ? slot : a.d_tail_out, sum);
    }
}

// (6 waves per SIMD: the sweep is a chain of dependent round trips per step, so resident waves are what hides them; the
//  compiler's own choice drifted to 86 VGPRs = 5 waves when the body became a function)
#ifndef GRX_BU_MIN_WAVES
#define GRX_BU_MIN_WAVES 6
#endif
template <int THREADS, int PROBE, int SOLO_LIMIT, typename ProblemData, typename Lookup>
__global__ __launch_bounds__(THREADS, GRX_BU_MIN_WAVES) void BottomUpKernel(
    BottomUpArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> a, typename ProblemData::DataSlice slice,
    Lookup in_frontier)
{
    DenseSweep<THREADS, PROBE, SOLO_LIMIT, ProblemData, Lookup>(a, slice, in_frontier);
}

#ifndef GRX_BU_HEADS_MIN_WAVES
#define GRX_BU_HEADS_MIN_WAVES 6
#endif
template <int THREADS, typename ProblemData, typename Lookup>
__global__ __launch_bounds__(THREADS, GRX_BU_HEADS_MIN_WAVES) void BottomUpHeadsKernel(
    BottomUpArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> a, typename ProblemData::DataSlice slice,
    Lookup in_frontier)
{
    DenseSweep<THREADS, 8, 32, ProblemData, Lookup, true>(a, slice, in_frontier);
}

// ---- bottom-up sweep for a nearly finished search (few unvisited vertices) ----
// The dense kernel runs its unrolled 8-word pipeline for every step that holds ANY unvisited vertex; late in a search that is
// every step (a few hundred thousand unvisited vertices spread over 262 K words) with a dozen busy lanes each, so a level
// costs the same ~25 us whether it finds 3 M vertices or 300.  Here a wave takes 16 words (1024 vertices), compacts their
// unvisited vertices onto its lanes (prefix sums of the words' popcounts; lane k takes the k-th one), and runs ONE pass of
// head probes + row walk per 64 of them.  Results are folded into per-wave found words in LDS and written back whole, so
// the bitmaps keep their single-writer property.
__device__ __forceinline__ int NthSetBit(unsigned long long x, int r)  // position of the r-th (0-based) set bit; r < popc(x)
{
    int pos = 0;
#pragma unroll
    for (int s = 32; s; s >>= 1) {
        const int c = __popcll(x & (((1ull << s) - 1ull) << pos));
        if (r >= c) {
            r -= c;
            pos += s;
        }
    }
    return pos;
}

// EMIT_QUEUE: the body CAN stage its finds and flush them as a queue (LDS for the staging buffer); `emit` says whether this launch
// does.  `grid_limit`: workgroups that take part (the caller has sent the others home).
template <int THREADS, int PROBE, int SOLO_LIMIT, typename ProblemData, typename Lookup, bool EMIT_QUEUE>
__device__ __forceinline__ void SparseSweep(const BottomUpArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> &a,
                                            typename ProblemData::DataSlice &slice, const Lookup &in_frontier, const bool emit,
                                            const unsigned grid_limit)
{
    typedef typename ProblemData::VertexId VertexId;
    typedef typename ProblemData::SizeT SizeT;
    constexpr int WAVES = THREADS / util::kWaveSize;
    constexpr int CHUNK_WORDS = kSparseChunkWords;
    static_assert(CHUNK_WORDS <= 64 && (CHUNK_WORDS & (CHUNK_WORDS - 1)) == 0, "one bitmap word per lane, a power of two");
    constexpr int CAPACITY = 16 * THREADS;
    typedef FrontierWriter<THREADS, EMIT_QUEUE ? CAPACITY : THREADS, VertexId, SizeT> Writer;
    __shared__ unsigned long long s_total[WAVES];
    __shared__ unsigned s_found[WAVES][CHUNK_WORDS * 2];  // found bits of the wave's current chunk, 32-bit halves
    util::DutyStamp duty(a.d_duty);
    __shared__ typename Writer::Storage s_writer;
    if (EMIT_QUEUE && emit) {
        Writer::Init(s_writer);
        __syncthreads();
    }

    const int tid = threadIdx.x;
    const unsigned lane = util::LaneId();
    const unsigned wave = tid / util::kWaveSize;
    if (blockIdx.x == 0 && tid == 0 && a.d_tail_clear) *a.d_tail_clear = 0ull;

    const long long words = (static_cast<long long>(a.nodes) + 63) / 64;
    const long long chunks = (words + CHUNK_WORDS - 1) / CHUNK_WORDS;
    const long long wave0 = (static_cast<long long>(blockIdx.x) * THREADS + tid) / util::kWaveSize;
    const long long nwaves = static_cast<long long>(grid_limit) * WAVES;
    const VertexId new_label = slice.iteration + 1;
    unsigned found_count = 0;

    for (long long chunk = wave0; chunk < chunks; chunk += nwaves) {
        const long long my_word = chunk * CHUNK_WORDS + lane;
        const bool owns_word = lane < CHUNK_WORDS && my_word < words;
        unsigned long long my_vis = ~0ull;
        if (owns_word) my_vis = a.d_visited[my_word];
        unsigned long long my_open = ~my_vis;
        if (owns_word && (my_word + 1) * 64 > a.nodes) {
            const int valid = static_cast<int>(a.nodes - my_word * 64);
            my_open &= (valid >= 64) ? ~0ull : ((1ull << valid) - 1ull);
        }
        const unsigned cnt = static_cast<unsigned>(__popcll(my_open));  // 0 on lanes that own no word
        const unsigned inc = util::WaveInclusiveSum(cnt);
        const unsigned total = __shfl(inc, CHUNK_WORDS - 1, util::kWaveSize);
        if (total == 0) {  // wave-uniform
            if (owns_word) a.d_frontier_out[my_word] = 0ull;
            continue;
        }
        // compacted heads (see BottomUpArgs): per-word rank tables
        unsigned long long my_with_edges = ~0ull;
        unsigned my_base = 0;
        if (a.d_head_base && owns_word) {
            my_with_edges = ~a.d_never[my_word];
            my_base = a.d_head_base[my_word];
        }
        for (int i = lane; i < CHUNK_WORDS * 2; i += util::kWaveSize) s_found[wave][i] = 0;  // wave-private rows: the wave's own program order is enough
        __builtin_amdgcn_wave_barrier();

        for (unsigned base = 0; base < total; base += util::kWaveSize) {
            const unsigned k = base + lane;
            const bool active = k < total;
            // word of the k-th unvisited vertex: number of words whose inclusive count is <= k
            int j = 0;  // (inc is non-decreasing over the lanes: binary search, log2(CHUNK_WORDS) shuffles)
#pragma unroll
            for (int s = CHUNK_WORDS / 2; s; s >>= 1)
                if (__shfl(inc, j + s - 1, util::kWaveSize) <= k) j += s;
            if (!active) j = 0;
            const unsigned before = __shfl(inc, j, util::kWaveSize) - __shfl(cnt, j, util::kWaveSize);
            const unsigned long long open_j = __shfl(my_open, j, util::kWaveSize);
            const int bit = active ? NthSetBit(open_j, static_cast<int>(k - before)) : 0;
            const VertexId v = static_cast<VertexId>((chunk * CHUNK_WORDS + j) * 64 + bit);
            // heads
            long long head_index = v;
            if (a.d_head_base) {
                const unsigned long long we = __shfl(my_with_edges, j, util::kWaveSize);
                head_index = __shfl(my_base, j, util::kWaveSize) + __popcll(we & ((1ull << bit) - 1ull));
            }
            const int2 head = a.d_inv_heads[active ? head_index : 0];
            VertexId parent = -1;
            const bool ask_x = active && head.x >= 0;
            const bool hit_x = in_frontier(ask_x ? head.x : 0) && ask_x;
            if (hit_x) parent = head.x;
            const bool ask_y = ask_x && !hit_x && head.y >= 0;
            bool more = false;
            if (__ballot(ask_y) != 0) {
                const bool hit_y = in_frontier(ask_y ? head.y : 0) && ask_y;
                if (hit_y) parent = head.y;
                more = ask_y && !hit_y;  // a full head: the row may hold more
            }
            if (a.heads_only) more = false;
            if (__ballot(more) != 0) {
                const VertexId late = WalkRow<PROBE, SOLO_LIMIT>(a, in_frontier, more, v, lane, head.x, head.y);
                if (more && late >= 0) parent = late;
            }
            if (parent >= 0) {
                if (!slice.defer_labels) slice.d_labels[v] = new_label;
                if (ProblemData::MARK_PREDECESSORS) slice.d_preds[v] = parent;
                atomicOr(&s_found[wave][2 * j + (bit >> 5)], 1u << (bit & 31));
            }
            if (EMIT_QUEUE && emit) {  // stage the finds for the queue (one LDS atomic per wave; past the capacity they are only counted)
                const unsigned long long fm = __ballot(parent >= 0);
                if (fm) {
                    const int leader = __ffsll(static_cast<long long>(fm)) - 1;
                    int at = 0;
                    if (static_cast<int>(lane) == leader) at = atomicAdd(&s_writer.count, __popcll(fm));
                    at = __shfl(at, leader, util::kWaveSize);
                    if (parent >= 0 && at + __popcll(fm) <= CAPACITY) s_writer.buf[at + util::RankInMask(fm)] = v;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (owns_word) {
            const unsigned long long f = static_cast<unsigned long long>(s_found[wave][2 * lane]) |
                                         (static_cast<unsigned long long>(s_found[wave][2 * lane + 1]) << 32);
            a.d_frontier_out[my_word] = f;
            if (f) a.d_visited[my_word] = my_vis | f;
            found_count += static_cast<unsigned>(__popcll(f));
        }
    }

    unsigned long long packed = util::PackTail(found_count, 0u);
    packed = util::WaveSum(packed);
    if (lane == 0) s_total[wave] = packed;
    __syncthreads();
    if (tid == 0) {
        unsigned long long sum = 0;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) sum += s_total[i];
        unsigned long long *slot = util::WideTailSlot(a.d_wide);
        if (sum) atomicAdd(slot ? slot : a.d_tail_out, sum);
    }
    if (EMIT_QUEUE && emit) {
        const int staged = Writer::Count(s_writer);  // (the barrier above ordered it after every append)
        __syncthreads();
        if (staged > CAPACITY) {
            if (tid == 0) *a.d_queue_invalid = 1;
        } else {
            Writer::template Flush<true>(s_writer, staged, a.queue_out, a.d_queue_tail, a.d_overflow, a.d_fwd_row_offsets);
        }
    }
}

template <int THREADS, int PROBE, int SOLO_LIMIT, typename ProblemData, typename Lookup, bool EMIT_QUEUE>
__global__ __launch_bounds__(THREADS) void BottomUpSparseKernel(
    BottomUpArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> a, typename ProblemData::DataSlice slice,
    Lookup in_frontier)
{
    SparseSweep<THREADS, PROBE, SOLO_LIMIT, ProblemData, Lookup, EMIT_QUEUE>(a, slice, in_frontier, EMIT_QUEUE, gridDim.x);
}

template <int THREADS, int PROBE, int SOLO_LIMIT, typename ProblemData, typename Lookup>
__global__ __launch_bounds__(THREADS, GRX_BU_MIN_WAVES) void BottomUpAutoKernel(
    BottomUpArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> a, typename ProblemData::DataSlice slice,
    Lookup in_frontier, SweepChain chain, unsigned sparse_grid, unsigned emit_grid)
{
    const int action = ChainAction(chain, util::LaneId());
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        chain.d_log[chain.index] = action;
        // a sweep that emits no queue leaves its queue-tail slot zero, so that a bitmap -> queue conversion can follow directly
        if (action >= kSweepDense && action != kSweepSparseEmit && a.d_queue_tail) *a.d_queue_tail = 0ull;
    }
    if (action < kSweepDense) return;  // (uniform over the grid)
    if (action == kSweepDense) {
        DenseSweep<THREADS, PROBE, SOLO_LIMIT, ProblemData, Lookup>(a, slice, in_frontier);
    } else {
        const unsigned want = action == kSweepSparseEmit ? emit_grid : sparse_grid;  // (emitting: every workgroup ends with an atomic on one word)
        const unsigned limit = want < gridDim.x ? want : gridDim.x;
        if (blockIdx.x >= limit) return;
        SparseSweep<THREADS, PROBE, SOLO_LIMIT, ProblemData, Lookup, true>(a, slice, in_frontier, action == kSweepSparseEmit, limit);
    }
}

// ---- bitmap -> frontier queue (vertex, row start, degree prefix) through the FrontierWriter ----
// Every workgroup owns a contiguous share of the bitmap.  It first counts the share's set bits (all word loads in flight
// together, one reduction): an empty share ends there, and a share that fits the staging buffer -- every conversion back to
// top-down, whose frontier is a few thousand vertices at most -- is appended wave by wave with NO workgroup barrier per
// step and flushed once.  The conversion of a near-empty 2 MiB bitmap went 16 us -> the cost of reading it; only a share
// with more set bits than the buffer holds (the queue of a binned level: millions of vertices) takes the stepwise path
// with intermediate flushes.
template <int THREADS, typename VertexId, typename SizeT>
__global__ __launch_bounds__(THREADS) void BitmapToQueueKernel(const unsigned *d_bitmap, SizeT nodes,
                                                               util::Frontier<VertexId, SizeT> out,
                                                               unsigned long long *d_tail_out, int *d_overflow,
                                                               const SizeT *d_row_offsets)
{
    constexpr int CAPACITY = 16 * THREADS;
    constexpr int WAVES = THREADS / util::kWaveSize;
    typedef FrontierWriter<THREADS, CAPACITY, VertexId, SizeT> Writer;
    __shared__ typename Writer::Storage s_writer;
    __shared__ unsigned s_pop[WAVES];

    const long long words = (static_cast<long long>(nodes) + 31) / 32;
    const long long per_wg = ((words + gridDim.x - 1) / gridDim.x + THREADS - 1) / THREADS * THREADS;  // whole steps
    const long long w_begin = static_cast<long long>(blockIdx.x) * per_wg;
    const long long w_end = (w_begin + per_wg < words) ? w_begin + per_wg : words;
    if (w_begin >= w_end) return;  // (workgroup-uniform)

    unsigned pop = 0;
    for (long long w = w_begin + threadIdx.x; w < w_end; w += THREADS) pop += static_cast<unsigned>(__popc(d_bitmap[w]));
    pop = util::WaveSum(pop);
    if (util::LaneId() == 0) s_pop[threadIdx.x / util::kWaveSize] = pop;
    Writer::Init(s_writer);
    __syncthreads();
    unsigned share = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) share += s_pop[w];
    if (share == 0) return;  // (workgroup-uniform)

    auto put = [&](long long wi, unsigned bits, int pos) {
        while (bits) {
            const int b = __ffs(bits) - 1;
            bits &= bits - 1;
            const long long v = wi * 32 + b;
            s_writer.buf[pos++] = static_cast<VertexId>(v < nodes ? v : nodes - 1);  // (tail bits are never set)
        }
    };

    if (share <= static_cast<unsigned>(CAPACITY)) {
        for (long long base = w_begin; base < w_end; base += THREADS) {  // (uniform trip count: Reserve is wave-collective)
            const long long wi = base + threadIdx.x;
            const unsigned word = (wi < w_end) ? d_bitmap[wi] : 0u;
            const int pos = Writer::Reserve(s_writer, __popc(word));
            put(wi, word, pos);
        }
        __syncthreads();
        const int staged = Writer::Count(s_writer);
        __syncthreads();
        Writer::template Flush<true>(s_writer, staged, out, d_tail_out, d_overflow, d_row_offsets);
        return;
    }

    for (long long base = w_begin; base < w_end; base += THREADS) {
        const int pending = Writer::Count(s_writer);
        __syncthreads();
        // a step can append at most 32 * THREADS entries: flush first if that might not fit.  To keep the
        // staging buffer small the step is split in four 8-bit slices of every word.
        const long long wi = base + threadIdx.x;
        const unsigned word = (wi < w_end) ? d_bitmap[wi] : 0u;
        int carried = pending;
#pragma unroll
        for (int slice8 = 0; slice8 < 4; ++slice8) {
            if (carried > CAPACITY - 8 * THREADS) {
                Writer::template Flush<true>(s_writer, carried, out, d_tail_out, d_overflow, d_row_offsets);
                carried = 0;
            }
            const unsigned bits = word & (0xFFu << (8 * slice8));
            const int pos = Writer::Reserve(s_writer, __popc(bits));
            put(wi, bits, pos);
            __syncthreads();
            carried = Writer::Count(s_writer);
            __syncthreads();
        }
    }
    const int rest = Writer::Count(s_writer);
    __syncthreads();
    Writer::template Flush<true>(s_writer, rest, out, d_tail_out, d_overflow, d_row_offsets);
}

}  // namespace advance
}  // namespace oprtr
}  // namespace gunrock
