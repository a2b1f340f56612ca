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
//   * found vertices and their degrees are reduced per workgroup and added to the step's packed tail
//     (vertices | edges<<32), the same word the forward advance produces, so the enactor's heuristic and
//     statistics see one format.
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/oprtr/frontier_writer.hpp>
#include <gunrock/util/device_intrinsics.hpp>
#include <gunrock/util/error_utils.hpp>
#include <gunrock/util/frontier.hpp>

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

// Frontier membership tests for the bottom-up sweep.
// BitmapLookup: one bitmap indexed by vertex id (single GPU).
// StripedBitmapLookup: vertex-cut over P ranks, owner = v mod P, local id = v div P (the reference's only multi-GPU
// vestige, problem_base.cuh:185-210); the all-gathered bitmaps of the P ranks sit back to back, words_per_rank apart.
template <typename VertexId>
struct BitmapLookup {
    const unsigned *bits;
    __device__ __forceinline__ bool operator()(VertexId u) const
    {
        return (bits[static_cast<unsigned>(u) >> 5] >> (u & 31)) & 1u;
    }
};
template <typename VertexId>
struct StripedBitmapLookup {
    const unsigned *bits;
    unsigned parts;
    unsigned words_per_rank;
    __device__ __forceinline__ bool operator()(VertexId u) const
    {
        const unsigned owner = static_cast<unsigned>(u) % parts, local = static_cast<unsigned>(u) / parts;
        return (bits[owner * words_per_rank + (local >> 5)] >> (local & 31)) & 1u;
    }
};

template <typename VertexId, typename SizeT>
struct BottomUpArgs {
    SizeT nodes;
    const SizeT *d_inv_row_offsets;
    const VertexId *d_inv_column_indices;
    unsigned long long *d_frontier_out;     // next frontier bitmap, every word is written
    unsigned long long *d_visited;          // visited bitmap, owner-updated
    unsigned long long *d_tail_out;
    unsigned long long *d_tail_clear;
};

template <int THREADS, int PROBE, int SOLO_LIMIT, typename ProblemData, typename Lookup>
__global__ __launch_bounds__(THREADS) void BottomUpKernel(
    BottomUpArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> a, typename ProblemData::DataSlice slice,
    Lookup in_frontier)
{
    typedef typename ProblemData::VertexId VertexId;
    typedef typename ProblemData::SizeT SizeT;
    constexpr int WAVES = THREADS / util::kWaveSize;
    __shared__ unsigned long long s_total[WAVES];

    const int tid = threadIdx.x;
    const unsigned lane = util::LaneId();
    if (blockIdx.x == 0 && tid == 0 && a.d_tail_clear) *a.d_tail_clear = 0ull;

    const long long words = (static_cast<long long>(a.nodes) + 63) / 64;
    const long long wave0 = (static_cast<long long>(blockIdx.x) * THREADS + tid) / util::kWaveSize;
    const long long nwaves = static_cast<long long>(gridDim.x) * WAVES;
    const VertexId new_label = slice.iteration + 1;

    unsigned found_count = 0;
    unsigned found_edges = 0;
    for (long long w = wave0; w < words; w += nwaves) {
        const unsigned long long vis = a.d_visited[w];  // wave-uniform address
        const VertexId v = static_cast<VertexId>(w * 64 + lane);
        bool open = v < a.nodes && ((vis >> lane) & 1ull) == 0;
        unsigned long long found_mask = 0;
        if (__ballot(open) != 0) {  // wave-uniform: some vertex of this word is still unvisited
            SizeT pos = 0, end = 0;
            if (open) {
                pos = a.d_inv_row_offsets[v];
                end = a.d_inv_row_offsets[v + 1];
            }
            const SizeT degree = end - pos;
            open = open && degree > 0;
            VertexId parent = -1;

            // phase A/B: the lane probes PROBE edges at a time, all loads in flight, for up to SOLO_LIMIT edges
            for (int done = 0; done < SOLO_LIMIT; done += PROBE) {
                if (__ballot(open && parent < 0 && pos < end) == 0) break;  // wave-uniform
                VertexId nb[PROBE];
                bool fw[PROBE];
#pragma unroll
                for (int j = 0; j < PROBE; ++j)
                    nb[j] = (open && parent < 0 && pos + j < end) ? a.d_inv_column_indices[pos + j] : static_cast<VertexId>(-1);
#pragma unroll
                for (int j = 0; j < PROBE; ++j)
                    fw[j] = (nb[j] >= 0) ? in_frontier(nb[j]) : false;
#pragma unroll
                for (int j = 0; j < PROBE; ++j)
                    if (parent < 0 && fw[j]) parent = nb[j];
                pos += PROBE;
            }

            // phase C: rows still unresolved are swept by the whole wave, 64 in-edges per step
            unsigned long long todo = __ballot(open && parent < 0 && pos < end);
            while (todo) {
                const int leader = __ffsll(static_cast<long long>(todo)) - 1;
                SizeT p = __shfl(pos, leader, util::kWaveSize);
                const SizeT e = __shfl(end, leader, util::kWaveSize);
                VertexId hit_parent = -1;
                for (; p < e; p += util::kWaveSize) {
                    const SizeT mine = p + static_cast<SizeT>(lane);
                    VertexId u = -1;
                    if (mine < e) u = a.d_inv_column_indices[mine];
                    bool hit = false;
                    if (u >= 0) hit = in_frontier(u);
                    const unsigned long long hm = __ballot(hit);
                    if (hm) {
                        hit_parent = __shfl(u, __ffsll(static_cast<long long>(hm)) - 1, util::kWaveSize);
                        break;
                    }
                }
                if (static_cast<int>(lane) == leader) {
                    parent = hit_parent;
                    pos = end;
                }
                todo &= todo - 1;
            }

            const bool found = open && parent >= 0;
            if (found) {
                slice.d_labels[v] = new_label;
                if (ProblemData::MARK_PREDECESSORS) slice.d_preds[v] = parent;
                found_count += 1;
                found_edges += static_cast<unsigned>(degree);
            }
            found_mask = __ballot(found);
        }
        if (lane == 0) {
            a.d_frontier_out[w] = found_mask;
            if (found_mask) a.d_visited[w] = vis | found_mask;
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
        if (sum) atomicAdd(a.d_tail_out, sum);
    }
}

// ---- bitmap -> frontier queue (vertex, row start, degree prefix) through the FrontierWriter ----
template <int THREADS, typename VertexId, typename SizeT>
__global__ __launch_bounds__(THREADS) void BitmapToQueueKernel(const unsigned *d_bitmap, SizeT nodes,
                                                               util::Frontier<VertexId, SizeT> out,
                                                               unsigned long long *d_tail_out, int *d_overflow,
                                                               const SizeT *d_row_offsets)
{
    constexpr int CAPACITY = 16 * THREADS;
    typedef FrontierWriter<THREADS, CAPACITY, VertexId, SizeT> Writer;
    __shared__ typename Writer::Storage s_writer;
    Writer::Init(s_writer);
    __syncthreads();

    const long long words = (static_cast<long long>(nodes) + 31) / 32;
    const long long chunk = THREADS;  // words per workgroup step: up to 32*THREADS appends
    for (long long base = static_cast<long long>(blockIdx.x) * chunk; base < words; base += static_cast<long long>(gridDim.x) * chunk) {
        const int pending = Writer::Count(s_writer);
        __syncthreads();
        // a step can append at most 32 * THREADS entries: flush first if that might not fit.  To keep the
        // staging buffer small the step is split in four 8-bit slices of every word.
        const long long wi = base + threadIdx.x;
        const unsigned word = (wi < words) ? d_bitmap[wi] : 0u;
        int carried = pending;
#pragma unroll
        for (int slice8 = 0; slice8 < 4; ++slice8) {
            if (carried > CAPACITY - 8 * THREADS) {
                Writer::template Flush<true>(s_writer, carried, out, d_tail_out, d_overflow, d_row_offsets);
                carried = 0;
            }
            unsigned bits = (word >> (8 * slice8)) & 0xFFu;
            const int mine = __popc(bits);
            int pos = Writer::Reserve(s_writer, mine);
            while (bits) {
                const int b = __ffs(bits) - 1;
                bits &= bits - 1;
                const long long v = wi * 32 + 8 * slice8 + b;
                if (v < nodes) s_writer.buf[pos++] = static_cast<VertexId>(v);
                else s_writer.buf[pos++] = static_cast<VertexId>(nodes - 1);  // unreachable: tail bits are never set
            }
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
