// frontier_writer.hpp -- workgroup-cooperative frontier enqueue for gfx950.
//
// Takes the place of the reference's per-tile "CTA scan + one AtomicInt::Add on the queue counter"
// (gunrock/util/scan/cooperative_scan.cuh:152-173,289-323; filter/cta.cuh:518-543) with a scheme
// sized for 256 CUs: one returning atomic on a single word saturates near 88 ops/us on MI355X, so a
// per-512-element tile reservation would cap a scale-24 BFS level at ~1 ms.  Here every workgroup
//   1. appends surviving vertices to an LDS staging buffer with one LDS atomic per WAVE
//      (64-bit ballot + v_mbcnt rank),
//   2. flushes the buffer rarely (when it could overflow, and once at exit): loads the row extent of
//      every staged vertex, block-scans the packed (1, degree) pairs, makes ONE packed 64-bit
//      atomicAdd on the step's tail word, and writes vertex / row-start / degree-prefix entries.
// Because vertex slots and edge ranges are reserved by the same atomic, the degree prefix written
// here is already the device-wide exclusive scan the next load-balanced advance needs.
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/util/device_intrinsics.hpp>
#include <gunrock/util/frontier.hpp>

namespace gunrock {
namespace oprtr {

template <int THREADS, int CAPACITY, typename VertexId, typename SizeT>
struct FrontierWriter {
    static_assert(CAPACITY % THREADS == 0, "staging capacity must be a multiple of the workgroup size");
    static constexpr int PER_THREAD = CAPACITY / THREADS;
    typedef util::BlockScan<THREADS, unsigned long long> Scan;

    struct Storage {
        VertexId buf[CAPACITY];
        typename Scan::Storage scan;
        unsigned long long base;
        int count;
    };

    static __device__ __forceinline__ void Init(Storage &st)
    {
        if (threadIdx.x == 0) st.count = 0;
    }

    // Every lane of the wave must call this (pred = false for lanes with nothing to add).
    static __device__ __forceinline__ void Append(Storage &st, bool pred, VertexId v)
    {
        const unsigned long long m = __ballot(pred);
        if (m == 0) return;  // wave-uniform
        const int leader = __ffsll(static_cast<long long>(m)) - 1;
        int base = 0;
        if (static_cast<int>(util::LaneId()) == leader) base = atomicAdd(&st.count, __popcll(m));
        base = __shfl(base, leader, util::kWaveSize);
        if (pred) st.buf[base + util::RankInMask(m)] = v;
    }

    // Wave-aggregated reservation of `mine` slots per lane: one LDS atomic per wave.  Every lane of the wave
    // must call; returns the lane's first slot in buf.
    static __device__ __forceinline__ int Reserve(Storage &st, int mine)
    {
        const int incl = util::WaveInclusiveSum(mine);
        const int total = __shfl(incl, util::kWaveSize - 1, util::kWaveSize);
        int base = 0;
        if (total == 0) return 0;  // wave-uniform
        if (util::LaneId() == util::kWaveSize - 1) base = atomicAdd(&st.count, total);
        base = __shfl(base, util::kWaveSize - 1, util::kWaveSize);
        return base + incl - mine;
    }

    // Number of staged entries.  Call it between two workgroup barriers that separate it from any
    // Append (so every thread reads the same value), then hand it to Flush.
    static __device__ __forceinline__ int Count(const Storage &st) { return st.count; }

    // All threads of the workgroup must call, with the same `n` = Count() read as described above;
    // no Append may run concurrently.  The buffer is empty afterwards.  DROP_ZERO_DEGREE removes
    // vertices without out-edges from the queue (they can never contribute to an advance; their
    // labels were written at discovery).
    template <bool DROP_ZERO_DEGREE>
    static __device__ __forceinline__ void Flush(Storage &st, const int n,
                                                 const util::Frontier<VertexId, SizeT> &out,
                                                 unsigned long long *d_tail, int *d_overflow,
                                                 const SizeT *__restrict__ d_row_offsets)
    {
        if (n == 0) return;  // uniform

        VertexId v[PER_THREAD];
        SizeT rs[PER_THREAD];
        SizeT deg[PER_THREAD];
        unsigned long long mine = 0;
#pragma unroll
        for (int j = 0; j < PER_THREAD; ++j) {
            const int i = j * THREADS + threadIdx.x;
            deg[j] = -1;
            if (i < n) {
                v[j] = st.buf[i];
                rs[j] = d_row_offsets[v[j]];
                deg[j] = d_row_offsets[v[j] + 1] - rs[j];
                if (DROP_ZERO_DEGREE && deg[j] == 0) deg[j] = -1;
                if (deg[j] >= 0) mine += util::PackTail(1u, static_cast<unsigned>(deg[j]));
            }
        }
        unsigned long long total;
        const unsigned long long excl = Scan::ExclusiveSum(mine, total, st.scan);
        if (threadIdx.x == 0) {
            st.base = total ? atomicAdd(d_tail, total) : 0ull;
            st.count = 0;
        }
        __syncthreads();
        const unsigned long long block_base = st.base;
        if (static_cast<unsigned long long>(util::TailCount(block_base)) + util::TailCount(total) >
            static_cast<unsigned long long>(out.capacity)) {
            if (threadIdx.x == 0) *d_overflow = 1;  // reference: "Frontier queue overflow" (filter/cta.cuh:526-529)
            return;
        }
        unsigned pos = util::TailCount(block_base) + util::TailCount(excl);
        unsigned epos = util::TailEdges(block_base) + util::TailEdges(excl);
#pragma unroll
        for (int j = 0; j < PER_THREAD; ++j) {
            if (deg[j] >= 0) {
                out.v[pos] = v[j];
                out.row_start[pos] = rs[j];
                out.scan[pos] = static_cast<SizeT>(epos);
                ++pos;
                epos += static_cast<unsigned>(deg[j]);
            }
        }
    }

    // Flush for queues that feed a FILTER (or an edge-parallel operator) rather than an advance: only the ids are
    // written, the tail counts entries (edge half stays 0).  Same calling rules as Flush.
    static __device__ __forceinline__ void FlushIds(Storage &st, const int n, VertexId *d_out, SizeT capacity,
                                                    unsigned long long *d_tail, int *d_overflow)
    {
        if (n == 0) return;  // uniform
        if (threadIdx.x == 0) {
            st.base = atomicAdd(d_tail, static_cast<unsigned long long>(n));
            st.count = 0;
        }
        VertexId v[PER_THREAD];
#pragma unroll
        for (int j = 0; j < PER_THREAD; ++j) {
            const int i = j * THREADS + threadIdx.x;
            if (i < n) v[j] = st.buf[i];
        }
        __syncthreads();
        const unsigned long long base = util::TailCount(st.base);
        if (base + static_cast<unsigned long long>(n) > static_cast<unsigned long long>(capacity)) {
            if (threadIdx.x == 0) *d_overflow = 1;
            return;
        }
#pragma unroll
        for (int j = 0; j < PER_THREAD; ++j) {
            const int i = j * THREADS + threadIdx.x;
            if (i < n) d_out[base + i] = v[j];  // coalesced
        }
        __syncthreads();  // st.base may be rewritten by the next flush
    }
};

}  // namespace oprtr
}  // namespace gunrock
