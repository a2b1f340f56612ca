// oprtr/advance/binned.hpp -- destination-binned hand-off between the two phases of an atomic-free advance.
//
// Why.  The load-balanced advance ends every accepted edge in Functor::CondEdge, which for BFS is an agent-scope
// atomicOr on the visited bitmap.  On MI355X those atomics execute at the memory side at ~27 G/s whatever their scope
// (tools/atomic_scope_bench.hip) and, worse, the status screen in front of them goes stale: a bit set from another XCD
// never reaches this XCD's L2 copy of the line, so most edges into a vertex discovered earlier in the same (or the
// previous) level still pay the atomic.  Measured on R-MAT scale-24, level 1 of the largest-degree source (117.8 M edge
// slots, 67.9 M of them into vertices unvisited at the level's start, 6.6 M discoveries): expansion + screen 0.44 ms,
// the claims another 1.17 ms.  The reference has the same structure (one atomicCAS per edge, bfs_functor.cuh:56-58) and
// hides it behind its idempotent mode's best-effort byte mask (filter/cta.cuh:166-207).
//
// What.  Phase 1 (advance::BinnedExpandKernel, kernel.hpp) expands and screens exactly like the load-balanced advance
// but hands every surviving (source, destination) pair to the bin of the destination's OWNER XCD instead of claiming it.
// Phase 2 (BinnedApplyKernel, below) runs the functor's CondEdge / ApplyEdge for the pairs of bin x on workgroups that
// execute on XCD x.  All status bytes of a destination are then read and written through ONE XCD's L2 -- coherent
// without atomics -- so the claim becomes "L2 load, store if clear": duplicates inside the few hundred nanoseconds a store
// takes to land are possible and harmless (the closing vertex-ordered sweep of the primitive dedupes them, every
// same-level source is a valid parent).  Correctness never depends on the placement: a workgroup that has drained its
// home bin helps with the others.
//
// Bins.  Owner XCD of vertex v = XOR-fold of (v >> 8) to 3 bits: 256-vertex granules (two 128-byte lines of flag bytes)
// dealt to the 8 XCDs by a hash, because every single id bit of an R-MAT graph is skewed 3:1 (a bin made of three id bits
// would be 27x heavier than its lightest sibling; the folded bins of the bench graph differ by < 2 %).
// Storage.  A pool of 4096-entry chunks; a workgroup owns one open chunk per bin (plus the next one, so a tile never
// waits for an allocation), positions inside it come from LDS counters, and per-bin chunk lists tell phase 2 what to
// read.  Per tile and wave the 4 x 64 survivors are ranked with ONE packed wave prefix sum (8 bins x 16-bit counters in
// two 64-bit words) and one LDS atomic per bin: ~0.5 instructions per edge slot, against ~12 for a ballot per bin.
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/oprtr/advance/functor_hooks.hpp>
#include <gunrock/util/device_intrinsics.hpp>
#include <gunrock/util/error_utils.hpp>

namespace gunrock {
namespace oprtr {
namespace advance {

constexpr int kBins = 8;            // = XCDs of an MI355X
constexpr int kBinChunk = 4096;     // entries per chunk (16 KiB of destinations)
constexpr int kBinCtrlStride = 32;  // ints between control words (128 bytes: one line each)

__device__ __forceinline__ unsigned XccId()
{
    return __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11)) & 7u;
}

template <typename VertexId>
__host__ __device__ __forceinline__ int OwnerBin(VertexId v)
{
    unsigned x = static_cast<unsigned>(v) >> 8;
    x ^= x >> 12;
    x ^= x >> 6;
    x ^= x >> 3;
    return static_cast<int>(x & 7u);
}

// Device view of the chunk pool (by-value kernel argument).
template <typename VertexId>
struct BinPool {
    VertexId *d_dst = nullptr;      // [max_chunks * kBinChunk]
    VertexId *d_src = nullptr;      // same shape; nullptr when the functor does not look at sources
    int *d_chunk_count = nullptr;   // [max_chunks] entries used in each chunk
    int *d_bin_list = nullptr;      // [kBins][max_chunks] chunk ids of each bin, in allocation order
    int *d_ctrl = nullptr;          // control words, kBinCtrlStride apart: [0] chunks allocated, [1 + b] chunks of bin b,
                                    // [1 + kBins + b] phase-2 cursor of bin b; zeroed before every phase 1
    int *d_overflow = nullptr;      // raised when the pool ran out (sized so that it cannot: see BinPoolChunks)
    int max_chunks = 0;

    __device__ __forceinline__ int *PoolNext() const { return d_ctrl; }
    __device__ __forceinline__ int *BinCount(int b) const { return d_ctrl + (1 + b) * kBinCtrlStride; }
    __device__ __forceinline__ int *BinCursor(int b) const { return d_ctrl + (1 + kBins + b) * kBinCtrlStride; }
    static constexpr size_t CtrlInts() { return static_cast<size_t>(1 + 2 * kBins) * kBinCtrlStride; }
};

// Chunks that can be needed by a phase 1 that hands over at most `entries` pairs from `workgroups` workgroups: full chunks
// plus, per workgroup and bin, the open chunk and the pre-allocated next one.
inline long long BinPoolChunks(long long entries, int workgroups)
{
    return entries / kBinChunk + 2ll * workgroups * kBins + 64;
}

// LDS state of a workgroup's open chunks.
struct BinnerStorage {
    int fill[kBins];      // entries written to the open chunk (may run past kBinChunk into the next one until the tile ends)
    int chunk[kBins][2];  // [0] the open chunk, [1] the next one
};

// Workgroup-side writer of phase 1.
template <int THREADS, int ITEMS, typename VertexId, bool WITH_SRC>
struct Binner {
    static_assert(ITEMS * util::kWaveSize <= 0xFFFF, "16-bit per-bin wave counters");
    static_assert(THREADS * ITEMS <= kBinChunk, "a tile must fit into the open chunk plus the next one");
    typedef BinnerStorage Storage;

    static __device__ __forceinline__ int Grab(const BinPool<VertexId> &pool, int b)
    {
        int c = atomicAdd(pool.PoolNext(), 1);
        if (c >= pool.max_chunks) {  // cannot happen with BinPoolChunks sizing; stay in bounds and fail loudly
            *pool.d_overflow = 1;
            c = pool.max_chunks - 1;
        }
        return c;
    }

    // a chunk enters its bin's list when it is closed: phase 2 never sees empty or unused ones
    static __device__ __forceinline__ void Close(const BinPool<VertexId> &pool, int b, int c, int count)
    {
        if (count <= 0) return;
        pool.d_chunk_count[c] = count;
        const int i = atomicAdd(pool.BinCount(b), 1);
        if (i < pool.max_chunks) pool.d_bin_list[static_cast<size_t>(b) * pool.max_chunks + i] = c;
    }

    // whole workgroup; a barrier must follow before the first Put
    static __device__ __forceinline__ void Init(Storage &st, const BinPool<VertexId> &pool)
    {
        if (threadIdx.x < kBins) {
            st.fill[threadIdx.x] = 0;
            st.chunk[threadIdx.x][0] = Grab(pool, threadIdx.x);
            st.chunk[threadIdx.x][1] = Grab(pool, threadIdx.x);
        }
    }

    // Every lane of every wave calls once per tile with its ITEMS candidate pairs.
    static __device__ __forceinline__ void Put(Storage &st, const BinPool<VertexId> &pool, const VertexId (&src)[ITEMS],
                                               const VertexId (&dst)[ITEMS], const bool (&live)[ITEMS])
    {
        const unsigned lane = util::LaneId();
        int bin[ITEMS];
        // packed per-bin counts of this lane: word w holds bins 2w (low half) and 2w + 1 (high half), 16 bits each; the
        // halves never overflow (<= ITEMS x 64 per wave), so the four words scan independently on the DPP path
        unsigned cnt[kBins / 2] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            bin[k] = live[k] ? OwnerBin(dst[k]) : 0;
            const unsigned inc = live[k] ? (1u << ((bin[k] & 1) * 16)) : 0u;
#pragma unroll
            for (int w = 0; w < kBins / 2; ++w) cnt[w] += ((bin[k] >> 1) == w) ? inc : 0u;
        }
        unsigned incl[kBins / 2], total[kBins / 2];
        unsigned any = 0;
#pragma unroll
        for (int w = 0; w < kBins / 2; ++w) {
            incl[w] = util::WaveInclusiveSumDpp(cnt[w]);
            total[w] = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(incl[w]), util::kWaveSize - 1));
            any |= total[w];
        }
        if (any == 0) return;  // wave-uniform
        int base = 0;
        if (lane < kBins) {  // one LDS atomic per bin and wave
            unsigned t = total[0];
#pragma unroll
            for (int w = 1; w < kBins / 2; ++w) t = (static_cast<int>(lane >> 1) == w) ? total[w] : t;
            const int mine = static_cast<int>((t >> ((lane & 1) * 16)) & 0xFFFFu);
            if (mine) base = atomicAdd(&st.fill[lane], mine);
        }
        unsigned excl[kBins / 2];  // exclusive over lanes; grows over this lane's own items
#pragma unroll
        for (int w = 0; w < kBins / 2; ++w) excl[w] = incl[w] - cnt[w];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int wave_first = __shfl(base, bin[k], util::kWaveSize);  // (all lanes: dead ones read bin 0)
            if (live[k]) {
                const int shift = (bin[k] & 1) * 16;
                unsigned e = excl[0];
#pragma unroll
                for (int w = 1; w < kBins / 2; ++w) e = ((bin[k] >> 1) == w) ? excl[w] : e;
                const int pos = wave_first + static_cast<int>((e >> shift) & 0xFFFFu);
                const int chunk = st.chunk[bin[k]][pos >= kBinChunk ? 1 : 0];
                const size_t at = static_cast<size_t>(chunk) * kBinChunk + (pos & (kBinChunk - 1));
                pool.d_dst[at] = dst[k];
                if (WITH_SRC) pool.d_src[at] = src[k];
#pragma unroll
                for (int w = 0; w < kBins / 2; ++w) excl[w] += ((bin[k] >> 1) == w) ? (1u << shift) : 0u;
            }
        }
    }

    // After the barrier that ends a tile: rotate the chunks that filled up.  The next Put must be separated from this by a
    // barrier (the staging barrier of the next tile).
    static __device__ __forceinline__ void EndTile(Storage &st, const BinPool<VertexId> &pool)
    {
        if (threadIdx.x < kBins && st.fill[threadIdx.x] >= kBinChunk) {
            const int b = threadIdx.x;
            Close(pool, b, st.chunk[b][0], kBinChunk);
            st.chunk[b][0] = st.chunk[b][1];
            st.fill[b] -= kBinChunk;
            st.chunk[b][1] = Grab(pool, b);
        }
    }

    // After the last tile's barrier (and EndTile).
    static __device__ __forceinline__ void Finish(Storage &st, const BinPool<VertexId> &pool)
    {
        if (threadIdx.x < kBins) {
            Close(pool, threadIdx.x, st.chunk[threadIdx.x][0], st.fill[threadIdx.x]);
        }
    }
};

// Phase 2: run [ScreenEdge,] CondEdge, ApplyEdge for every binned pair, each bin on the XCD that owns it (home bin first, then
// help).  ScreenEdge (side-effect free) is evaluated for a whole batch first so the status loads of a batch are in flight
// together -- CondEdge's store may alias the next pair's load, which would otherwise chain the round trips.  The kernel gets
// its own DataSlice copy, so a functor can tell the phases apart.
// The functor sees e_id = e_id_in = 0: a binned advance is for functors that only need (source, destination).
template <int THREADS, typename ProblemData, typename Functor, bool WITH_SRC>
__global__ __launch_bounds__(THREADS) void BinnedApplyKernel(BinPool<typename ProblemData::VertexId> pool,
                                                             typename ProblemData::DataSlice slice)
{
    typedef typename ProblemData::VertexId VertexId;
    // Every WAVE works on its own: it draws a chunk, sweeps it in batches of 64 x BATCH pairs with all loads of a batch in
    // flight together, and never meets the other waves at a barrier (a round trip to memory costs microseconds under load;
    // what this kernel needs is many of them overlapping).  The next chunk index is drawn before the current chunk is swept.
    constexpr int BATCH = WITH_SRC ? 8 : 16;
    const unsigned lane = util::LaneId();
    const unsigned home = XccId();  // (measured: draining bins regardless of the XCD costs 1.75x -- 720 vs 411 us for 68 M pairs)
    for (int r = 0; r < kBins; ++r) {
        const int b = static_cast<int>((home + r) & (kBins - 1));
        const int listed = *pool.BinCount(b);
        const int chunks = listed < pool.max_chunks ? listed : pool.max_chunks;
        const int *list = pool.d_bin_list + static_cast<size_t>(b) * pool.max_chunks;
        // (a glance before drawing: the cursors are eight hot addresses, ~88 atomics per microsecond each)
        if (__hip_atomic_load(pool.BinCursor(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= chunks) continue;
        int next = 0;
        if (lane == 0) next = atomicAdd(pool.BinCursor(b), 1);
        for (;;) {
            const int i = __builtin_amdgcn_readfirstlane(next);
            if (i >= chunks) break;  // wave-uniform
            if (lane == 0) next = atomicAdd(pool.BinCursor(b), 1);  // (its latency hides behind this chunk's sweep)
            const int c = list[i];
            const int count = pool.d_chunk_count[c];
            const VertexId *dst = pool.d_dst + static_cast<size_t>(c) * kBinChunk;
            const VertexId *src = WITH_SRC ? pool.d_src + static_cast<size_t>(c) * kBinChunk : nullptr;
            for (int j0 = 0; j0 < count; j0 += util::kWaveSize * BATCH) {
                VertexId d[BATCH], s[BATCH];
                bool ok[BATCH];
#pragma unroll
                for (int k = 0; k < BATCH; ++k) {
                    const int j = j0 + k * util::kWaveSize + static_cast<int>(lane);
                    ok[k] = j < count;
                    d[k] = ok[k] ? __builtin_nontemporal_load(dst + j) : 0;
                    s[k] = (WITH_SRC && ok[k]) ? __builtin_nontemporal_load(src + j) : 0;
                }
                if constexpr (HasScreenEdge<Functor, VertexId, typename ProblemData::DataSlice>::value) {
#pragma unroll
                    for (int k = 0; k < BATCH; ++k) ok[k] = ok[k] & Functor::ScreenEdge(s[k], d[k], &slice, 0, 0);  // (no branch)
                }
#pragma unroll
                for (int k = 0; k < BATCH; ++k) ok[k] = ok[k] && Functor::CondEdge(s[k], d[k], &slice, 0, 0);
#pragma unroll
                for (int k = 0; k < BATCH; ++k)
                    if (ok[k]) Functor::ApplyEdge(s[k], d[k], &slice, 0, 0);
            }
        }
    }
}

// Host side of the pool: owned by the problem that uses a binned advance.
template <typename VertexId>
struct BinPoolStorage {
    BinPool<VertexId> view;
    long long capacity_entries = 0;
    int workgroups = 0;

    bool Ready() const { return view.d_dst != nullptr; }

    hipError_t Allocate(long long max_entries, int max_workgroups, bool with_src, int *d_overflow)
    {
        hipError_t retval = hipSuccess;
        Release();
        const long long chunks = BinPoolChunks(max_entries, max_workgroups);
        if (chunks > 0x7FFFFFFFll / 2) return util::GRError(hipErrorInvalidValue, "BinPool: too many chunks", __FILE__, __LINE__);
        view.max_chunks = static_cast<int>(chunks);
        const size_t entries = static_cast<size_t>(chunks) * kBinChunk;
        GR_CHECK(hipMalloc(&view.d_dst, sizeof(VertexId) * entries), "BinPool hipMalloc d_dst failed");
        if (with_src) GR_CHECK(hipMalloc(&view.d_src, sizeof(VertexId) * entries), "BinPool hipMalloc d_src failed");
        GR_CHECK(hipMalloc(&view.d_chunk_count, sizeof(int) * static_cast<size_t>(chunks)), "BinPool hipMalloc failed");
        GR_CHECK(hipMalloc(&view.d_bin_list, sizeof(int) * static_cast<size_t>(chunks) * kBins), "BinPool hipMalloc failed");
        GR_CHECK(hipMalloc(&view.d_ctrl, sizeof(int) * BinPool<VertexId>::CtrlInts()), "BinPool hipMalloc failed");
        view.d_overflow = d_overflow;
        capacity_entries = max_entries;
        workgroups = max_workgroups;
        return retval;
    }

    hipError_t Arm(hipStream_t stream)  // before every phase 1
    {
        return util::GRError(hipMemsetAsync(view.d_ctrl, 0, sizeof(int) * BinPool<VertexId>::CtrlInts(), stream),
                             "BinPool arm failed", __FILE__, __LINE__);
    }

    void Release()
    {
        if (view.d_dst) util::GRError(hipFree(view.d_dst), "BinPool hipFree failed", __FILE__, __LINE__);
        if (view.d_src) util::GRError(hipFree(view.d_src), "BinPool hipFree failed", __FILE__, __LINE__);
        if (view.d_chunk_count) util::GRError(hipFree(view.d_chunk_count), "BinPool hipFree failed", __FILE__, __LINE__);
        if (view.d_bin_list) util::GRError(hipFree(view.d_bin_list), "BinPool hipFree failed", __FILE__, __LINE__);
        if (view.d_ctrl) util::GRError(hipFree(view.d_ctrl), "BinPool hipFree failed", __FILE__, __LINE__);
        view = BinPool<VertexId>();
        capacity_entries = 0;
        workgroups = 0;
    }
};

}  // namespace advance
}  // namespace oprtr
}  // namespace gunrock
