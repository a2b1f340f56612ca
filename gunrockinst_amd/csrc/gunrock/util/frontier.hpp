// frontier.hpp -- device frontier queues and the work-progress counters that go with them.
//
// MI355X-first redesign of the reference's ping-pong frontier queues
// (gunrock/app/problem_base.cuh:73-140 GraphSlice::frontier_queues, util/multiple_buffering.cuh:92-131)
// and of CtaWorkProgress (gunrock/util/cta_work_progress.cuh:51-360):
//
//  * A frontier is three parallel arrays: vertex id, the vertex's first edge (row start) and the
//    EXCLUSIVE prefix sum of degrees in queue order.  The prefix is produced by whoever WRITES the
//    frontier (FrontierWriter in oprtr/frontier_writer.hpp) from one packed 64-bit reservation per
//    flushed batch, so the load-balanced advance needs no separate degree-gather, device-wide scan or
//    sorted-search pass per BFS level (reference: GetEdgeCounts + mgpu::Scan + MarkPartitionSizes +
//    mgpu::SortedSearch, oprtr/advance/kernel.cuh:300-368) and no blocking 4-byte length read inside
//    the operator (advance/kernel.cuh:315-317).
//  * One packed counter per BSP step: low 32 bits = queued vertices, high 32 bits = queued edges.
//    Counters form a ring of 4 like the reference's queue-length slots; the kernel of step k writes
//    slot (k+1)&3 and clears slot (k+2)&3 (cta_work_progress.cuh:55-72 protocol).
#pragma once

#include <hip/hip_runtime.h>
#include <set>
#include <cstdint>
#include <cstring>

#include <gunrock/util/error_utils.hpp>

namespace gunrock {
namespace util {

template <typename VertexId, typename SizeT>
struct Frontier {
    VertexId *v = nullptr;   // vertex ids
    SizeT *row_start = nullptr;  // first edge of each vertex (offset into column_indices)
    SizeT *scan = nullptr;   // exclusive prefix sum of degrees, in queue order
    SizeT capacity = 0;
};

__host__ __device__ __forceinline__ unsigned long long PackTail(unsigned count, unsigned edges)
{
    return (static_cast<unsigned long long>(edges) << 32) | count;
}
__host__ __device__ __forceinline__ unsigned TailCount(unsigned long long t) { return static_cast<unsigned>(t); }
__host__ __device__ __forceinline__ unsigned TailEdges(unsigned long long t) { return static_cast<unsigned>(t >> 32); }

// Host wait for the stream: spin on hipStreamQuery instead of hipStreamSynchronize.  The BSP loop waits dozens of times
// per search for kernels that run 10-100 us; a blocking wait adds wake-up latency to each of them.
inline hipError_t SpinSync(hipStream_t stream)
{
    hipError_t rc;
    while ((rc = hipStreamQuery(stream)) == hipErrorNotReady) {
    }
    return rc;
}

// The calling workgroup's line of a "wide" tail (see WorkProgress::d_wide); nullptr stays nullptr.
__device__ __forceinline__ unsigned long long *WideTailSlot(unsigned long long *d_wide)
{
    return d_wide ? d_wide + (blockIdx.x & 31u) * 16u : nullptr;
}

// Pinned, device-mapped host block the PublishKernel writes straight into (no copy engine, no blit kernel per word).
struct HostMailbox {
    unsigned long long seq;       // written last, system-scope release
    unsigned long long tail[8];   // WorkProgress::d_tail
    unsigned long long sums[2];   // WorkProgress::d_sums
    unsigned long long wide;      // packed sum of the wide tail lines (which the kernel clears again)
    unsigned long long overflow;  // *d_overflow
    unsigned long long set_value; // staging word of WorkProgress::SetTail (host -> device)
    unsigned long long wide_set[8];  // packed sums of the further wide-counter sets (chained bottom-up sweeps; [0] unused: `wide`)
    int chain_log[8];             // action each sweep of the last chain took (oprtr/advance/bottom_up.hpp SweepAction)
};

// One wave: mirror the enactor's device words into the mailbox, fold + re-arm the wide tail, then publish `seq`.
static __global__ void PublishKernel(unsigned long long *d_tail, const unsigned long long *d_sums, unsigned long long *d_wide,
                                     const int *d_overflow, HostMailbox *box, unsigned long long seq, unsigned clear_mask,
                                     unsigned ones_mask, int wide_sets, const int *d_chain_log)
{
    // clear_mask / ones_mask: slots to zero / to set to all ones AFTER mirroring -- the re-arming an enactor
    // would otherwise do with one hipMemsetAsync per word before its next kernel
    const unsigned lane = threadIdx.x;
    if (lane < 8) {
        const unsigned long long v = d_tail[lane];
        box->tail[lane] = v;
        if (lane == 7 || ((clear_mask >> lane) & 1u)) d_tail[lane] = 0ull;  // slot 7: grid barrier of the persistent levels kernel
        else if ((ones_mask >> lane) & 1u) d_tail[lane] = ~0ull;
    }
    if (lane < 2) box->sums[lane] = d_sums[lane];
    unsigned long long w = 0;
    if (lane < 32) {
        w = d_wide[lane * 16];
        if (w) d_wide[lane * 16] = 0;
    }
    for (int o = 16; o; o >>= 1) w += __shfl_xor(w, o, 64);  // packed halves add independently (counts stay below 2^32)
    if (lane == 0) {
        box->wide = w;
        box->overflow = static_cast<unsigned long long>(*d_overflow);
    }
    if (wide_sets > 1) {  // (uniform) the sets of a sweep chain: all loads first, then fold, mirror and re-arm each
        unsigned long long ws[7];
#pragma unroll
        for (int k = 1; k < 8; ++k) ws[k - 1] = (k < wide_sets && lane < 32) ? d_wide[k * 512 + lane * 16] : 0ull;
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            if (k < wide_sets) {
                if (ws[k - 1]) d_wide[k * 512 + lane * 16] = 0ull;
                unsigned long long t = ws[k - 1];
                for (int o = 16; o; o >>= 1) t += __shfl_xor(t, o, 64);
                if (lane == 0) box->wide_set[k] = t;
            }
        }
        if (lane < 8) box->chain_log[lane] = d_chain_log[lane];
    }
    __threadfence_system();
    __syncthreads();
    if (lane == 0) __hip_atomic_store(&box->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One wave: what WorkProgress::Reset + SetTail do with three fill blits and one copy blit (4-5 us each, serialised at the
// start of every Enact) -- zero the ring, the overflow flag and the wide tail, then seed one ring slot.
static __global__ void ArmKernel(unsigned long long *d_tail, int *d_overflow, unsigned long long *d_wide, int slots, int wide_lines,
                                 int wide_stride, int seed_slot, unsigned long long seed_value, int wide_sets)
{
    const int lane = threadIdx.x;
    if (lane < slots) d_tail[lane] = (lane == seed_slot) ? seed_value : 0ull;
    if (lane < wide_lines)
        for (int k = 0; k < wide_sets; ++k) d_wide[(k * wide_lines + lane) * wide_stride] = 0ull;
    if (lane == 0) *d_overflow = 0;
}

// Device words shared by all kernels of one enactor.
struct WorkProgress {
    static constexpr int kSlots = 8;       // 0..3: BSP ring, 4: auxiliary tail, 5: SSSP far-min, 6: tail-kernel level count,
                                           // 7: grid barrier (counter | timeout); tail-kernel sums live in d_sums[2]
    static constexpr int kAux = 4;
    unsigned long long *d_tail = nullptr;  // [kSlots] packed (edges<<32 | vertices)
    int *d_overflow = nullptr;             // set when a writer ran out of queue capacity
    unsigned long long *d_sums = nullptr;  // [2] tail kernel: summed frontier lengths / edges
    // "Wide" tail: kWideLines packed counters, 128 bytes apart.  Atomics on ONE address retire at ~80 per microsecond on
    // MI355X (measured: 8192 wave-level adds = 103 us, tools/xcd_store_bench.hip), so kernels whose every workgroup reports
    // a count (bottom-up sweep, fresh-flag pass) spread their adds over these lines; PublishKernel folds and clears them.
    static constexpr int kWideLines = 32;
    static constexpr int kWideStride = 16;  // in 8-byte words
    // kWideSets such sets back to back: set 0 is "the" wide tail; sets 1.. take the finds of the sweeps of a chain (bottom_up.hpp)
    static constexpr int kWideSets = 8;
    static constexpr int kWideSetWords = kWideLines * kWideStride;
    unsigned long long *d_wide = nullptr;
    int *d_chain_log = nullptr;            // [8] action each sweep of a chain took; [8], [9]: gate words of the kernels queued behind
                                           // a chain (did the closing levels run; how many sweeps ran)
    int publish_sets = 1;                  // wide sets PublishKernel folds (enactors that chain sweeps raise it)
    // Host view.  Every blocking read-back of a BSP step is ONE tiny kernel that writes the mailbox in pinned host memory and
    // a host spin on its sequence word: the previous form (one or two hipMemcpyAsync = blit kernels of 4-5 us each, then a
    // stream query loop) cost ~20 us per level.
    HostMailbox *box = nullptr;            // pinned + mapped
    unsigned long long *h_sums = nullptr;  // = box->sums
    unsigned long long *h_tail = nullptr;  // = box->tail
    unsigned long long seq = 0;

    // WorkProgress objects whose device words exist.  A problem that arms an enactor's words from its own Reset kernel
    // (BFSProblem) keeps a pointer to them across searches; it asks here whether the enactor is still alive before using it.
    static std::set<const WorkProgress *> &Live()
    {
        static std::set<const WorkProgress *> *live = new std::set<const WorkProgress *>();  // (never destroyed: an enactor may be
                                                                                             //  released during static destruction)
        return *live;
    }
    static bool IsLive(const WorkProgress *p) { return p && Live().count(p) != 0; }

    hipError_t Init()
    {
        hipError_t retval = hipSuccess;
        if (d_tail) return retval;
        Live().insert(this);
        GR_CHECK(hipMalloc(&d_tail, sizeof(unsigned long long) * kSlots), "WorkProgress hipMalloc d_tail failed");
        GR_CHECK(hipMalloc(&d_overflow, sizeof(int)), "WorkProgress hipMalloc d_overflow failed");
        GR_CHECK(hipHostMalloc(reinterpret_cast<void **>(&box), sizeof(HostMailbox), hipHostMallocMapped),
                 "WorkProgress hipHostMalloc failed");
        std::memset(box, 0, sizeof(HostMailbox));
        h_tail = box->tail;
        h_sums = box->sums;
        GR_CHECK(hipMalloc(&d_sums, sizeof(unsigned long long) * 2), "WorkProgress hipMalloc d_sums failed");
        GR_CHECK(hipMemset(d_sums, 0, sizeof(unsigned long long) * 2), "WorkProgress memset failed");
        GR_CHECK(hipMalloc(&d_wide, sizeof(unsigned long long) * kWideSets * kWideSetWords), "WorkProgress hipMalloc d_wide failed");
        GR_CHECK(hipMalloc(&d_chain_log, sizeof(int) * 16), "WorkProgress hipMalloc d_chain_log failed");
        GR_CHECK(hipMemset(d_chain_log, 0, sizeof(int) * 16), "WorkProgress memset failed");
        // Blocking clears: the enactors work on non-blocking streams, which are NOT ordered behind the null stream -- an
        // asynchronous clear issued here could land after the first search's seed (seen: a search that found nothing).
        GR_CHECK(hipMemset(d_tail, 0, sizeof(unsigned long long) * kSlots), "WorkProgress memset failed");
        GR_CHECK(hipMemset(d_overflow, 0, sizeof(int)), "WorkProgress memset failed");
        GR_CHECK(hipMemset(d_wide, 0, sizeof(unsigned long long) * kWideSets * kWideSetWords), "WorkProgress memset failed");
        GR_CHECK(hipDeviceSynchronize(), "WorkProgress init sync failed");
        return retval;
    }

    hipError_t Reset(hipStream_t stream)
    {
        hipError_t retval = hipSuccess;
        GR_CHECK(hipMemsetAsync(d_tail, 0, sizeof(unsigned long long) * kSlots, stream), "WorkProgress memset failed");
        GR_CHECK(hipMemsetAsync(d_overflow, 0, sizeof(int), stream), "WorkProgress memset failed");
        GR_CHECK(hipMemsetAsync(d_wide, 0, sizeof(unsigned long long) * kWideSets * kWideSetWords, stream),
                 "WorkProgress memset failed");
        return retval;
    }

    // Reset + SetTail(slot, count, edges) in one launch
    hipError_t ResetWithTail(int slot, unsigned count, unsigned edges, hipStream_t stream)
    {
        box->overflow = 0;  // (a new search starts: forget the last search's flag)
        hipLaunchKernelGGL(ArmKernel, dim3(1), dim3(64), 0, stream, d_tail, d_overflow, d_wide, kSlots, kWideLines, kWideStride, slot & 3,
                           PackTail(count, edges), kWideSets);
        return GRError(hipGetLastError(), "WorkProgress ArmKernel launch failed", __FILE__, __LINE__);
    }

    hipError_t SetTail(int slot, unsigned count, unsigned edges, hipStream_t stream)
    {
        box->set_value = PackTail(count, edges);
        box->overflow = 0;  // (a new search starts: forget the last search's flag)
        return GRError(hipMemcpyAsync(d_tail + (slot & 3), &box->set_value, sizeof(unsigned long long), hipMemcpyHostToDevice, stream),
                       "WorkProgress SetTail failed", __FILE__, __LINE__);
    }

    // Mirror all device words into the mailbox and wait for them (the only host<->device sync of a BSP step).
    hipError_t Sync(hipStream_t stream, unsigned clear_mask = 0, unsigned ones_mask = 0)
    {
        hipError_t retval = hipSuccess;
        ++seq;
        hipLaunchKernelGGL(PublishKernel, dim3(1), dim3(64), 0, stream, d_tail, d_sums, d_wide, d_overflow, box, seq, clear_mask,
                           ones_mask, publish_sets, d_chain_log);
        GR_CHECK(hipGetLastError(), "WorkProgress PublishKernel launch failed");
        volatile unsigned long long *flag = &box->seq;
        unsigned spins = 0;
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
            if ((++spins & 0x3FFu) == 0) {  // now and then: did the stream fail, or finish without us seeing the flag yet?
                hipError_t rc = hipStreamQuery(stream);
                if (rc == hipSuccess) {
                    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
                    continue;
                }
                if (rc != hipErrorNotReady) return GRError(rc, "WorkProgress Sync: stream failed", __FILE__, __LINE__);
            }
        }
        return retval;
    }

    // Blocking read of one slot.
    hipError_t GetTail(int slot, unsigned &count, unsigned &edges, hipStream_t stream)
    {
        hipError_t retval = hipSuccess;
        GR_CHECK(Sync(stream), "WorkProgress GetTail sync failed");
        count = TailCount(h_tail[slot & 3]);
        edges = TailEdges(h_tail[slot & 3]);
        return retval;
    }

    // Blocking read of one ring slot plus the wide counters; (count, edges) = their sum.
    hipError_t GetTailWide(int slot, unsigned &count, unsigned &edges, hipStream_t stream)
    {
        hipError_t retval = hipSuccess;
        GR_CHECK(Sync(stream), "WorkProgress GetTailWide sync failed");
        count = TailCount(h_tail[slot & 3]) + TailCount(box->wide);
        edges = TailEdges(h_tail[slot & 3]) + TailEdges(box->wide);
        return retval;
    }

    // One blocking read of the whole ring + the tail kernel's outputs.
    hipError_t GetAll(hipStream_t stream) { return Sync(stream); }

    // slot 7: grid-barrier counter (low word) and its timeout flag (high word) of the persistent levels kernel
    unsigned *BarrierCounter() { return reinterpret_cast<unsigned *>(d_tail + 7); }
    int *BarrierTimeout() { return reinterpret_cast<int *>(d_tail + 7) + 1; }
    bool HostBarrierTimedOut() const { return (h_tail[7] >> 32) != 0; }
    int *LevelsDone() { return reinterpret_cast<int *>(d_tail + 6); }
    int HostLevelsDone() const { return *reinterpret_cast<const int *>(h_tail + 6); }

    unsigned long long *AuxTail() { return d_tail + kAux; }
    hipError_t ClearAux(hipStream_t stream)
    {
        return GRError(hipMemsetAsync(d_tail + kAux, 0, sizeof(unsigned long long), stream),
                       "WorkProgress ClearAux failed", __FILE__, __LINE__);
    }
    hipError_t GetAux(unsigned &count, unsigned &edges, hipStream_t stream)
    {
        hipError_t retval = hipSuccess;
        GR_CHECK(Sync(stream), "WorkProgress GetAux sync failed");
        count = TailCount(h_tail[kAux]);
        edges = TailEdges(h_tail[kAux]);
        return retval;
    }

    // Overflow flag as of the last Sync (every enactor loop ends on a Sync that follows its last kernel).
    bool OverflowAtLastSync() const { return box->overflow != 0; }

    hipError_t CheckOverflow(bool &overflow, hipStream_t stream)
    {
        hipError_t retval = hipSuccess;
        GR_CHECK(Sync(stream), "WorkProgress CheckOverflow sync failed");
        overflow = OverflowAtLastSync();
        return retval;
    }

    void Release()
    {
        Live().erase(this);
        if (d_tail) GRError(hipFree(d_tail), "WorkProgress hipFree failed", __FILE__, __LINE__);
        if (d_overflow) GRError(hipFree(d_overflow), "WorkProgress hipFree failed", __FILE__, __LINE__);
        if (box) GRError(hipHostFree(box), "WorkProgress hipHostFree failed", __FILE__, __LINE__);
        if (d_sums) GRError(hipFree(d_sums), "WorkProgress hipFree failed", __FILE__, __LINE__);
        if (d_wide) GRError(hipFree(d_wide), "WorkProgress hipFree failed", __FILE__, __LINE__);
        if (d_chain_log) GRError(hipFree(d_chain_log), "WorkProgress hipFree failed", __FILE__, __LINE__);
        d_chain_log = nullptr;
        d_tail = nullptr; d_overflow = nullptr; box = nullptr; h_tail = nullptr; h_sums = nullptr; d_sums = nullptr; d_wide = nullptr;
    }
};

}  // namespace util
}  // namespace gunrock
