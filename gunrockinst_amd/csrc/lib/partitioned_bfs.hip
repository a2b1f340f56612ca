// lib/partitioned_bfs.hip -- per-rank (one GPU) building blocks of the vertex-partitioned multi-GPU BFS.
//
// The reference is single-GPU (multi-GPU is a TODO: gunrock/app/problem_base.cuh:336-338, bfs_problem.cuh:171-173);
// its only multi-GPU rule is the striped ownership owner = v mod P, local row = v div P
// (problem_base.cuh:185-210, filter/cta.cuh:219,241).  This file keeps that rule.  One process per GPU holds the CSR
// rows of the vertices it owns (LOCAL row ids, GLOBAL column ids), their labels and visited bits.  A BSP level is
//     top-down : local advance (claim each destination once per rank in a global "sent" bitmap) -> bucket the claimed
//                ids by owner -> [all-to-all over RCCL, done by the caller] -> filter the received ids against the
//                local visited bitmap, label them, build the next local frontier;
//     bottom-up: [all-gather of the per-rank frontier bitmaps, done by the caller] -> local sweep of unvisited owned
//                vertices against the gathered bitmap (no id exchange at all).
// Two ways to drive it:
//   * grx_pbfs_search -- the whole level loop in C++ (Pbfs::Search): one host call per search, the exchange through a
//     Transport.  RcclTransport issues ncclAllGather / grouped ncclSend+ncclRecv (all 7 xGMI links of a GPU at once) on the
//     engine's own stream, so kernels and collectives are ordered by the stream and the host waits only where it needs a
//     number: twice per top-down level (the P x P count matrix, the global frontier size), once per bottom-up level (the
//     frontier sizes that ride in the trailing words of the gathered bitmaps).  RCCL is loaded with dlopen at the first
//     use, so the library itself does not link against it.  CallbackTransport hands the same three exchanges to the
//     caller (tests: several ranks sharing one GPU over gloo).
//   * the step-wise entry points (grx_pbfs_advance_local, ...) with the collectives in the caller
//     (gunrockinst_amd/multi_gpu.py): the executable model of the protocol that the CPU tests run over gloo.
// With mark_pred the top-down exchange carries (local id, parent) pairs and the bottom-up sweep records the parent it
// found, so the assembled predecessors are valid BFS parents on any number of ranks.
#include <gunrock/gunrock_mi355x.h>

#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <gunrock/app/bfs/bfs_problem.hpp>
#include <gunrock/app/enactor_base.hpp>
#include <gunrock/app/problem_base.hpp>
#include <gunrock/graphio/device_sort.hpp>
#include <gunrock/oprtr/advance/bottom_up.hpp>
#include <gunrock/oprtr/advance/kernel.hpp>
#include <gunrock/oprtr/filter/kernel.hpp>
#include <gunrock/util/memset_kernel.hpp>

using namespace gunrock;

namespace {

struct ncclUniqueIdBytes {  // layout of ncclUniqueId (rccl.h: char internal[128]), passed by value to ncclCommInitRank
    char internal[128];
};

struct PbfsProblem {
    typedef int VertexId;
    typedef int SizeT;
    typedef int Value;
    static constexpr bool MARK_PREDECESSORS = true;  // (the bottom-up sweep records the parent it finds: one store per discovery)
    struct DataSlice {
        int *d_labels;              // local
        int *d_preds;               // local: GLOBAL id of the parent (-1 source, -2 unreached)
        unsigned *d_visited_mask;   // local ids
        unsigned *d_sent_mask;      // GLOBAL ids: destinations this rank has already forwarded
        int iteration;
        int defer_labels = 0;       // (BFSProblem's deferred labelling is not used here: the sweeps write labels themselves)
        int *d_pred_global;         // mark_pred: GLOBAL ids -> the local source that forwarded it (as a global id), else nullptr
        const int *d_recv_preds;    // mark_pred: parents that arrived with the ids being filtered, else nullptr
        int parts, rank;
        // count-only ("marked") top-down level: one byte per GLOBAL vertex in OWNER-MAJOR order -- vertex v lives at
        // (v mod parts) * mark_stride + v div parts -- so that what goes to one owner is one contiguous slice
        unsigned char *d_mark = nullptr;
        unsigned mark_stride = 0;            // bytes (= bits of the exchanged bitmap) per owner slice
        unsigned long long mark_magic = 0;   // v div parts by a reciprocal multiplication (as StripedBitmapLookup, bottom_up.hpp)
        unsigned mark_shift = 0;
    };
};

// Count-only top-down level of the partitioned search: every edge into a vertex this rank has not forwarded yet marks the
// destination's byte -- a plain store, no claim, no id queue, nothing to bucket.  The bytes become per-owner bitmaps, ONE
// all-to-all moves slice q to rank q (n / 8 bytes per rank in total, what a bottom-up level's all-gather moves), and the owner
// ORs what it received into its visited bitmap, labelling the new bits in vertex order.  The search stays bottom-up afterwards
// (as it does after its first sweep), so the "already forwarded" bitmap is not needed again and is not updated.
struct MarkFunctor {
    typedef PbfsProblem::DataSlice DataSlice;
    static __device__ __forceinline__ bool ScreenEdge(int, int d, DataSlice *p, int = 0, int = 0)
    {
        return ((p->d_sent_mask[static_cast<unsigned>(d) >> 5] >> (d & 31)) & 1u) == 0;
    }
    static __device__ __forceinline__ bool CondEdge(int, int d, DataSlice *p, int = 0, int = 0)
    {
        const unsigned local = static_cast<unsigned>((static_cast<unsigned long long>(static_cast<unsigned>(d)) * p->mark_magic) >> p->mark_shift);
        const unsigned owner = static_cast<unsigned>(d) - local * static_cast<unsigned>(p->parts);
        p->d_mark[static_cast<size_t>(owner) * p->mark_stride + local] = 1;
        return true;
    }
    static __device__ __forceinline__ void ApplyEdge(int, int, DataSlice *, int = 0, int = 0) {}
};

// mark bytes -> bits (one thread per 32-bit word of the owner-major bitmap), bytes cleared for the next such level
__global__ void MarkBytesToBitsKernel(unsigned char *d_mark, long long words, unsigned *d_bits)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long w = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; w < words; w += stride) {
        uint4 *src = reinterpret_cast<uint4 *>(d_mark) + 2 * w;
        const uint4 a = src[0], b = src[1];
        // bytes are 0 or 1: (x * 0x01020408) >> 24 gathers the low bits of a dword's four bytes
        const unsigned lo = ((a.x * 0x01020408u) >> 24 & 0xFu) | ((a.y * 0x01020408u) >> 24 & 0xFu) << 4 |
                            ((a.z * 0x01020408u) >> 24 & 0xFu) << 8 | ((a.w * 0x01020408u) >> 24 & 0xFu) << 12;
        const unsigned hi = ((b.x * 0x01020408u) >> 24 & 0xFu) | ((b.y * 0x01020408u) >> 24 & 0xFu) << 4 |
                            ((b.z * 0x01020408u) >> 24 & 0xFu) << 8 | ((b.w * 0x01020408u) >> 24 & 0xFu) << 12;
        if (lo) src[0] = make_uint4(0, 0, 0, 0);
        if (hi) src[1] = make_uint4(0, 0, 0, 0);
        d_bits[w] = lo | (hi << 16);
    }
}

// owner side: OR of the slices that arrived (slice p at d_recv + p * words), new = that & ~visited; visited, labels (vertex
// order), the level's frontier bitmap and the count (wide tail)
__global__ void OrSlicesKernel(const unsigned *d_recv, int parts, long long slice_words, long long words, long long n_local, unsigned *d_visited,
                               int *d_labels, int label, unsigned long long *d_wide)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    unsigned count = 0;
    for (long long w = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; w < words; w += stride) {
        unsigned cand = 0;
        for (int p = 0; p < parts; ++p) cand |= d_recv[static_cast<size_t>(p) * slice_words + w];
        const long long first = w * 32;
        if (first >= n_local) cand = 0;
        else if (first + 32 > n_local) cand &= (1u << (n_local - first)) - 1u;
        unsigned fresh = cand & ~d_visited[w];
        if (fresh) {
            d_visited[w] |= fresh;  // (one thread per word: no atomics)
            count += static_cast<unsigned>(__popc(fresh));
            while (fresh) {
                const int b = __ffs(fresh) - 1;
                fresh &= fresh - 1;
                d_labels[first + b] = label;
            }
        }
    }
    const unsigned long long total = util::WaveSum(static_cast<unsigned long long>(count));
    if (util::LaneId() == 0 && total) atomicAdd(util::WideTailSlot(d_wide), total);
}


// advance functor: forward every destination at most once per rank
struct SendFunctor {
    typedef PbfsProblem::DataSlice DataSlice;
    static __device__ __forceinline__ bool ScreenEdge(int, int d, DataSlice *p, int = 0, int = 0)
    {
        return ((p->d_sent_mask[static_cast<unsigned>(d) >> 5] >> (d & 31)) & 1u) == 0;
    }
    static __device__ __forceinline__ bool CondEdge(int, int d, DataSlice *p, int = 0, int = 0)
    {
        const unsigned bit = 1u << (d & 31);
        return (atomicOr(p->d_sent_mask + (static_cast<unsigned>(d) >> 5), bit) & bit) == 0;
    }
    // (the claim in two halves: a tile's atomics overlap, oprtr/advance/functor_hooks.hpp)
    static __device__ __forceinline__ unsigned IssueEdge(int, int d, DataSlice *p, int = 0, int = 0)
    {
        return atomicOr(p->d_sent_mask + (static_cast<unsigned>(d) >> 5), 1u << (d & 31));
    }
    static __device__ __forceinline__ bool ResolveEdge(unsigned token, int, int d, DataSlice *, int = 0, int = 0) { return (token & (1u << (d & 31))) == 0; }
    static __device__ __forceinline__ void ApplyEdge(int s, int d, DataSlice *p, int = 0, int = 0)
    {
        if (p->d_pred_global) p->d_pred_global[d] = s * p->parts + p->rank;  // (s is a LOCAL row: its global id)
    }
};

// filter functor for received LOCAL ids: first arrival claims the vertex and labels it
struct ReceiveFunctor {
    typedef PbfsProblem::DataSlice DataSlice;
    static __device__ __forceinline__ bool CondFilter(int node, DataSlice *p, int = 0, int = 0)
    {
        unsigned *word = p->d_visited_mask + (static_cast<unsigned>(node) >> 5);
        const unsigned bit = 1u << (node & 31);
        if (*word & bit) return false;
        return (atomicOr(word, bit) & bit) == 0;
    }
    static __device__ __forceinline__ bool ScreenFilter(int node, DataSlice *p, int = 0, int = 0)
    {
        return ((p->d_visited_mask[static_cast<unsigned>(node) >> 5] >> (node & 31)) & 1u) == 0;
    }
    static __device__ __forceinline__ unsigned IssueFilter(int node, DataSlice *p, int = 0, int = 0)
    {
        return atomicOr(p->d_visited_mask + (static_cast<unsigned>(node) >> 5), 1u << (node & 31));
    }
    static __device__ __forceinline__ bool ResolveFilter(unsigned token, int node, DataSlice *, int = 0, int = 0) { return (token & (1u << (node & 31))) == 0; }
    static __device__ __forceinline__ void ApplyFilter(int node, DataSlice *p, int = 0, int nid = 0)
    {
        p->d_labels[node] = p->iteration + 1;
        if (p->d_recv_preds) p->d_preds[node] = p->d_recv_preds[nid];  // the pair that won the claim
    }
};

// histogram of owners (v mod parts) -- LDS counters, then `parts` global atomics per workgroup
__global__ void CountOwnersKernel(const int *d_ids, int n, int parts, unsigned *d_counts)
{
    __shared__ unsigned s_count[64];
    if (threadIdx.x < 64) s_count[threadIdx.x] = 0;
    __syncthreads();
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        atomicAdd(&s_count[static_cast<unsigned>(d_ids[i]) % static_cast<unsigned>(parts)], 1u);
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < parts && s_count[threadIdx.x]) atomicAdd(d_counts + threadIdx.x, s_count[threadIdx.x]);
}

// scatter ids into per-owner contiguous segments as LOCAL ids (v div parts); d_cursor[o] starts at segment o's offset.
// One wave-aggregated atomic per (wave, owner present in the wave).
__global__ void ScatterByOwnerKernel(const int *d_ids, int n, int parts, unsigned *d_cursor, int *d_out)
{
    const int stride = gridDim.x * blockDim.x;
    const int rounds = (n + stride - 1) / stride;
    for (int r = 0; r < rounds; ++r) {
        const int i = r * stride + blockIdx.x * blockDim.x + threadIdx.x;
        const bool live = i < n;
        const unsigned v = live ? static_cast<unsigned>(d_ids[i]) : 0u;
        const unsigned owner = v % static_cast<unsigned>(parts);
        unsigned long long todo = __ballot(live);
        while (todo) {  // wave-uniform loop over the distinct owners present
            const int leader = __ffsll(static_cast<long long>(todo)) - 1;
            const unsigned o = __shfl(owner, leader, util::kWaveSize);
            const unsigned long long same = __ballot(live && owner == o);
            unsigned base = 0;
            if (static_cast<int>(util::LaneId()) == leader) base = atomicAdd(d_cursor + o, static_cast<unsigned>(__popcll(same)));
            base = __shfl(base, leader, util::kWaveSize);
            if (live && owner == o) d_out[base + util::RankInMask(same)] = static_cast<int>(v / static_cast<unsigned>(parts));
            todo &= ~same;
        }
    }
}

// ---- the same two bucketing steps for the in-library loop: the candidate count stays on the device (low word of a packed
//      tail), the scatter computes the segment offsets itself, parents travel next to the ids ----
// layout of Pbfs::d_small (words): [0] error word, [1, 65) counts to send, [65, 129) cursors, [192, ...) gathered matrices
constexpr int kSmallCounts = 1, kSmallCursors = 65, kSmallGathered = 192, kSmallMaxParts = 64;
constexpr int kSmallWords = kSmallGathered + (kSmallMaxParts + 1) * kSmallMaxParts;   // device words
constexpr int kMailWords = (kSmallMaxParts + 1) * kSmallMaxParts;                     // pinned mailbox words (+ sequence word behind)

__global__ void CountOwnersDeviceKernel(const int *d_ids, const unsigned long long *d_n, int parts, unsigned *d_counts)
{
    __shared__ unsigned s_count[64];
    if (threadIdx.x < 64) s_count[threadIdx.x] = 0;
    __syncthreads();
    const int n = static_cast<int>(util::TailCount(*d_n));
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        atomicAdd(&s_count[static_cast<unsigned>(d_ids[i]) % static_cast<unsigned>(parts)], 1u);
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < parts && s_count[threadIdx.x]) atomicAdd(d_counts + threadIdx.x, s_count[threadIdx.x]);
}

// d_counts[0..parts) = per-owner totals (complete), d_cursor[0..parts) = 0 on entry.
// Two sweeps over the workgroup's share: count per owner in LDS, reserve each owner's run with ONE global atomic per
// workgroup (a cursor is a hot address: ~88 atomics per microsecond -- one per wave cost 238 us for 1.3 M ids), then place.
__global__ void ScatterByOwnerDeviceKernel(const int *d_ids, const unsigned long long *d_n, int parts, const unsigned *d_counts,
                                           unsigned *d_cursor, int *d_out, const int *d_pred_global, int *d_out_preds)
{
    __shared__ unsigned s_count[64], s_base[64];
    if (threadIdx.x < 64) s_count[threadIdx.x] = 0;
    __syncthreads();
    const int n = static_cast<int>(util::TailCount(*d_n));
    const int per_block = (n + gridDim.x - 1) / gridDim.x;
    const int begin = blockIdx.x * per_block;
    const int end = begin + per_block < n ? begin + per_block : n;
    for (int i = begin + threadIdx.x; i < end; i += blockDim.x)
        atomicAdd(&s_count[static_cast<unsigned>(d_ids[i]) % static_cast<unsigned>(parts)], 1u);
    __syncthreads();
    if (static_cast<int>(threadIdx.x) < parts) {
        unsigned off = 0;  // start of this owner's segment = sum of the totals before it
        for (int o = 0; o < static_cast<int>(threadIdx.x); ++o) off += d_counts[o];
        const unsigned mine = s_count[threadIdx.x];
        s_base[threadIdx.x] = off + (mine ? atomicAdd(d_cursor + threadIdx.x, mine) : 0u);
        s_count[threadIdx.x] = 0;
    }
    __syncthreads();
    for (int i = begin + threadIdx.x; i < end; i += blockDim.x) {
        const unsigned v = static_cast<unsigned>(d_ids[i]);
        const unsigned o = v % static_cast<unsigned>(parts);
        const unsigned at = s_base[o] + atomicAdd(&s_count[o], 1u);
        d_out[at] = static_cast<int>(v / static_cast<unsigned>(parts));
        if (d_out_preds) d_out_preds[at] = d_pred_global[v];
    }
}

// a few device words -> pinned host memory + a sequence word (system-scope release): the host spins on the word instead of
// paying a copy-engine transfer and a stream synchronisation (~25 us each) for every number it needs
__global__ void BitmapOrKernel(unsigned long long *d_into, const unsigned long long *d_from, long long words64)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long w = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; w < words64; w += stride) d_into[w] |= d_from[w];
}

__global__ void MailKernel(const unsigned *d_src, int count, int stride, unsigned *h_box, unsigned long long *h_seq, unsigned long long seq)
{
    for (int i = threadIdx.x; i < count; i += blockDim.x) h_box[i] = d_src[static_cast<size_t>(i) * stride];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Reset of a search in ONE launch (Pbfs::Search): labels, parents, the visited bitmap and its "before the level" snapshot
// (edgeless vertices pre-marked when d_never is given), the sender's global bitmap, the enactor's tail ring / overflow flag /
// wide counters, and the source patched in: label 0 and queue entry on its owner, "already forwarded" everywhere.
__global__ void PbfsResetKernel(int src, int parts, int rank, const int *d_row_offsets, int *d_labels, int *d_preds, long long n_local,
                                unsigned *d_visited, unsigned *d_before, const unsigned *d_never, long long local_words, unsigned *d_sent,
                                long long sent_words, util::Frontier<int, int> queue0, unsigned long long *d_tail, int tail_slots,
                                int *d_overflow, unsigned long long *d_wide, int wide_words)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    const long long tid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const bool mine = src % parts == rank;
    const int local = src / parts;
    for (long long i = tid; i < n_local; i += stride) {
        const bool is_src = mine && i == local;
        d_labels[i] = is_src ? 0 : -1;
        d_preds[i] = is_src ? -1 : -2;
    }
    for (long long w = tid; w < local_words; w += stride) {
        const unsigned never = d_never ? d_never[w] : 0u;
        d_before[w] = never;
        d_visited[w] = (mine && w == (local >> 5)) ? (never | (1u << (local & 31))) : never;
    }
    for (long long w = tid; w < sent_words; w += stride) d_sent[w] = (w == (src >> 5)) ? (1u << (src & 31)) : 0u;
    if (tid < wide_words) d_wide[tid] = 0ull;
    if (tid == 0) {
        *d_overflow = 0;
        unsigned long long tail = 0ull;
        if (mine) {
            const int begin = d_row_offsets[local], end = d_row_offsets[local + 1];
            queue0.v[0] = local;
            queue0.row_start[0] = begin;
            queue0.scan[0] = 0;
            if (end > begin) tail = util::PackTail(1u, static_cast<unsigned>(end - begin));
        }
        for (int s = 0; s < tail_slots; ++s) d_tail[s] = s == 0 ? tail : 0ull;
    }
}

// start of a top-down level: the two tail slots and the owner histogram (counts, cursors) back to zero -- one launch
// d_small[0] = this rank's error word for the level's count all-gather (the queue-overflow flag as earlier levels left it):
// every rank sees every rank's word in the gathered matrix, so all of them leave the level loop at the same level
__global__ void PbfsArmLevelKernel(unsigned long long *d_tail, unsigned *d_small, const int *d_overflow)
{
    if (threadIdx.x < 2) d_tail[threadIdx.x] = 0ull;
    if (threadIdx.x < kSmallGathered) d_small[threadIdx.x] = (threadIdx.x == 0 && *d_overflow != 0) ? 1u : 0u;
}

// sum of the wide tail lines (the bottom-up sweep's per-workgroup find counts) -> one word, lines cleared
__global__ void FoldWideKernel(unsigned long long *d_wide, unsigned *d_out, unsigned long long *d_tail_slot)
{
    const unsigned lane = threadIdx.x;
    unsigned long long w = 0;
    if (lane < 32) {
        w = d_wide[lane * 16];
        if (w) d_wide[lane * 16] = 0;
    }
    for (int o = 16; o; o >>= 1) w += __shfl_xor(w, o, 64);
    if (lane == 0) {
        if (d_tail_slot) {
            w += *d_tail_slot;
            *d_tail_slot = 0ull;
        }
        *d_out = util::TailCount(w);
    }
}

// ---- the exchange: what a level needs from the other ranks ----
struct Transport {
    virtual ~Transport() {}
    // every rank contributes `words` 32-bit words; d_recv receives parts * words (rank order).  Enqueued on `stream` or
    // complete on return; the caller synchronises the stream before it reads the result on the host.
    virtual int AllGather(const unsigned *d_send, unsigned *d_recv, size_t words, hipStream_t stream) = 0;
    // 32-bit words: segment p of d_send goes to rank p, segment p of d_recv comes from rank p
    virtual int AllToAllV(const int *d_send, const size_t *send_counts, const size_t *send_offsets, int *d_recv,
                          const size_t *recv_counts, const size_t *recv_offsets, hipStream_t stream) = 0;
    virtual const char *Name() const = 0;
};

// RCCL through dlopen: the library does not link against librccl; torch's copy is reused when it is already loaded
struct RcclApi {
    typedef int (*GetUniqueId_t)(void *);
    typedef int (*CommInitRank_t)(void **, int, ncclUniqueIdBytes, int);
    void *handle = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **comm, int nranks, ncclUniqueIdBytes id, int rank) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;

    static RcclApi &Get()
    {
        static RcclApi api;
        return api;
    }
    bool Load()
    {
        if (handle) return true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names)
            if ((handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
        if (!handle) {
            std::fprintf(stderr, "gunrock: cannot load RCCL (%s)\n", dlerror());
            return false;
        }
        bool ok = true;
        auto sym = [&](const char *nm) {
            void *p = dlsym(handle, nm);
            if (!p) {
                std::fprintf(stderr, "gunrock: RCCL symbol %s missing\n", nm);
                ok = false;
            }
            return p;
        };
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) handle = nullptr;
        return ok;
    }
};

struct RcclTransport : Transport {
    void *comm = nullptr;
    int parts = 1, rank = 0;
    static constexpr int kInt32 = 2;  // ncclInt32 (rccl.h ncclDataType_t)
    int Check(int rc, const char *what)
    {
        if (rc != 0) std::fprintf(stderr, "gunrock: RCCL %s failed: %s\n", what, RcclApi::Get().GetErrorString ? RcclApi::Get().GetErrorString(rc) : "?");
        return rc;
    }
    int Init(int parts_, int rank_, const char id[128])
    {
        RcclApi &api = RcclApi::Get();
        if (!api.Load()) return -1;
        parts = parts_;
        rank = rank_;
        ncclUniqueIdBytes uid;
        std::memcpy(uid.internal, id, 128);
        return Check(api.CommInitRank(&comm, parts, uid, rank), "ncclCommInitRank");
    }
    ~RcclTransport() override
    {
        if (comm) RcclApi::Get().CommDestroy(comm);
    }
    int AllGather(const unsigned *d_send, unsigned *d_recv, size_t words, hipStream_t stream) override
    {
        return Check(RcclApi::Get().AllGather(d_send, d_recv, words, kInt32, comm, stream), "ncclAllGather");
    }
    int AllToAllV(const int *d_send, const size_t *sc, const size_t *so, int *d_recv, const size_t *rc, const size_t *ro,
                  hipStream_t stream) override
    {
        RcclApi &api = RcclApi::Get();
        int err = api.GroupStart();
        for (int p = 0; p < parts && !err; ++p) {  // all peers at once: xGMI is point-to-point, every link carries its own pair
            if (sc[p]) err = api.Send(d_send + so[p], sc[p], kInt32, p, comm, stream);
            if (!err && rc[p]) err = api.Recv(d_recv + ro[p], rc[p], kInt32, p, comm, stream);
        }
        const int end = api.GroupEnd();
        return Check(err ? err : end, "ncclSend/ncclRecv group");
    }
    const char *Name() const override { return "rccl"; }
};

// the caller performs the exchange (synchronously); used by tests that run several ranks on one GPU over gloo
struct CallbackTransport : Transport {
    void *ctx = nullptr;
    grx_all_gather_fn gather = nullptr;
    grx_all_to_all_v_fn a2a = nullptr;
    int AllGather(const unsigned *d_send, unsigned *d_recv, size_t words, hipStream_t stream) override
    {
        if (hipStreamSynchronize(stream) != hipSuccess) return -1;
        return gather(ctx, d_send, d_recv, words);
    }
    int AllToAllV(const int *d_send, const size_t *sc, const size_t *so, int *d_recv, const size_t *rc, const size_t *ro,
                  hipStream_t stream) override
    {
        if (hipStreamSynchronize(stream) != hipSuccess) return -1;
        return a2a(ctx, d_send, sc, so, d_recv, rc, ro);
    }
    const char *Name() const override { return "callbacks"; }
};

struct Pbfs : app::EnactorBase {
    int parts = 1, rank = 0;
    int n_global = 0, n_local = 0, n_local_max = 0, m_local = 0;
    int *d_row_offsets = nullptr, *d_col_indices = nullptr;  // borrowed
    PbfsProblem::DataSlice ds{};
    unsigned *d_frontier_mask[2] = {nullptr, nullptr};
    int2 *d_heads = nullptr;
    // bottom-up statics (the rows double as in-lists: the graph is symmetric whenever a search goes bottom-up):
    // bit v of d_never = local vertex v has no edge -- preloaded into the visited bitmap of a direction-optimizing search so
    // the sweeps skip such vertices 64 at a time; heads only exist for the others (d_head_base, bottom_up.hpp)
    unsigned *d_never = nullptr;
    unsigned *d_head_base = nullptr;
    long long with_in_edges = 0;
    int sparse_sweep_div = 6;            // compacting sweep when at most n_local / 6 local vertices can still be unvisited (64-word chunks)
    bool never_applied = false;          // step-wise path: visited |= never happened since the last Reset
    util::Frontier<int, int> queues[2];
    int *d_candidates = nullptr, *d_send = nullptr;
    unsigned *d_counts = nullptr, *h_counts = nullptr;  // 2 * 64: counts, cursors
    int candidate_capacity = 0;
    hipStream_t stream = 0;
    int selector = 0, cur_mask = 0;
    unsigned frontier_len = 0, frontier_edges = 0;
    int level = 0;
    // ---- in-library level loop (Search) ----
    Transport *transport = nullptr;
    bool mark_pred = false;
    // top-down -> bottom-up when global frontier edges * alpha > unexplored edges.  30, not the single-GPU enactor's 10: a
    // top-down level here pays its claims twice (sender's "sent" bitmap, owner's visited bitmap) plus the bucketing and the
    // exchange, so the sweep wins earlier (scale-24, 65 sources, one rank: mean 0.845 ms at 10, 0.787 at 30, 0.814 at 60)
    // Round 3: with count-only levels in front of the sweeps the optimum moved back to the single-GPU value (forced one-rank run,
    // lite_factor 230: 0.71 ms at 30, 0.60 at 10..14, 0.65 at 6).
    double alpha = 12.0;
    long long m_global = -1;             // sum of the ranks' edge counts (learned through the transport)
    int *d_recv = nullptr;               // received local ids, then (mark_pred) received parents behind them
    int recv_capacity = 0;
    int *d_send_preds = nullptr;
    unsigned *d_small = nullptr;         // kSmall* layout: error word, counts to send, cursors, gathered matrices
    unsigned *h_small = nullptr;         // pinned + mapped mirror of the gathered part; word [kMailWords] onwards: the mail sequence
    unsigned long long mail_seq = 0;
    unsigned *d_mark_bits = nullptr;     // owner-major bitmap of a marked level: parts x mark words
    // a top-down level runs count-only ("marked") when global frontier edges * alpha * lite_factor > unexplored edges (and no
    // parents are wanted: a marked vertex does not say who marked it); 0 = never.  The single-GPU enactor's rule with the
    // same product alpha * lite_factor (bfs_problem.hpp).
    double lite_factor = 230.0;
    long long marked_levels = 0;         // statistics: count-only levels run so far
    unsigned *d_visited_before = nullptr;  // local visited bitmap as it was before the last top-down level
    unsigned *d_gathered = nullptr;      // parts x (bitmap words + 2)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    std::vector<size_t> sc, so, rc, ro;

    Pbfs() : app::EnactorBase(app::VERTEX_FRONTIERS, false) {}

    int MaskWords(int n) const { return ((n + 63) / 64) * 2; }

    hipError_t Init(int n_global_, int parts_, int rank_, int n_local_, int m_local_, int *d_ro, int *d_ci)
    {
        hipError_t retval = hipSuccess;
        parts = parts_; rank = rank_; n_global = n_global_; n_local = n_local_; m_local = m_local_;
        d_row_offsets = d_ro; d_col_indices = d_ci;
        n_local_max = (n_global + parts - 1) / parts;  // same bitmap length on every rank (all-gather needs it)
        if ((retval = EnactorBase::Setup(0, 3, 8))) return retval;
        GR_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking), "Pbfs hipStreamCreate failed");
        const size_t nl = static_cast<size_t>(n_local > 0 ? n_local : 1);
        GR_CHECK(hipMalloc(&ds.d_labels, sizeof(int) * nl), "Pbfs hipMalloc failed");
        GR_CHECK(hipMalloc(&ds.d_visited_mask, sizeof(unsigned) * (MaskWords(n_local) + 2)), "Pbfs hipMalloc failed");
        GR_CHECK(hipMalloc(&ds.d_sent_mask, sizeof(unsigned) * (MaskWords(n_global) + 2)), "Pbfs hipMalloc failed");
        GR_CHECK(hipMalloc(&ds.d_preds, sizeof(int) * nl), "Pbfs hipMalloc failed");
        ds.d_pred_global = nullptr;
        ds.d_recv_preds = nullptr;
        ds.parts = parts;
        ds.rank = rank;
        for (int i = 0; i < 2; ++i) {
            GR_CHECK(hipMalloc(&d_frontier_mask[i], sizeof(unsigned) * (MaskWords(n_local_max) + 2)), "Pbfs hipMalloc failed");
            const size_t cap = nl + 1024;
            GR_CHECK(hipMalloc(&queues[i].v, sizeof(int) * cap), "Pbfs hipMalloc failed");
            GR_CHECK(hipMalloc(&queues[i].row_start, sizeof(int) * cap), "Pbfs hipMalloc failed");
            GR_CHECK(hipMalloc(&queues[i].scan, sizeof(int) * cap), "Pbfs hipMalloc failed");
            queues[i].capacity = static_cast<int>(cap);
        }
        GR_CHECK(hipMalloc(&d_heads, sizeof(int2) * nl), "Pbfs hipMalloc failed");
        {   // never mask, compact head index (as BFSProblem::SetInverseGraph, app/bfs/bfs_problem.hpp)
            const long long words64 = static_cast<long long>(MaskWords(n_local)) / 2 + 1;
            GR_CHECK(hipMalloc(&d_never, sizeof(unsigned) * (MaskWords(n_local) + 2)), "Pbfs hipMalloc failed");
            GR_CHECK(hipMalloc(&d_head_base, sizeof(unsigned) * static_cast<size_t>(words64 + 1)), "Pbfs hipMalloc failed");
            long long grid = (words64 + 3) / 4;
            if (grid > 2048) grid = 2048;
            hipLaunchKernelGGL((app::bfs::NoInEdgeMaskKernel<int>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, stream, d_row_offsets,
                               static_cast<long long>(n_local), words64, reinterpret_cast<unsigned long long *>(d_never));
            GR_CHECK(hipGetLastError(), "NoInEdgeMaskKernel launch failed");
            unsigned *d_word_counts = nullptr;
            unsigned long long *d_scan_sums = nullptr;
            GR_CHECK(hipMalloc(&d_word_counts, sizeof(unsigned) * static_cast<size_t>(words64 + 1)), "Pbfs hipMalloc failed");
            GR_CHECK(hipMalloc(&d_scan_sums, sizeof(unsigned long long) * static_cast<size_t>(graphio::ScanScratchWords(words64))), "Pbfs hipMalloc failed");
            hipLaunchKernelGGL(app::bfs::WithInEdgesCountKernel, dim3(static_cast<unsigned>((words64 + 255) / 256)), dim3(256), 0, stream,
                               reinterpret_cast<const unsigned long long *>(d_never), static_cast<long long>(n_local), words64, d_word_counts);
            GR_CHECK(hipGetLastError(), "WithInEdgesCountKernel launch failed");
            GR_CHECK(graphio::DeviceExclusiveScan<unsigned>(d_word_counts, d_head_base, words64, d_scan_sums, stream), "Pbfs head-base scan failed");
            unsigned last_base = 0, last_count = 0;
            GR_CHECK(hipMemcpyAsync(&last_base, d_head_base + (words64 - 1), sizeof(unsigned), hipMemcpyDeviceToHost, stream), "Pbfs read failed");
            GR_CHECK(hipMemcpyAsync(&last_count, d_word_counts + (words64 - 1), sizeof(unsigned), hipMemcpyDeviceToHost, stream), "Pbfs read failed");
            GR_CHECK(hipStreamSynchronize(stream), "Pbfs sync failed");
            with_in_edges = static_cast<long long>(last_base) + last_count;
            GR_CHECK(hipFree(d_word_counts), "Pbfs hipFree failed");
            GR_CHECK(hipFree(d_scan_sums), "Pbfs hipFree failed");
        }
        if (n_local > 0) {
            // Heads = the two neighbours of largest degree among the first 512 of the row (bottom_up.hpp).  The columns are GLOBAL
            // ids: with one part they index the rows themselves; otherwise a vertex's degree is estimated by how often it occurs
            // in this rank's columns (its edges into the 1/P sample of the graph that lives here) -- histogram + scan, once.
            int *d_degree_offsets = nullptr;
            if (parts > 1) {
                const long long words = static_cast<long long>(n_global) + 1;
                unsigned *d_hist = nullptr;
                unsigned long long *d_sums = nullptr;
                GR_CHECK(hipMalloc(&d_hist, sizeof(unsigned) * static_cast<size_t>(words)), "Pbfs hipMalloc failed");
                GR_CHECK(hipMalloc(&d_degree_offsets, sizeof(int) * static_cast<size_t>(words)), "Pbfs hipMalloc failed");
                GR_CHECK(hipMalloc(&d_sums, sizeof(unsigned long long) * static_cast<size_t>(graphio::ScanScratchWords(words))), "Pbfs hipMalloc failed");
                GR_CHECK(hipMemsetAsync(d_hist, 0, sizeof(unsigned) * static_cast<size_t>(words), stream), "Pbfs memset failed");
                if (m_local > 0) {
                    hipLaunchKernelGGL(graphio::InDegreeKernel, dim3(2048), dim3(256), 0, stream, d_col_indices, static_cast<long long>(m_local), d_hist);
                    GR_CHECK(hipGetLastError(), "InDegreeKernel launch failed");
                }
                GR_CHECK(graphio::DeviceExclusiveScan<int>(d_hist, d_degree_offsets, words, d_sums, stream), "Pbfs degree scan failed");
                GR_CHECK(hipStreamSynchronize(stream), "Pbfs sync failed");
                GR_CHECK(hipFree(d_hist), "Pbfs hipFree failed");
                GR_CHECK(hipFree(d_sums), "Pbfs hipFree failed");
            }
            hipLaunchKernelGGL((oprtr::advance::BuildHeadsKernel<int, int>), dim3((n_local + 3) / 4 < 8192 ? (n_local + 3) / 4 : 8192),
                               dim3(256), 0, stream, d_row_offsets, d_col_indices, static_cast<long long>(n_local), d_heads,
                               static_cast<const int *>(parts > 1 ? d_degree_offsets : d_row_offsets),
                               reinterpret_cast<const unsigned long long *>(d_never), d_head_base);
            GR_CHECK(hipGetLastError(), "BuildHeadsKernel launch failed");
            GR_CHECK(hipStreamSynchronize(stream), "Pbfs sync failed");
            if (d_degree_offsets) GR_CHECK(hipFree(d_degree_offsets), "Pbfs hipFree failed");
        }
        // a rank forwards each global vertex at most once over the whole search
        candidate_capacity = n_global + 1024;
        GR_CHECK(hipMalloc(&d_candidates, sizeof(int) * static_cast<size_t>(candidate_capacity)), "Pbfs hipMalloc failed");
        GR_CHECK(hipMalloc(&d_send, sizeof(int) * static_cast<size_t>(candidate_capacity)), "Pbfs hipMalloc failed");
        GR_CHECK(hipMalloc(&d_counts, sizeof(unsigned) * 128), "Pbfs hipMalloc failed");
        GR_CHECK(hipHostMalloc(&h_counts, sizeof(unsigned) * 128, hipHostMallocDefault), "Pbfs hipHostMalloc failed");
        return retval;
    }

    ~Pbfs() override
    {
        delete transport;
        if (ds.d_labels) hipFree(ds.d_labels);
        if (ds.d_preds) hipFree(ds.d_preds);
        if (ds.d_pred_global) hipFree(ds.d_pred_global);
        if (ds.d_mark) hipFree(ds.d_mark);
        if (d_mark_bits) hipFree(d_mark_bits);
        if (d_recv) hipFree(d_recv);
        if (d_send_preds) hipFree(d_send_preds);
        if (d_small) hipFree(d_small);
        if (h_small) hipHostFree(h_small);
        if (d_gathered) hipFree(d_gathered);
        if (d_visited_before) hipFree(d_visited_before);
        if (ev_start) hipEventDestroy(ev_start);
        if (ev_stop) hipEventDestroy(ev_stop);
        if (ds.d_visited_mask) hipFree(ds.d_visited_mask);
        if (ds.d_sent_mask) hipFree(ds.d_sent_mask);
        for (int i = 0; i < 2; ++i) {
            if (d_frontier_mask[i]) hipFree(d_frontier_mask[i]);
            if (queues[i].v) hipFree(queues[i].v);
            if (queues[i].row_start) hipFree(queues[i].row_start);
            if (queues[i].scan) hipFree(queues[i].scan);
        }
        if (d_heads) hipFree(d_heads);
        if (d_never) hipFree(d_never);
        if (d_head_base) hipFree(d_head_base);
        if (d_candidates) hipFree(d_candidates);
        if (d_send) hipFree(d_send);
        if (d_counts) hipFree(d_counts);
        if (h_counts) hipHostFree(h_counts);
        if (stream) hipStreamDestroy(stream);
    }

    hipError_t Reset(int src)
    {
        hipError_t retval = hipSuccess;
        util::Memset(ds.d_labels, -1, n_local, stream);
        util::Memset(ds.d_preds, -2, n_local, stream);
        util::Memset(ds.d_visited_mask, 0u, MaskWords(n_local) + 2, stream);
        util::Memset(ds.d_sent_mask, 0u, MaskWords(n_global) + 2, stream);
        for (int i = 0; i < 2; ++i) util::Memset(d_frontier_mask[i], 0u, MaskWords(n_local_max) + 2, stream);
        if ((retval = work_progress.Reset(stream))) return retval;
        selector = 0; cur_mask = 0; level = 0; frontier_len = 0; frontier_edges = 0;
        never_applied = false;
        // every rank marks the source as already forwarded
        if (src >= 0 && src < n_global) {
            const unsigned bit = 1u << (src & 31);
            GR_CHECK(hipMemcpyAsync(ds.d_sent_mask + (src >> 5), &bit, sizeof(unsigned), hipMemcpyHostToDevice, stream), "Pbfs seed failed");
            if (src % parts == rank) {
                const int local = src / parts;
                int row[2];
                GR_CHECK(hipMemcpyAsync(row, d_row_offsets + local, sizeof(int) * 2, hipMemcpyDeviceToHost, stream), "Pbfs seed failed");
                GR_CHECK(hipStreamSynchronize(stream), "Pbfs seed failed");
                const int zero = 0;
                const unsigned lbit = 1u << (local & 31);
                const int minus_one = -1;
                GR_CHECK(hipMemcpyAsync(ds.d_labels + local, &zero, sizeof(int), hipMemcpyHostToDevice, stream), "Pbfs seed failed");
                GR_CHECK(hipMemcpyAsync(ds.d_preds + local, &minus_one, sizeof(int), hipMemcpyHostToDevice, stream), "Pbfs seed failed");
                GR_CHECK(hipMemcpyAsync(ds.d_visited_mask + (local >> 5), &lbit, sizeof(unsigned), hipMemcpyHostToDevice, stream), "Pbfs seed failed");
                GR_CHECK(hipMemcpyAsync(queues[0].v, &local, sizeof(int), hipMemcpyHostToDevice, stream), "Pbfs seed failed");
                GR_CHECK(hipMemcpyAsync(queues[0].row_start, &row[0], sizeof(int), hipMemcpyHostToDevice, stream), "Pbfs seed failed");
                GR_CHECK(hipMemcpyAsync(queues[0].scan, &zero, sizeof(int), hipMemcpyHostToDevice, stream), "Pbfs seed failed");
                GR_CHECK(hipStreamSynchronize(stream), "Pbfs seed failed");
                if (row[1] - row[0] > 0) { frontier_len = 1; frontier_edges = static_cast<unsigned>(row[1] - row[0]); }
            }
        }
        GR_CHECK(hipStreamSynchronize(stream), "Pbfs Reset sync failed");
        return retval;
    }

    // top-down step 1: expand the local frontier, bucket the newly forwarded ids by owner.
    // h_send_counts[parts] receives the per-destination counts; the bucketed LOCAL ids are in d_send (segments in rank order).
    hipError_t AdvanceLocal(unsigned *h_send_counts)
    {
        hipError_t retval = hipSuccess;
        for (int i = 0; i < parts; ++i) h_send_counts[i] = 0;
        GR_CHECK(hipMemsetAsync(work_progress.d_tail, 0, sizeof(unsigned long long) * 2, stream), "Pbfs clear tail failed");
        unsigned candidates = 0, unused = 0;
        if (frontier_len > 0) {
            oprtr::advance::AdvanceArgs<int, int> args;
            args.in = queues[selector];
            args.out = util::Frontier<int, int>();
            args.out.v = d_candidates;
            args.out.capacity = candidate_capacity;
            args.in_len = static_cast<int>(frontier_len);
            args.in_edges = static_cast<int>(frontier_edges);
            args.d_row_offsets = d_row_offsets;
            args.d_column_indices = d_col_indices;
            args.d_tail_out = work_progress.d_tail + 0;
            args.d_tail_clear = nullptr;
            args.d_overflow = work_progress.d_overflow;
            ds.iteration = level;
            typedef oprtr::advance::KernelPolicy<256, 8, 3, oprtr::advance::LB> Policy;
            if ((retval = oprtr::advance::LaunchKernel<Policy, PbfsProblem, SendFunctor, false>(args, ds, 0, stream)))
                return retval;
            if ((retval = work_progress.GetTail(0, candidates, unused, stream))) return retval;
        }
        if (candidates == 0) return retval;
        GR_CHECK(hipMemsetAsync(d_counts, 0, sizeof(unsigned) * 128, stream), "Pbfs clear counts failed");
        const int grid = (static_cast<int>(candidates) + 255) / 256 < cu_count * 4 ? (static_cast<int>(candidates) + 255) / 256 : cu_count * 4;
        hipLaunchKernelGGL(CountOwnersKernel, dim3(grid), dim3(256), 0, stream, d_candidates, static_cast<int>(candidates), parts, d_counts);
        GR_CHECK(hipGetLastError(), "CountOwnersKernel launch failed");
        GR_CHECK(hipMemcpyAsync(h_counts, d_counts, sizeof(unsigned) * 64, hipMemcpyDeviceToHost, stream), "Pbfs read counts failed");
        GR_CHECK(hipStreamSynchronize(stream), "Pbfs sync failed");
        unsigned offset = 0;
        for (int i = 0; i < parts; ++i) {
            h_send_counts[i] = h_counts[i];
            h_counts[64 + i] = offset;
            offset += h_counts[i];
        }
        GR_CHECK(hipMemcpyAsync(d_counts + 64, h_counts + 64, sizeof(unsigned) * 64, hipMemcpyHostToDevice, stream), "Pbfs write cursors failed");
        hipLaunchKernelGGL(ScatterByOwnerKernel, dim3(grid), dim3(256), 0, stream, d_candidates, static_cast<int>(candidates), parts,
                           d_counts + 64, d_send);
        GR_CHECK(hipGetLastError(), "ScatterByOwnerKernel launch failed");
        GR_CHECK(hipStreamSynchronize(stream), "Pbfs sync failed");
        return retval;
    }

    // top-down step 2: claim + label the received LOCAL ids, build the next local frontier
    hipError_t FilterReceived(const int *d_recv, int n_recv, unsigned *next_len, unsigned *next_edges)
    {
        hipError_t retval = hipSuccess;
        GR_CHECK(hipMemsetAsync(work_progress.d_tail + 1, 0, sizeof(unsigned long long), stream), "Pbfs clear tail failed");
        frontier_len = 0; frontier_edges = 0;
        if (n_recv > 0) {
            oprtr::filter::FilterArgs<int, int> f;
            f.d_in = d_recv;
            f.num_elements = n_recv;
            f.out = queues[selector ^ 1];
            f.d_tail_out = work_progress.d_tail + 1;
            f.d_tail_clear = nullptr;
            f.d_overflow = work_progress.d_overflow;
            f.d_row_offsets = d_row_offsets;
            ds.iteration = level;
            typedef oprtr::filter::KernelPolicy<256, 4, 8> Policy;
            if ((retval = oprtr::filter::LaunchKernel<Policy, PbfsProblem, ReceiveFunctor, true>(f, ds, cu_count * 8, stream)))
                return retval;
            if ((retval = work_progress.GetTail(1, frontier_len, frontier_edges, stream))) return retval;
        }
        selector ^= 1;
        ++level;
        *next_len = frontier_len;
        *next_edges = frontier_edges;
        return retval;
    }

    // queue -> local frontier bitmap (before the first bottom-up level)
    hipError_t QueueToBitmap()
    {
        hipError_t retval = hipSuccess;
        GR_CHECK(hipMemsetAsync(d_frontier_mask[cur_mask], 0, sizeof(unsigned) * (MaskWords(n_local_max) + 2), stream), "Pbfs memset failed");
        if (frontier_len > 0) {
            hipLaunchKernelGGL((oprtr::advance::QueueToBitmapKernel<int, int>), dim3(cu_count * 4), dim3(256), 0, stream,
                               queues[selector].v, static_cast<int>(frontier_len), d_frontier_mask[cur_mask]);
            GR_CHECK(hipGetLastError(), "QueueToBitmapKernel launch failed");
        }
        GR_CHECK(hipStreamSynchronize(stream), "Pbfs sync failed");
        return retval;
    }

    // bottom-up level against the all-gathered frontier bitmaps (parts x words_per_rank words)
    hipError_t BottomUp(const unsigned *d_gathered, int words_per_rank, unsigned *found, unsigned *found_edges)
    {
        hipError_t retval = hipSuccess;
        GR_CHECK(hipMemsetAsync(work_progress.d_tail + 1, 0, sizeof(unsigned long long), stream), "Pbfs clear tail failed");
        if (!never_applied) {  // (a search that goes bottom-up runs on a symmetric graph: an edgeless vertex is never discovered)
            hipLaunchKernelGGL(BitmapOrKernel, dim3(cu_count * 2), dim3(256), 0, stream, reinterpret_cast<unsigned long long *>(ds.d_visited_mask),
                               reinterpret_cast<const unsigned long long *>(d_never), static_cast<long long>(MaskWords(n_local) / 2));
            GR_CHECK(hipGetLastError(), "BitmapOrKernel launch failed");
            never_applied = true;
        }
        if ((retval = LaunchSweep(d_gathered, words_per_rank, -1))) return retval;
        if ((retval = work_progress.GetTailWide(1, frontier_len, frontier_edges, stream))) return retval;
        cur_mask ^= 1;
        ++level;
        *found = frontier_len;
        *found_edges = frontier_edges;
        return retval;
    }

    // One bottom-up sweep of the local vertices against the gathered frontier bitmaps: finds -> d_frontier_mask[cur_mask ^ 1],
    // visited, labels, parents; counts -> the wide tail.  open_estimate = local vertices with edges that can still be
    // unvisited (-1 unknown): few of them -> the compacting sweep (oprtr/advance/bottom_up.hpp, BottomUpSparseKernel).
    hipError_t LaunchSweep(const unsigned *d_gathered_masks, int words_per_rank, long long open_estimate)
    {
        hipError_t retval = hipSuccess;
        typedef oprtr::advance::StripedBitmapLookup<int> L;
        oprtr::advance::BottomUpArgs<int, int> b;
        b.nodes = n_local;
        b.d_inv_row_offsets = d_row_offsets;
        b.d_inv_column_indices = d_col_indices;
        b.d_inv_heads = d_heads;
        b.head_skip = 0;  // ranked heads: the row walk starts at the row's first entry
        b.d_never = reinterpret_cast<const unsigned long long *>(d_never);
        b.d_head_base = d_head_base;
        b.d_frontier_out = reinterpret_cast<unsigned long long *>(d_frontier_mask[cur_mask ^ 1]);
        b.d_visited = reinterpret_cast<unsigned long long *>(ds.d_visited_mask);
        b.d_tail_out = work_progress.d_tail + 1;
        b.d_tail_clear = nullptr;
        b.d_wide = work_progress.d_wide;  // per-workgroup counts spread over 32 lines, folded by the read-back
        ds.iteration = level;
        L lookup{d_gathered_masks, static_cast<unsigned>(parts), static_cast<unsigned>(words_per_rank)};
        const long long words64 = (static_cast<long long>(n_local) + 63) / 64;
        if (sparse_sweep_div > 0 && open_estimate >= 0 && open_estimate * sparse_sweep_div <= static_cast<long long>(n_local)) {
            long long sgrid = ((words64 + oprtr::advance::kSparseChunkWords - 1) / oprtr::advance::kSparseChunkWords + 3) / 4;
            const long long scap = util::ResidentGrid(oprtr::advance::BottomUpSparseKernel<256, 8, 32, PbfsProblem, L, false>, 256);
            if (sgrid > scap) sgrid = scap;
            if (sgrid < 1) sgrid = 1;
            hipLaunchKernelGGL((oprtr::advance::BottomUpSparseKernel<256, 8, 32, PbfsProblem, L, false>), dim3(static_cast<unsigned>(sgrid)),
                               dim3(256), 0, stream, b, ds, lookup);
            GR_CHECK(hipGetLastError(), "BottomUpSparseKernel launch failed");
            return retval;
        }
        const long long bu_steps = (words64 + oprtr::advance::kBottomUpStepWords - 1) / oprtr::advance::kBottomUpStepWords;
        long long grid = (bu_steps + 3) / 4;
        const long long cap = util::ResidentGrid(oprtr::advance::BottomUpKernel<256, 8, 32, PbfsProblem, L>, 256);
        if (grid > cap) grid = cap;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL((oprtr::advance::BottomUpKernel<256, 8, 32, PbfsProblem, L>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, stream,
                           b, ds, lookup);
        GR_CHECK(hipGetLastError(), "BottomUpKernel launch failed");
        return retval;
    }

    // local frontier bitmap -> queue (leaving bottom-up)
    hipError_t BitmapToQueue(unsigned *len, unsigned *edges)
    {
        hipError_t retval = hipSuccess;
        if ((retval = work_progress.ClearAux(stream))) return retval;
        hipLaunchKernelGGL((oprtr::advance::BitmapToQueueKernel<256, int, int>), dim3(cu_count * 4), dim3(256), 0, stream,
                           d_frontier_mask[cur_mask], n_local, queues[selector], work_progress.AuxTail(), work_progress.d_overflow,
                           d_row_offsets);
        GR_CHECK(hipGetLastError(), "BitmapToQueueKernel launch failed");
        if ((retval = work_progress.GetAux(frontier_len, frontier_edges, stream))) return retval;
        *len = frontier_len;
        *edges = frontier_edges;
        return retval;
    }

    // ================================================================================================================
    // The level loop inside the library
    // ================================================================================================================
    hipError_t PrepareSearch()
    {
        hipError_t retval = hipSuccess;
        if (!transport) return hipErrorNotInitialized;
        if (!d_small) {
            if (parts > kSmallMaxParts) return util::GRError(hipErrorInvalidValue, "Pbfs: more than 64 ranks", __FILE__, __LINE__);
            GR_CHECK(hipMalloc(&d_small, sizeof(unsigned) * kSmallWords), "Pbfs hipMalloc failed");
            GR_CHECK(hipHostMalloc(&h_small, sizeof(unsigned) * (kMailWords + 4), hipHostMallocMapped), "Pbfs hipHostMalloc failed");
            std::memset(h_small, 0, sizeof(unsigned) * (kMailWords + 4));
            GR_CHECK(hipMalloc(&d_visited_before, sizeof(unsigned) * (MaskWords(n_local) + 2)), "Pbfs hipMalloc failed");
            GR_CHECK(hipMalloc(&d_gathered, sizeof(unsigned) * static_cast<size_t>(parts) * (MaskWords(n_local_max) + 2)), "Pbfs hipMalloc failed");
            // a rank forwards every vertex at most once per search, so at most parts * (my vertices) ids arrive per search
            recv_capacity = parts * n_local_max + 1024;
            GR_CHECK(hipMalloc(&d_recv, sizeof(int) * 2 * static_cast<size_t>(recv_capacity)), "Pbfs hipMalloc failed");
            GR_CHECK(hipEventCreate(&ev_start), "Pbfs hipEventCreate failed");
            GR_CHECK(hipEventCreate(&ev_stop), "Pbfs hipEventCreate failed");
            sc.assign(parts, 0); so.assign(parts, 0); rc.assign(parts, 0); ro.assign(parts, 0);
        }
        if (!ds.d_mark) {  // marked levels: bytes and bits, owner-major (a slice = whole 32-bit words of bits)
            const unsigned slice_bits = static_cast<unsigned>(MaskWords(n_local_max)) * 32u;
            const size_t bytes = static_cast<size_t>(parts) * slice_bits;
            GR_CHECK(hipMalloc(&ds.d_mark, bytes), "Pbfs hipMalloc failed");
            GR_CHECK(hipMemsetAsync(ds.d_mark, 0, bytes, stream), "Pbfs memset failed");  // (every marked level leaves them zero again)
            GR_CHECK(hipMalloc(&d_mark_bits, bytes / 8), "Pbfs hipMalloc failed");
            ds.mark_stride = slice_bits;
            unsigned sh = 0;
            while ((1u << sh) < static_cast<unsigned>(parts)) ++sh;
            ds.mark_shift = 32 + sh;
            ds.mark_magic = (1ull << ds.mark_shift) / static_cast<unsigned>(parts) + 1ull;
        }
        if (mark_pred && !ds.d_pred_global) {
            GR_CHECK(hipMalloc(&ds.d_pred_global, sizeof(int) * static_cast<size_t>(n_global + 1)), "Pbfs hipMalloc failed");
            GR_CHECK(hipMalloc(&d_send_preds, sizeof(int) * static_cast<size_t>(candidate_capacity)), "Pbfs hipMalloc failed");
        }
        if (m_global < 0) {  // sum of the local edge counts (two 31-bit halves per rank through the word all-gather)
            const unsigned halves[2] = {static_cast<unsigned>(m_local) & 0xFFFFu, static_cast<unsigned>(m_local) >> 16};
            GR_CHECK(hipMemcpyAsync(d_small, halves, sizeof(halves), hipMemcpyHostToDevice, stream), "Pbfs copy failed");
            if (transport->AllGather(d_small, d_small + kSmallGathered, 2, stream)) return hipErrorUnknown;
            GR_CHECK(hipMemcpyAsync(h_small, d_small + kSmallGathered, sizeof(unsigned) * 2 * parts, hipMemcpyDeviceToHost, stream), "Pbfs copy failed");
            GR_CHECK(hipStreamSynchronize(stream), "Pbfs sync failed");
            m_global = 0;
            for (int p = 0; p < parts; ++p) m_global += static_cast<long long>(h_small[2 * p]) + (static_cast<long long>(h_small[2 * p + 1]) << 16);
        }
        return retval;
    }

    // `count` device words, `stride` words apart, to h_small[0..count): one tiny kernel + a host spin on its sequence word
    hipError_t Mail(const unsigned *d_src, int count, int stride)
    {
        hipError_t retval = hipSuccess;
        unsigned long long *h_seq = reinterpret_cast<unsigned long long *>(h_small + kMailWords);
        ++mail_seq;
        hipLaunchKernelGGL(MailKernel, dim3(1), dim3(64), 0, stream, d_src, count, stride, h_small, h_seq, mail_seq);
        GR_CHECK(hipGetLastError(), "MailKernel launch failed");
        unsigned spins = 0;
        while (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) != mail_seq) {
            if ((++spins & 0x3FFu) == 0) {
                const hipError_t rc = hipStreamQuery(stream);
                if (rc == hipSuccess) {
                    if (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) == mail_seq) break;
                    continue;
                }
                if (rc != hipErrorNotReady) return util::GRError(rc, "Pbfs Mail: stream failed", __FILE__, __LINE__);
            }
        }
        return retval;
    }

    // all-gather of 2 words per rank (this rank's packed tail at `d_tail_word`) -> global frontier (vertices, edges)
    hipError_t GlobalTail(const unsigned long long *d_tail_word, unsigned long long &glen, unsigned long long &gedges, unsigned &my_len,
                          unsigned &my_edges)
    {
        hipError_t retval = hipSuccess;
        if (transport->AllGather(reinterpret_cast<const unsigned *>(d_tail_word), d_small + kSmallGathered, 2, stream)) return hipErrorUnknown;
        if ((retval = Mail(d_small + kSmallGathered, 2 * parts, 1))) return retval;
        glen = 0; gedges = 0;
        for (int p = 0; p < parts; ++p) {  // PackTail: low word = vertices, high word = edges
            glen += h_small[2 * p];
            gedges += h_small[2 * p + 1];
        }
        my_len = h_small[2 * rank];
        my_edges = h_small[2 * rank + 1];
        return retval;
    }

    // One search, every level inside this call.  levels_out = BSP levels executed (the reference's search_depth).
    hipError_t Search(int src, bool direction_optimizing, int *levels_out, float *elapsed_ms)
    {
        hipError_t retval = hipSuccess;
        if ((retval = PrepareSearch())) return retval;
        if (src < 0 || src >= n_global) return hipErrorInvalidValue;
        GR_CHECK(hipEventRecord(ev_start, stream), "Pbfs hipEventRecord failed");
        // ---- reset + seed (the reference times Reset outside Enact; here it is inside the call and the bench says so) ----
        // (direction-optimizing = symmetric graph: edgeless vertices start out "visited", the sweeps skip them)
        never_applied = direction_optimizing;
        hipLaunchKernelGGL(PbfsResetKernel, dim3(cu_count * 8), dim3(256), 0, stream, src, parts, rank, d_row_offsets, ds.d_labels, ds.d_preds,
                           static_cast<long long>(n_local), ds.d_visited_mask, d_visited_before, direction_optimizing ? d_never : nullptr,
                           static_cast<long long>(MaskWords(n_local) + 2), ds.d_sent_mask, static_cast<long long>(MaskWords(n_global) + 2),
                           queues[0], work_progress.d_tail, util::WorkProgress::kSlots, work_progress.d_overflow, work_progress.d_wide,
                           util::WorkProgress::kWideLines * util::WorkProgress::kWideStride);
        GR_CHECK(hipGetLastError(), "PbfsResetKernel launch failed");
        selector = 0; cur_mask = 0; level = 0;
        unsigned long long glen = 0, gedges = 0;
        if ((retval = GlobalTail(work_progress.d_tail + 0, glen, gedges, frontier_len, frontier_edges))) return retval;
        long long unexplored = m_global;
        int levels = 0;
        const int wpr = MaskWords(n_local_max);  // bitmap words per rank; two trailing words carry the frontier size
        const int wpr_local = MaskWords(n_local);
        long long local_found = frontier_len;  // local vertices with edges discovered so far (sizes the compacting-sweep test)
        bool force_bottom_up = false;          // the last level ran count-only: its discoveries exist as a bitmap only
        while (glen > 0) {
            if (direction_optimizing && (force_bottom_up || static_cast<double>(gedges) * alpha > static_cast<double>(unexplored))) {
                // ---- bottom-up to the end: ONE collective per level, the all-gather of the frontier bitmaps ----
                // the local frontier = what the last top-down level added to the local visited bitmap (zero-degree finds
                // included: nobody can adopt them as parent); at level 0 the snapshot is the empty bitmap
                GR_CHECK(hipMemsetAsync(work_progress.d_tail + 1, 0, sizeof(unsigned long long), stream), "Pbfs clear tail failed");
                GR_CHECK(hipMemsetAsync(d_frontier_mask[cur_mask] + wpr, 0, sizeof(unsigned) * 2, stream), "Pbfs memset failed");
                hipLaunchKernelGGL(oprtr::advance::BitmapDiffKernel, dim3(cu_count * 2), dim3(256), 0, stream,
                                   reinterpret_cast<const unsigned long long *>(ds.d_visited_mask),
                                   reinterpret_cast<const unsigned long long *>(d_visited_before),
                                   reinterpret_cast<unsigned long long *>(d_frontier_mask[cur_mask]), static_cast<long long>(wpr_local / 2));
                GR_CHECK(hipGetLastError(), "BitmapDiffKernel launch failed");
                if (wpr > wpr_local)
                    GR_CHECK(hipMemsetAsync(d_frontier_mask[cur_mask] + wpr_local, 0, sizeof(unsigned) * (wpr - wpr_local), stream), "Pbfs memset failed");
                GR_CHECK(hipMemcpyAsync(d_frontier_mask[cur_mask] + wpr, &frontier_len, sizeof(unsigned), hipMemcpyHostToDevice, stream),
                         "Pbfs copy failed");
                int bottom_up_levels = 0;
                for (;;) {
                    if (transport->AllGather(d_frontier_mask[cur_mask], d_gathered, static_cast<size_t>(wpr + 2), stream)) return hipErrorUnknown;
                    if ((retval = Mail(d_gathered + wpr, parts, wpr + 2))) return retval;
                    unsigned long long total = 0;
                    for (int p = 0; p < parts; ++p) total += h_small[p];
                    if (total == 0) break;
                    if (bottom_up_levels > 0) local_found += h_small[rank];  // (level 0's frontier was counted when it was discovered)
                    ++bottom_up_levels;
                    if ((retval = LaunchSweep(d_gathered, wpr + 2, with_in_edges - local_found))) return retval;
                    // this level's finds = the next frontier's size: folded on the device into the bitmap's trailing word
                    hipLaunchKernelGGL(FoldWideKernel, dim3(1), dim3(64), 0, stream, work_progress.d_wide, d_frontier_mask[cur_mask ^ 1] + wpr,
                                       work_progress.d_tail + 1);
                    GR_CHECK(hipGetLastError(), "FoldWideKernel launch failed");
                    cur_mask ^= 1;
                    ++level;
                    ++levels;
                }
                break;
            }
            // ---- top-down level: advance -> bucket by owner -> counts all-gather -> ids (+ parents) all-to-all -> filter ----
            unexplored -= static_cast<long long>(gedges);
            GR_CHECK(hipMemcpyAsync(d_visited_before, ds.d_visited_mask, sizeof(unsigned) * (wpr_local + 2), hipMemcpyDeviceToDevice, stream),
                     "Pbfs visited snapshot failed");
            if (direction_optimizing && !mark_pred && lite_factor > 0 &&
                static_cast<double>(gedges) * alpha * lite_factor > static_cast<double>(unexplored + static_cast<long long>(gedges))) {
                // ---- count-only ("marked") level: mark -> bits -> ONE all-to-all of bitmap slices -> OR into visited ----
                const int mark_words = MaskWords(n_local_max);  // 32-bit words per owner slice
                if (frontier_len > 0) {
                    oprtr::advance::AdvanceArgs<int, int> args;
                    args.in = queues[selector];
                    args.out = util::Frontier<int, int>();
                    args.in_len = static_cast<int>(frontier_len);
                    args.in_edges = static_cast<int>(frontier_edges);
                    args.d_row_offsets = d_row_offsets;
                    args.d_column_indices = d_col_indices;
                    args.d_tail_out = nullptr;
                    args.d_tail_clear = nullptr;
                    args.d_overflow = work_progress.d_overflow;
                    ds.iteration = level;
                    typedef oprtr::advance::KernelPolicy<256, 4, 8, oprtr::advance::LB> MarkPolicy;
                    if ((retval = oprtr::advance::LaunchKernel<MarkPolicy, PbfsProblem, MarkFunctor, true, true>(args, ds, 0, stream))) return retval;
                }
                const long long all_words = static_cast<long long>(parts) * mark_words;
                hipLaunchKernelGGL(MarkBytesToBitsKernel, dim3(cu_count * 4), dim3(256), 0, stream, ds.d_mark, all_words, d_mark_bits);
                GR_CHECK(hipGetLastError(), "MarkBytesToBitsKernel launch failed");
                for (int p = 0; p < parts; ++p) {  // slice p (mark_words words) to rank p; the same from everybody
                    sc[p] = rc[p] = static_cast<size_t>(mark_words);
                    so[p] = ro[p] = static_cast<size_t>(p) * mark_words;
                }
                if (transport->AllToAllV(reinterpret_cast<const int *>(d_mark_bits), sc.data(), so.data(), reinterpret_cast<int *>(d_gathered),
                                         rc.data(), ro.data(), stream))
                    return hipErrorUnknown;
                hipLaunchKernelGGL(OrSlicesKernel, dim3(cu_count * 4), dim3(256), 0, stream, d_gathered, parts, static_cast<long long>(mark_words),
                                   static_cast<long long>(wpr_local), static_cast<long long>(n_local), ds.d_visited_mask, ds.d_labels, level + 1,
                                   work_progress.d_wide);
                GR_CHECK(hipGetLastError(), "OrSlicesKernel launch failed");
                // this rank's finds -> host (the bottom-up loop below writes it behind the frontier bitmap it builds from
                // visited XOR visited-before)
                hipLaunchKernelGGL(FoldWideKernel, dim3(1), dim3(64), 0, stream, work_progress.d_wide, d_small + kSmallGathered,
                                   static_cast<unsigned long long *>(nullptr));
                GR_CHECK(hipGetLastError(), "FoldWideKernel launch failed");
                if ((retval = Mail(d_small + kSmallGathered, 1, 1))) return retval;
                frontier_len = h_small[0];
                frontier_edges = 0;
                local_found += frontier_len;
                ++level;
                ++levels;
                ++marked_levels;
                force_bottom_up = true;  // (glen stays as it is: whether anything was found shows in the first all-gather below)
                continue;
            }
            hipLaunchKernelGGL(PbfsArmLevelKernel, dim3(1), dim3(256), 0, stream, work_progress.d_tail, d_small, work_progress.d_overflow);
            GR_CHECK(hipGetLastError(), "PbfsArmLevelKernel launch failed");
            if (frontier_len > 0) {
                oprtr::advance::AdvanceArgs<int, int> args;
                args.in = queues[selector];
                args.out = util::Frontier<int, int>();
                args.out.v = d_candidates;
                args.out.capacity = candidate_capacity;
                args.in_len = static_cast<int>(frontier_len);
                args.in_edges = static_cast<int>(frontier_edges);
                args.d_row_offsets = d_row_offsets;
                args.d_column_indices = d_col_indices;
                args.d_tail_out = work_progress.d_tail + 0;
                args.d_tail_clear = nullptr;
                args.d_overflow = work_progress.d_overflow;
                ds.iteration = level;
                typedef oprtr::advance::KernelPolicy<256, 8, 3, oprtr::advance::LB> Policy;
                if ((retval = oprtr::advance::LaunchKernel<Policy, PbfsProblem, SendFunctor, false>(args, ds, 0, stream))) return retval;
                // (the candidate count stays on the device: the bucketing kernels read it there)
                const int grid = cu_count * 4;
                hipLaunchKernelGGL(CountOwnersDeviceKernel, dim3(grid), dim3(256), 0, stream, d_candidates, work_progress.d_tail + 0, parts, d_small + kSmallCounts);
                GR_CHECK(hipGetLastError(), "CountOwnersDeviceKernel launch failed");
                hipLaunchKernelGGL(ScatterByOwnerDeviceKernel, dim3(grid), dim3(256), 0, stream, d_candidates, work_progress.d_tail + 0, parts,
                                   d_small + kSmallCounts, d_small + kSmallCursors, d_send, mark_pred ? ds.d_pred_global : nullptr, mark_pred ? d_send_preds : nullptr);
                GR_CHECK(hipGetLastError(), "ScatterByOwnerDeviceKernel launch failed");
            }
            // P x (1 + P) matrix: row p = rank p's error word, then what rank p sends to everybody
            const int row = parts + 1;
            if (transport->AllGather(d_small, d_small + kSmallGathered, static_cast<size_t>(row), stream)) return hipErrorUnknown;
            if ((retval = Mail(d_small + kSmallGathered, parts * row, 1))) return retval;
            size_t send_total = 0, recv_total = 0;
            for (int p = 0; p < parts; ++p) {
                sc[p] = h_small[rank * row + 1 + p];
                so[p] = send_total;
                send_total += sc[p];
                rc[p] = h_small[p * row + 1 + rank];
                ro[p] = recv_total;
                recv_total += rc[p];
            }
            // Errors every rank can read off the same matrix, so that ALL ranks leave here together and nobody is left waiting
            // in the next collective: a rank whose queues overflowed on an earlier level (its error word), or a rank that would
            // receive more ids than its buffer holds (column sums; the capacity is the same on every rank).
            for (int p = 0; p < parts; ++p) {
                size_t into_p = 0;
                for (int q = 0; q < parts; ++q) into_p += h_small[q * row + 1 + p];
                if (h_small[p * row] != 0)
                    return util::GRError(hipErrorInvalidConfiguration, p == rank ? "Frontier queue overflow. Please increase queue-sizing factor."
                                                                                 : "Pbfs: a peer rank reported a frontier queue overflow", __FILE__, __LINE__);
                if (into_p > static_cast<size_t>(recv_capacity))
                    return util::GRError(hipErrorInvalidConfiguration, p == rank ? "Pbfs receive buffer overflow" : "Pbfs: a peer rank's receive buffer would overflow",
                                         __FILE__, __LINE__);
            }
            if (transport->AllToAllV(d_send, sc.data(), so.data(), d_recv, rc.data(), ro.data(), stream)) return hipErrorUnknown;
            if (mark_pred && transport->AllToAllV(d_send_preds, sc.data(), so.data(), d_recv + recv_capacity, rc.data(), ro.data(), stream))
                return hipErrorUnknown;
            if (recv_total > 0) {
                oprtr::filter::FilterArgs<int, int> f;
                f.d_in = d_recv;
                f.num_elements = static_cast<int>(recv_total);
                f.out = queues[selector ^ 1];
                f.d_tail_out = work_progress.d_tail + 1;
                f.d_tail_clear = nullptr;
                f.d_overflow = work_progress.d_overflow;
                f.d_row_offsets = d_row_offsets;
                ds.iteration = level;
                ds.d_recv_preds = mark_pred ? d_recv + recv_capacity : nullptr;
                typedef oprtr::filter::KernelPolicy<256, 4, 8> Policy;
                retval = oprtr::filter::LaunchKernel<Policy, PbfsProblem, ReceiveFunctor, true>(f, ds, cu_count * 8, stream);
                ds.d_recv_preds = nullptr;
                if (retval) return retval;
            }
            selector ^= 1;
            ++level;
            ++levels;
            if ((retval = GlobalTail(work_progress.d_tail + 1, glen, gedges, frontier_len, frontier_edges))) return retval;
            local_found += frontier_len;
        }
        GR_CHECK(hipEventRecord(ev_stop, stream), "Pbfs hipEventRecord failed");
        GR_CHECK(hipEventSynchronize(ev_stop), "Pbfs hipEventSynchronize failed");
        if (elapsed_ms) GR_CHECK(hipEventElapsedTime(elapsed_ms, ev_start, ev_stop), "Pbfs hipEventElapsedTime failed");
        if (levels_out) *levels_out = levels;
        int overflow = 0;
        GR_CHECK(hipMemcpy(&overflow, work_progress.d_overflow, sizeof(int), hipMemcpyDeviceToHost), "Pbfs read overflow failed");
        if (overflow) return util::GRError(hipErrorInvalidConfiguration, "Frontier queue overflow. Please increase queue-sizing factor.", __FILE__, __LINE__);
        return retval;
    }
};

}  // namespace

struct grx_pbfs {
    Pbfs impl;
};

extern "C" {

int grx_pbfs_create(grx_pbfs **out, int device)
{
    if (!out) return -1;
    if (hipSetDevice(device) != hipSuccess) return -2;
    *out = new grx_pbfs();
    return 0;
}

int grx_pbfs_init_device(grx_pbfs *p, int n_global, int parts, int rank, int n_local, int m_local, int *d_row_offsets,
                         int *d_col_indices)
{
    if (!p || parts < 1 || parts > 64 || rank < 0 || rank >= parts || !d_row_offsets) return -1;
    return static_cast<int>(p->impl.Init(n_global, parts, rank, n_local, m_local, d_row_offsets, d_col_indices));
}

int grx_pbfs_reset(grx_pbfs *p, int src) { return p ? static_cast<int>(p->impl.Reset(src)) : -1; }

int grx_pbfs_frontier(grx_pbfs *p, unsigned *len, unsigned *edges)
{
    if (!p) return -1;
    if (len) *len = p->impl.frontier_len;
    if (edges) *edges = p->impl.frontier_edges;
    return 0;
}

int grx_pbfs_advance_local(grx_pbfs *p, unsigned *h_send_counts, int **d_send_buffer)
{
    if (!p || !h_send_counts) return -1;
    if (d_send_buffer) *d_send_buffer = p->impl.d_send;
    return static_cast<int>(p->impl.AdvanceLocal(h_send_counts));
}

int grx_pbfs_filter_received(grx_pbfs *p, const int *d_recv, int n_recv, unsigned *next_len, unsigned *next_edges)
{
    if (!p || !next_len || !next_edges) return -1;
    return static_cast<int>(p->impl.FilterReceived(d_recv, n_recv, next_len, next_edges));
}

int grx_pbfs_queue_to_bitmap(grx_pbfs *p) { return p ? static_cast<int>(p->impl.QueueToBitmap()) : -1; }

int grx_pbfs_frontier_bitmap(grx_pbfs *p, unsigned **d_bitmap, int *words)
{
    if (!p || !d_bitmap || !words) return -1;
    *d_bitmap = p->impl.d_frontier_mask[p->impl.cur_mask];
    *words = p->impl.MaskWords(p->impl.n_local_max);
    return 0;
}

int grx_pbfs_bottom_up(grx_pbfs *p, const unsigned *d_gathered, int words_per_rank, unsigned *found, unsigned *found_edges)
{
    if (!p || !d_gathered || !found || !found_edges) return -1;
    return static_cast<int>(p->impl.BottomUp(d_gathered, words_per_rank, found, found_edges));
}

int grx_pbfs_bitmap_to_queue(grx_pbfs *p, unsigned *len, unsigned *edges)
{
    if (!p || !len || !edges) return -1;
    return static_cast<int>(p->impl.BitmapToQueue(len, edges));
}

int grx_pbfs_labels(grx_pbfs *p, int **d_labels)
{
    if (!p || !d_labels) return -1;
    *d_labels = p->impl.ds.d_labels;
    return 0;
}

int grx_pbfs_preds(grx_pbfs *p, int **d_preds)
{
    if (!p || !d_preds) return -1;
    *d_preds = p->impl.ds.d_preds;
    return 0;
}

int grx_pbfs_set_option(grx_pbfs *p, const char *name, double value)
{
    if (!p || !name) return -1;
    const std::string key(name);
    if (key == "lite_factor") p->impl.lite_factor = value;
    else if (key == "alpha") p->impl.alpha = value;
    else if (key == "sparse_sweep_div") p->impl.sparse_sweep_div = static_cast<int>(value);
    else return 1;
    return 0;
}

long long grx_pbfs_stat(grx_pbfs *p, const char *name)
{
    if (!p || !name) return -1;
    const std::string key(name);
    if (key == "marked_levels") return p->impl.marked_levels;
    return -1;
}

int grx_rccl_load(void)
{
    return RcclApi::Get().Load() ? 0 : -2;  // dlopen + dlsym only: no communicator, nothing collective
}

int grx_rccl_unique_id(char id[128])
{
    if (!id) return -1;
    RcclApi &api = RcclApi::Get();
    if (!api.Load()) return -2;
    ncclUniqueIdBytes uid;
    std::memset(&uid, 0, sizeof(uid));
    const int rc = api.GetUniqueId(&uid);
    std::memcpy(id, uid.internal, 128);
    return rc;
}

int grx_pbfs_comm_init_rccl(grx_pbfs *p, const char id[128])
{
    if (!p || !id) return -1;
    RcclTransport *t = new RcclTransport();
    const int rc = t->Init(p->impl.parts, p->impl.rank, id);
    if (rc) {
        delete t;
        return rc;
    }
    delete p->impl.transport;
    p->impl.transport = t;
    p->impl.m_global = -1;
    return 0;
}

int grx_pbfs_set_transport(grx_pbfs *p, void *ctx, grx_all_gather_fn all_gather, grx_all_to_all_v_fn all_to_all_v)
{
    if (!p || !all_gather || !all_to_all_v) return -1;
    CallbackTransport *t = new CallbackTransport();
    t->ctx = ctx;
    t->gather = all_gather;
    t->a2a = all_to_all_v;
    delete p->impl.transport;
    p->impl.transport = t;
    p->impl.m_global = -1;
    return 0;
}

int grx_pbfs_set_options(grx_pbfs *p, int mark_pred, float alpha)
{
    if (!p) return -1;
    p->impl.mark_pred = mark_pred != 0;
    if (alpha > 0) p->impl.alpha = alpha;
    return 0;
}

int grx_pbfs_search(grx_pbfs *p, int src, int direction_optimizing, int *levels, float *elapsed_ms)
{
    if (!p) return -1;
    return static_cast<int>(p->impl.Search(src, direction_optimizing != 0, levels, elapsed_ms));
}

void grx_pbfs_destroy(grx_pbfs *p) { delete p; }

}  // extern "C"
