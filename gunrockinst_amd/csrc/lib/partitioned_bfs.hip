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
// The collectives live in gunrockinst_amd/multi_gpu.py (torch.distributed: RCCL on GPUs, gloo in CPU tests); this file
// only exposes the local steps through the C ABI and never communicates.
#include <gunrock/gunrock_mi355x.h>

#include <cstdio>
#include <vector>

#include <gunrock/app/enactor_base.hpp>
#include <gunrock/app/problem_base.hpp>
#include <gunrock/oprtr/advance/bottom_up.hpp>
#include <gunrock/oprtr/advance/kernel.hpp>
#include <gunrock/oprtr/filter/kernel.hpp>
#include <gunrock/util/memset_kernel.hpp>

using namespace gunrock;

namespace {

struct PbfsProblem {
    typedef int VertexId;
    typedef int SizeT;
    typedef int Value;
    static constexpr bool MARK_PREDECESSORS = false;
    struct DataSlice {
        int *d_labels;              // local
        int *d_preds;               // unused (MARK_PREDECESSORS = false)
        unsigned *d_visited_mask;   // local ids
        unsigned *d_sent_mask;      // GLOBAL ids: destinations this rank has already forwarded
        int iteration;
    };
};

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
    static __device__ __forceinline__ void ApplyEdge(int, int, DataSlice *, int = 0, int = 0) {}
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
    static __device__ __forceinline__ void ApplyFilter(int node, DataSlice *p, int = 0, int = 0)
    {
        p->d_labels[node] = p->iteration + 1;
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

struct Pbfs : app::EnactorBase {
    int parts = 1, rank = 0;
    int n_global = 0, n_local = 0, n_local_max = 0, m_local = 0;
    int *d_row_offsets = nullptr, *d_col_indices = nullptr;  // borrowed
    PbfsProblem::DataSlice ds{};
    unsigned *d_frontier_mask[2] = {nullptr, nullptr};
    int2 *d_heads = nullptr;
    util::Frontier<int, int> queues[2];
    int *d_candidates = nullptr, *d_send = nullptr;
    unsigned *d_counts = nullptr, *h_counts = nullptr;  // 2 * 64: counts, cursors
    int candidate_capacity = 0;
    hipStream_t stream = 0;
    int selector = 0, cur_mask = 0;
    unsigned frontier_len = 0, frontier_edges = 0;
    int level = 0;

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
        ds.d_preds = nullptr;
        for (int i = 0; i < 2; ++i) {
            GR_CHECK(hipMalloc(&d_frontier_mask[i], sizeof(unsigned) * (MaskWords(n_local_max) + 2)), "Pbfs hipMalloc failed");
            const size_t cap = nl + 1024;
            GR_CHECK(hipMalloc(&queues[i].v, sizeof(int) * cap), "Pbfs hipMalloc failed");
            GR_CHECK(hipMalloc(&queues[i].row_start, sizeof(int) * cap), "Pbfs hipMalloc failed");
            GR_CHECK(hipMalloc(&queues[i].scan, sizeof(int) * cap), "Pbfs hipMalloc failed");
            queues[i].capacity = static_cast<int>(cap);
        }
        GR_CHECK(hipMalloc(&d_heads, sizeof(int2) * nl), "Pbfs hipMalloc failed");
        if (n_local > 0) {
            // one wave per local vertex; the columns are global ids, so no degree table: the first two of the row
            hipLaunchKernelGGL((oprtr::advance::BuildHeadsKernel<int, int>), dim3((n_local + 3) / 4 < 8192 ? (n_local + 3) / 4 : 8192),
                               dim3(256), 0, stream, d_row_offsets, d_col_indices, static_cast<long long>(n_local), d_heads,
                               static_cast<const int *>(nullptr));
            GR_CHECK(hipGetLastError(), "BuildHeadsKernel launch failed");
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
        if (ds.d_labels) hipFree(ds.d_labels);
        if (ds.d_visited_mask) hipFree(ds.d_visited_mask);
        if (ds.d_sent_mask) hipFree(ds.d_sent_mask);
        for (int i = 0; i < 2; ++i) {
            if (d_frontier_mask[i]) hipFree(d_frontier_mask[i]);
            if (queues[i].v) hipFree(queues[i].v);
            if (queues[i].row_start) hipFree(queues[i].row_start);
            if (queues[i].scan) hipFree(queues[i].scan);
        }
        if (d_heads) hipFree(d_heads);
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
        util::Memset(ds.d_visited_mask, 0u, MaskWords(n_local) + 2, stream);
        util::Memset(ds.d_sent_mask, 0u, MaskWords(n_global) + 2, stream);
        for (int i = 0; i < 2; ++i) util::Memset(d_frontier_mask[i], 0u, MaskWords(n_local_max) + 2, stream);
        if ((retval = work_progress.Reset(stream))) return retval;
        selector = 0; cur_mask = 0; level = 0; frontier_len = 0; frontier_edges = 0;
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
                GR_CHECK(hipMemcpyAsync(ds.d_labels + local, &zero, sizeof(int), hipMemcpyHostToDevice, stream), "Pbfs seed failed");
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
        oprtr::advance::BottomUpArgs<int, int> b;
        b.nodes = n_local;
        b.d_inv_row_offsets = d_row_offsets;
        b.d_inv_column_indices = d_col_indices;
        b.d_inv_heads = d_heads;
        b.d_frontier_out = reinterpret_cast<unsigned long long *>(d_frontier_mask[cur_mask ^ 1]);
        b.d_visited = reinterpret_cast<unsigned long long *>(ds.d_visited_mask);
        b.d_tail_out = work_progress.d_tail + 1;
        b.d_tail_clear = nullptr;
        b.d_wide = work_progress.d_wide;  // per-workgroup counts spread over 32 lines, folded by the read-back
        ds.iteration = level;
        oprtr::advance::StripedBitmapLookup<int> lookup{d_gathered, static_cast<unsigned>(parts), static_cast<unsigned>(words_per_rank)};
        const long long bu_steps = ((static_cast<long long>(n_local) + 63) / 64 + oprtr::advance::kBottomUpStepWords - 1) / oprtr::advance::kBottomUpStepWords;
        long long grid = (bu_steps + 3) / 4;
        const long long cap = util::ResidentGrid(oprtr::advance::BottomUpKernel<256, 8, 32, PbfsProblem, oprtr::advance::StripedBitmapLookup<int>>, 256);
        if (grid > cap) grid = cap;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL((oprtr::advance::BottomUpKernel<256, 8, 32, PbfsProblem, oprtr::advance::StripedBitmapLookup<int>>),
                           dim3(static_cast<unsigned>(grid)), dim3(256), 0, stream, b, ds, lookup);
        GR_CHECK(hipGetLastError(), "BottomUpKernel launch failed");
        if ((retval = work_progress.GetTailWide(1, frontier_len, frontier_edges, stream))) return retval;
        cur_mask ^= 1;
        ++level;
        *found = frontier_len;
        *found_edges = frontier_edges;
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

void grx_pbfs_destroy(grx_pbfs *p) { delete p; }

}  // extern "C"
