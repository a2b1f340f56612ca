// app/bfs/bfs_problem.hpp -- device data for breadth-first search.
//
// Same contract as the reference's BFSProblem (gunrock/app/bfs/bfs_problem.cuh:41-364):
//   DataSlice { d_labels, d_preds, d_visited_mask }          (:57-63)
//   Init(stream_from_host, graph, num_gpus)                  (:188-261)
//   Reset(src, frontier_type, queue_sizing): labels = -1, preds = -2, mask = 0; then the source gets
//        label 0, pred -1 and is queued                      (:272-360)
//   Extract(h_labels, h_preds)                               (:144-177)
// MI355X-first differences:
//   * d_visited_mask is a bitmap of 32-bit words (n/32 words: 2 MiB at scale-24, resident in every
//     XCD's 4 MiB L2) updated with agent-scope atomicOr, so each vertex is discovered exactly once in
//     every mode -- the reference's byte mask with non-atomic RMW (filter/cta.cuh:166-207) tolerates
//     duplicate discovery, which costs redundant edge expansion on the next level;
//   * it is allocated in all four (mark_pred, idempotence) modes, and d_preds whenever MARK_PREDECESSORS
//     (the reference drops preds in idempotent+pred mode and stores garbage in labels, SURVEY appendix C;
//     here that mode yields depth labels AND valid parents);
//   * the current BSP iteration travels inside the by-value DataSlice kernel argument.
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/app/problem_base.hpp>
#include <gunrock/graphio/device_csr.hpp>
#include <gunrock/oprtr/advance/binned.hpp>
#include <gunrock/oprtr/advance/bottom_up.hpp>
#include <gunrock/util/memset_kernel.hpp>

namespace gunrock {
namespace app {
namespace bfs {

// bit v = (v has no in-edge); one wave per 64 vertices, the word is written whole
template <typename SizeT>
__global__ void NoInEdgeMaskKernel(const SizeT *d_inv_row_offsets, long long nodes, long long words64, unsigned long long *d_mask)
{
    const unsigned lane = threadIdx.x & 63;
    const long long wave0 = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) / 64;
    const long long nwaves = static_cast<long long>(gridDim.x) * blockDim.x / 64;
    for (long long w = wave0; w < words64; w += nwaves) {
        const long long v = w * 64 + lane;
        const bool none = v < nodes && d_inv_row_offsets[v + 1] == d_inv_row_offsets[v];
        const unsigned long long m = __ballot(none);
        if (lane == 0) d_mask[w] = m;
    }
}

// vertices WITH in-edges per 64-vertex word (bits past the vertex count do not count)
static __global__ void WithInEdgesCountKernel(const unsigned long long *d_never, long long nodes, long long words64, unsigned *d_counts)
{
    const long long w = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (w >= words64) return;
    unsigned long long with_edges = ~d_never[w];
    const long long first = w * 64;
    if (first >= nodes) with_edges = 0;
    else if (first + 64 > nodes) with_edges &= (1ull << (nodes - first)) - 1ull;
    d_counts[w] = static_cast<unsigned>(__popcll(with_edges));
}

// Reset in one launch: 16-byte stores for labels / preds, the source patched in flight, queue entry 0 seeded.
// fill_labels = 0: labels are deferred (BFSProblem::labels_deferred) -- only the source's label is written here and
// EmitLabelsKernel writes every label once, at the end of Enact.
template <typename VertexId, typename SizeT, bool PRED>
__global__ void BfsResetKernel(VertexId *d_labels, VertexId *d_preds, unsigned *d_visited, const unsigned *d_never,
                               unsigned *d_snapshot, int fill_labels, long long nodes,
                               long long mask_words, VertexId src, const SizeT *d_row_offsets,
                               util::Frontier<VertexId, SizeT> queue0, SizeT *h_src_row, unsigned long long *h_src_seq,
                               unsigned long long seq, unsigned long long *d_arm_tail, int *d_arm_overflow, unsigned long long *d_arm_wide,
                               int *d_arm_log)
{
    // d_arm_*: the device words of the enactor that ran the last search on this problem (util::WorkProgress): zeroing them and
    // seeding ring slot 0 with the source's (1, degree) HERE saves that enactor its own arming kernel and the host wait for the
    // source's degree in front of it (~8 us at the start of every Enact)
    typedef __attribute__((ext_vector_type(4))) int V4;
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    const long long tid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const long long nvec = nodes / 4;
    V4 *labels4 = reinterpret_cast<V4 *>(d_labels);
    V4 *preds4 = reinterpret_cast<V4 *>(d_preds);
    if (fill_labels || PRED) {
        for (long long k = tid; k < nvec; k += stride) {
            V4 l = {-1, -1, -1, -1};
            V4 p = {-2, -2, -2, -2};
            if (src >= 0 && k == src / 4) {
                l[src & 3] = 0;
                p[src & 3] = -1;
            }
            if (fill_labels) labels4[k] = l;
            if (PRED) preds4[k] = p;
        }
        for (long long i = nvec * 4 + tid; i < nodes; i += stride) {
            if (fill_labels) d_labels[i] = (i == src) ? 0 : -1;
            if (PRED) d_preds[i] = (i == src) ? -1 : -2;
        }
    }
    if (!fill_labels && tid == 0 && src >= 0) d_labels[src] = 0;
    for (long long w = tid; w < mask_words; w += stride)
    {
        const unsigned never = d_never ? d_never[w] : 0u;
        const unsigned src_bit = (src >= 0 && w == (src >> 5)) ? (1u << (src & 31)) : 0u;
        d_visited[w] = never | src_bit;
        // "visited before the search": what a direction switch at level 0 diffs against; the source must survive the diff
        if (d_snapshot) d_snapshot[w] = never & ~src_bit;
    }
    if (d_arm_tail && blockIdx.x == 0) {
        const int t = threadIdx.x;
        if (t >= 1 && t < util::WorkProgress::kSlots) d_arm_tail[t] = 0ull;  // (slot 0 below)
        if (t < util::WorkProgress::kWideLines)
            for (int k = 0; k < util::WorkProgress::kWideSets; ++k) d_arm_wide[(k * util::WorkProgress::kWideLines + t) * util::WorkProgress::kWideStride] = 0ull;
        if (t < 8) d_arm_log[t] = 0;
        if (t == 0) {
            *d_arm_overflow = 0;
            unsigned long long seed = 0ull;
            if (src >= 0) {
                const SizeT b = d_row_offsets[src], e = d_row_offsets[src + 1];
                if (e > b) seed = util::PackTail(1u, static_cast<unsigned>(e - b));
            }
            d_arm_tail[0] = seed;
        }
    }
    if (tid == 0 && src >= 0) {
        const SizeT begin = d_row_offsets[src], end = d_row_offsets[src + 1];
        queue0.v[0] = src;
        queue0.row_start[0] = begin;
        queue0.scan[0] = 0;
        // straight into pinned host memory, sequence word last: the host learns the source's degree while this kernel is
        // still filling labels, and queues the first level behind it without a stream synchronisation
        h_src_row[0] = begin;
        h_src_row[1] = end;
        __threadfence_system();
        __hip_atomic_store(h_src_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- deferred labels (direction-optimizing searches) ----
// A bottom-up sweep or the closing pass of a count-only level finds its vertices in VERTEX order but only some of every 16:
// writing their labels there is one partial 64-byte sector after the other (PMC, round 2: 2.3x write amplification, 34 us of
// a 350 us scale-24 search), on top of the 64 MB fill of -1 at Reset.  BFS depth = the level whose frontier bitmap holds the
// vertex, so those levels only KEEP their output bitmap (n/8 bytes each, a pool of kLevelMasks) and this kernel writes every
// label of the search once, 16 bytes per lane, at the end of Enact:
//   bit set in the bitmap of level L              -> L
//   visited, in no kept bitmap, has in-edges       -> what a top-down kernel wrote (they label at discovery: few vertices)
//   the source                                     -> 0 (Reset wrote it)
//   otherwise                                      -> -1
// FULL = false is the flush used when the pool runs out of bitmaps in the middle of a search: only the set bits are stored.
constexpr int kLevelMasks = 12;

template <typename VertexId>
struct LevelMaskList {
    const unsigned long long *mask[kLevelMasks];
    VertexId label[kLevelMasks];
    int count = 0;
    // a pass queued behind a chain of sweeps before the host knows how the chain ended: entries [chain_first, count) are the
    // chain's bitmaps in order, of which only the first d_gate[1] were filled; d_gate[0] == 0: do not run at all
    int chain_first = 0;
    const int *d_gate = nullptr;
};

// KMAX = bitmaps the kernel reads (entries past levels.count repeat entry 0: the same bit gives the same label).  All KMAX + 2
// word loads of a lane are issued before the first is looked at -- with a run-time loop over the list every bitmap was its own
// dependent round trip (33 us for the 64 MB of a scale-24 graph; the store stream alone is ~14 us).
template <typename VertexId, bool FULL, int KMAX>
__global__ __launch_bounds__(256) void EmitLabelsKernel(LevelMaskList<VertexId> levels, const unsigned long long *d_visited,
                                                        const unsigned long long *d_never, long long nodes, VertexId src, VertexId *d_labels)
{
    typedef __attribute__((ext_vector_type(4))) int V4;
    int valid = KMAX;  // entries that count (the padding repeats entry 0: harmless)
    if (levels.d_gate) {  // (uniform)
        if (levels.d_gate[0] == 0) return;
        valid = levels.chain_first + levels.d_gate[1];
    }
    const long long quads = (nodes + 3) / 4;  // four consecutive vertices per lane: one 16-byte store
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long q = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; q < quads; q += stride) {
        const long long v0 = q * 4;
        const long long w = v0 >> 6;
        const int sh = static_cast<int>(v0 & 63);
        unsigned long long mw[KMAX > 0 ? KMAX : 1];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) mw[k] = levels.mask[k][w];  // (the sixteen lanes of a word read the same address)
        unsigned long long vis = 0, nev = 0;
        if (FULL) {
            vis = d_visited[w];
            nev = d_never ? d_never[w] : 0ull;  // pre-marked "visited": never discovered
        }
        V4 out = {-1, -1, -1, -1};
        unsigned covered = 0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const unsigned m = (k < valid) ? static_cast<unsigned>(mw[k] >> sh) & 0xFu : 0u;
            covered |= m;
            const VertexId l = levels.label[k];
            if (m & 1u) out[0] = l;
            if (m & 2u) out[1] = l;
            if (m & 4u) out[2] = l;
            if (m & 8u) out[3] = l;
        }
        if (FULL) {
            unsigned keep = static_cast<unsigned>((vis & ~nev) >> sh) & 0xFu & ~covered;
            if (src >= v0 && src < v0 + 4) keep |= 1u << (src - v0);
            if (keep) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (((keep >> b) & 1u) && v0 + b < nodes) out[b] = d_labels[v0 + b];
            }
            if (v0 + 3 < nodes) *reinterpret_cast<V4 *>(d_labels + v0) = out;
            else
                for (int b = 0; b < 4 && v0 + b < nodes; ++b) d_labels[v0 + b] = out[b];
        } else if (covered) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (((covered >> b) & 1u) && v0 + b < nodes) d_labels[v0 + b] = out[b];
        }
    }
}

template <typename _VertexId, typename _SizeT, typename _Value, bool _MARK_PREDECESSORS,
          bool _ENABLE_IDEMPOTENCE, bool _USE_DOUBLE_BUFFER>
struct BFSProblem : ProblemBase<_VertexId, _SizeT, _Value, _USE_DOUBLE_BUFFER> {
    typedef ProblemBase<_VertexId, _SizeT, _Value, _USE_DOUBLE_BUFFER> Base;
    typedef _VertexId VertexId;
    typedef _SizeT SizeT;
    typedef _Value Value;
    static constexpr bool MARK_PREDECESSORS = _MARK_PREDECESSORS;
    static constexpr bool ENABLE_IDEMPOTENCE = _ENABLE_IDEMPOTENCE;

    struct DataSlice {
        VertexId *d_labels = nullptr;         // depth per vertex, -1 = unreached
        VertexId *d_preds = nullptr;          // parent per vertex (-2 unset, -1 source)
        unsigned *d_visited_mask = nullptr;   // 1 bit per vertex
        VertexId iteration = 0;               // current BSP level (labels written = iteration + 1)
        int lite = 0;                         // 1: count-only level: mark d_fresh[d] with a plain byte store, no claim, no label
        int defer_labels = 0;                 // 1: sweeps in vertex order leave the labels to EmitLabelsKernel (their bitmaps are kept)
        unsigned char *d_fresh = nullptr;     // one byte per vertex, all zero between levels (direction-optimizing only)
        // direction-optimizing traversal (reference app/dobfs: d_frontier_map_in/out, dobfs_problem.cuh):
        unsigned *d_frontier_mask[kLevelMasks] = {};        // 1 bit per vertex: pool of frontier bitmaps (current / next / kept levels)
        unsigned *d_snapshot = nullptr;                     // the visited bitmap as it was before the last top-down level
        unsigned *d_never_mask = nullptr;                   // vertices without in-edges: nothing can ever discover them
        unsigned *d_head_base = nullptr;                    // compacted heads: first entry of every 64-vertex word (bottom_up.hpp)
        int2 *d_inv_heads = nullptr;                        // first two in-neighbours per vertex (bottom_up.hpp)
        const SizeT *d_inv_row_offsets = nullptr;           // in-neighbour CSR (CSC of the graph)
        const VertexId *d_inv_column_indices = nullptr;
    };

    // direction-optimizing switches, names from the reference's DOBFS driver (tests/dobfs/test_dobfs.cu:530-534)
    bool direction_optimizing = false;
    float alpha = 10.0f;  // top-down -> bottom-up when frontier_edges * alpha > unexplored_edges (measured optimum 8..14 on R-MAT)
    float beta = 4000.0f; // bottom-up -> top-down when frontier_vertices * beta < nodes (measured at scale-24: 24..200 equal, 1000..4000 a little better;
                          // the reference's vertex rule, dobfs_enactor.cuh:569, with a later switch: sweeps of a nearly finished search are cheap)
    // traversal_mode 1 / low-degree graphs: a frontier of at most kTwcCapacity vertices and this many edges runs in the TWC
    // workgroup (oprtr/advance/twc.hpp); 0 = never
    int twc_edge_limit = 8192;
    // A top-down level runs "count only" (byte stores instead of claims, then bottom-up) when frontier_edges * alpha * lite_factor >
    // unexplored_edges.  700 = practically every level of a direction-optimizing search beyond ~40 K frontier edges at scale-24:
    // a claimed level of 0.3-0.5 M edges costs ~50 us (memory-side atomics at 27 G/s, queue entries, scattered labels, then a read-back,
    // a snapshot copy and a bitmap diff before the sweep), the same level counted costs ~15 us and the sweeps follow without a round
    // trip.  Measured (ms per scale-24 search): 12 -> 0.327, 100 -> 0.319, 500 -> 0.312, 700 -> 0.311, 2000 -> 0.323, 5000 -> 0.346 (then
    // an 11 K-edge level turns bottom-up with a frontier of a few thousand vertices: too early).
    float lite_factor = 700.0f;
    // direction-optimizing: a level that would run count-only or bottom-up and has between min and max frontier edges starts
    // with a heads-only bottom-up pass.  -1 = automatic: edges/30 .. edges/7.8 (measured on R-MAT scale-24: below, the plain
    // count-only level is cheaper; above, the frontier is dense enough for the full bottom-up sweep to win); min 0 = never,
    // max 0 = no upper bound.
    int head_pass_min_edges = -1;
    int head_pass_max_edges = -1;
    long long with_in_edges = 0;  // vertices that have in-edges (only they can ever be discovered bottom-up)
    // a bottom-up level runs the compacting sweep (BottomUpSparseKernel) when at most nodes / sparse_sweep_div such vertices
    // can still be unvisited (0 = never)
    int sparse_sweep_div = 6;  // (measured at scale-24 with 64-word chunks: 16 -> 0.348, 8 -> 0.324, 6 -> 0.321, 4 -> 0.326, 2 -> 0.45 ms per search)
    // ... and that sweep also writes its finds as the next top-down queue when its input frontier is within this factor of the
    // switch-back threshold (frontier * beta < factor * nodes): the switch then needs no bitmap -> queue pass
    float emit_queue_factor = 128.0f;
    bool chain_closing = true;     // the closing top-down levels (and the label pass) are queued behind a chain of sweeps, gated on the device
    bool speculative_emit = true;  // deferred labels: the emit pass is queued right behind the closing top-down launch
    int chain_sweeps = 4;          // bottom-up sweeps queued per host round trip (BottomUpAutoKernel decides on the device what each
                                   // of them does); 0 = one sweep per round trip, chosen by the host
    long long HeadPassMin() const { return head_pass_min_edges >= 0 ? head_pass_min_edges : static_cast<long long>(this->edges) / 30 + 1; }
    long long HeadPassMax() const
    {
        if (head_pass_max_edges > 0) return head_pass_max_edges;
        if (head_pass_max_edges == 0) return 1ll << 40;
        return static_cast<long long>(static_cast<double>(this->edges) / 7.8);
    }
    // Top-down levels with at least this many frontier edges run as a destination-binned advance (oprtr/advance/binned.hpp):
    // expand + screen, claims on the destination's owner XCD without atomics, then a vertex-ordered closing sweep
    // (FreshToBitmapKernel + BitmapToQueueKernel) that labels, dedupes and enqueues.  0 = never.
    long long binned_min_edges = 1ll << 23;
    oprtr::advance::BinPoolStorage<VertexId> bin_pool;
    bool cooperative_launch = false;  // persistent levels kernel through hipLaunchCooperativeKernel (launch-time residency check)
    int persistent_edge_limit = 1 << 20;  // ... and up to this many inside the persistent multi-workgroup kernel (0 = off)
    int tail_edge_limit = 8192;  // levels with at most this many edge slots run inside the single-workgroup tail kernel

    DataSlice **data_slices = nullptr;  // host copies (by-value kernel arguments), one per GPU
    DataSlice **d_data_slices = nullptr;  // kept for source compatibility; unused (no device-side struct)

    BFSProblem() {}

    ~BFSProblem() override
    {
        if (data_slices) {
            DataSlice *ds = data_slices[0];
            if (ds) {
                if (ds->d_labels) util::GRError(hipFree(ds->d_labels), "BFSProblem hipFree d_labels failed", __FILE__, __LINE__);
                if (ds->d_preds) util::GRError(hipFree(ds->d_preds), "BFSProblem hipFree d_preds failed", __FILE__, __LINE__);
                if (ds->d_visited_mask) util::GRError(hipFree(ds->d_visited_mask), "BFSProblem hipFree d_visited_mask failed", __FILE__, __LINE__);
                for (int i = 0; i < kLevelMasks; ++i)
                    if (ds->d_frontier_mask[i]) util::GRError(hipFree(ds->d_frontier_mask[i]), "BFSProblem hipFree d_frontier_mask failed", __FILE__, __LINE__);
                if (ds->d_snapshot) util::GRError(hipFree(ds->d_snapshot), "BFSProblem hipFree d_snapshot failed", __FILE__, __LINE__);
                if (ds->d_never_mask) util::GRError(hipFree(ds->d_never_mask), "BFSProblem hipFree d_never_mask failed", __FILE__, __LINE__);
                if (ds->d_fresh) util::GRError(hipFree(ds->d_fresh), "BFSProblem hipFree d_fresh failed", __FILE__, __LINE__);
                if (ds->d_head_base) util::GRError(hipFree(ds->d_head_base), "BFSProblem hipFree d_head_base failed", __FILE__, __LINE__);
                if (ds->d_inv_heads) util::GRError(hipFree(ds->d_inv_heads), "BFSProblem hipFree d_inv_heads failed", __FILE__, __LINE__);
                delete ds;
            }
            delete[] data_slices;
        }
        if (h_src_box) util::GRError(hipHostFree(h_src_box), "BFSProblem hipHostFree failed", __FILE__, __LINE__);
        bin_pool.Release();
    }

    // bitmaps are sized in whole 64-bit words: one wave owns one word in the bottom-up sweep
    SizeT MaskWords() const { return ((this->nodes + 63) / 64) * 2; }

    // Enable direction-optimizing traversal.  The in-neighbour CSR must stay valid while the problem lives;
    // for an undirected (symmetric) graph pass the problem's own device arrays (InverseIsSelf()).
    hipError_t SetInverseGraph(const SizeT *d_inv_row_offsets, const VertexId *d_inv_column_indices,
                               float alpha_ = 0.0f, float beta_ = 0.0f)
    {
        hipError_t retval = hipSuccess;
        DataSlice *ds = data_slices[0];
        ds->d_inv_row_offsets = d_inv_row_offsets;
        ds->d_inv_column_indices = d_inv_column_indices;
        for (int i = 0; i < kLevelMasks; ++i)
            if (!ds->d_frontier_mask[i])
                GR_CHECK(hipMalloc(&ds->d_frontier_mask[i], sizeof(unsigned) * static_cast<size_t>(MaskWords() + 2)),
                         "BFSProblem hipMalloc d_frontier_mask failed");
        if (!ds->d_snapshot)
            GR_CHECK(hipMalloc(&ds->d_snapshot, sizeof(unsigned) * static_cast<size_t>(MaskWords() + 2)), "BFSProblem hipMalloc d_snapshot failed");
        // Static per graph: bit v set when v has no in-edge.  Reset preloads the visited bitmap with it, so the bottom-up
        // sweep skips those vertices (half of an R-MAT graph) without touching their row offsets, 64 at a time.
        if (!ds->d_never_mask)
            GR_CHECK(hipMalloc(&ds->d_never_mask, sizeof(unsigned) * static_cast<size_t>(MaskWords() + 2)),
                     "BFSProblem hipMalloc d_never_mask failed");
        if (!ds->d_fresh) {
            const size_t bytes = (static_cast<size_t>(this->nodes) + 1023) / 1024 * 1024 + 1024;  // FreshToBitmapKernel reads 1 KiB steps
            GR_CHECK(hipMalloc(&ds->d_fresh, bytes), "BFSProblem hipMalloc d_fresh failed");
            GR_CHECK(hipMemset(ds->d_fresh, 0, bytes), "BFSProblem hipMemset d_fresh failed");  // levels leave it zero again
            GR_CHECK(hipDeviceSynchronize(), "BFSProblem sync failed");  // (null-stream memset: the problem's stream is not ordered behind it)
        }
        if (!ds->d_inv_heads)
            GR_CHECK(hipMalloc(&ds->d_inv_heads, sizeof(int2) * static_cast<size_t>(this->nodes > 0 ? this->nodes : 1)),
                     "BFSProblem hipMalloc d_inv_heads failed");
        {
            const long long words64 = static_cast<long long>(MaskWords()) / 2 + 1;
            long long grid = (words64 + 3) / 4;
            if (grid > 2048) grid = 2048;
            hipLaunchKernelGGL((NoInEdgeMaskKernel<SizeT>), dim3(static_cast<unsigned>(grid)), dim3(256), 0,
                               this->graph_slices[0]->stream, d_inv_row_offsets, static_cast<long long>(this->nodes), words64,
                               reinterpret_cast<unsigned long long *>(ds->d_never_mask));
            GR_CHECK(hipGetLastError(), "NoInEdgeMaskKernel launch failed");
            // heads are stored only for vertices that have in-edges: d_head_base[w] = such vertices before word w
            if (!ds->d_head_base)
                GR_CHECK(hipMalloc(&ds->d_head_base, sizeof(unsigned) * static_cast<size_t>(words64 + 1)), "BFSProblem hipMalloc d_head_base failed");
            unsigned *d_counts = nullptr;
            unsigned long long *d_scan_sums = nullptr;
            GR_CHECK(hipMalloc(&d_counts, sizeof(unsigned) * static_cast<size_t>(words64 + 1)), "BFSProblem hipMalloc failed");
            GR_CHECK(hipMalloc(&d_scan_sums, sizeof(unsigned long long) * static_cast<size_t>(graphio::ScanScratchWords(words64))),
                     "BFSProblem hipMalloc failed");
            hipLaunchKernelGGL(WithInEdgesCountKernel, dim3(static_cast<unsigned>((words64 + 255) / 256)), dim3(256), 0,
                               this->graph_slices[0]->stream, reinterpret_cast<const unsigned long long *>(ds->d_never_mask),
                               static_cast<long long>(this->nodes), words64, d_counts);
            GR_CHECK(hipGetLastError(), "WithInEdgesCountKernel launch failed");
            GR_CHECK(graphio::DeviceExclusiveScan<unsigned>(d_counts, ds->d_head_base, words64, d_scan_sums, this->graph_slices[0]->stream),
                     "BFSProblem head-base scan failed");
            GR_CHECK(hipStreamSynchronize(this->graph_slices[0]->stream), "NoInEdgeMaskKernel failed");
            {   // vertices that have in-edges = last base + last count (sizes the "nearly finished" test of the enactor)
                unsigned last_base = 0, last_count = 0;
                GR_CHECK(hipMemcpy(&last_base, ds->d_head_base + (words64 - 1), sizeof(unsigned), hipMemcpyDeviceToHost), "BFSProblem read failed");
                GR_CHECK(hipMemcpy(&last_count, d_counts + (words64 - 1), sizeof(unsigned), hipMemcpyDeviceToHost), "BFSProblem read failed");
                with_in_edges = static_cast<long long>(last_base) + last_count;
            }
            GR_CHECK(hipFree(d_counts), "BFSProblem hipFree failed");
            GR_CHECK(hipFree(d_scan_sums), "BFSProblem hipFree failed");
        }
        if (this->nodes > 0) {
            long long grid = (static_cast<long long>(this->nodes) + 3) / 4;  // one wave per vertex
            if (grid > 8192) grid = 8192;
            hipLaunchKernelGGL((oprtr::advance::BuildHeadsKernel<VertexId, SizeT>), dim3(static_cast<unsigned>(grid)), dim3(256), 0,
                               this->graph_slices[0]->stream, d_inv_row_offsets, d_inv_column_indices,
                               static_cast<long long>(this->nodes), ds->d_inv_heads, d_inv_row_offsets,
                               reinterpret_cast<const unsigned long long *>(ds->d_never_mask), ds->d_head_base);
            GR_CHECK(hipGetLastError(), "BuildHeadsKernel launch failed");
            GR_CHECK(hipStreamSynchronize(this->graph_slices[0]->stream), "BuildHeadsKernel failed");
        }

        if (alpha_ > 0) alpha = alpha_;
        if (beta_ > 0) beta = beta_;
        direction_optimizing = true;
        return retval;
    }
    // Scratch of the binned advance: the chunk pool (sized for a level that hands over every edge of the graph from
    // `workgroups` workgroups), the flag bytes and the bitmap its closing sweep writes.  Allocated on first use.
    hipError_t EnsureBinned(int workgroups, int *d_overflow)
    {
        hipError_t retval = hipSuccess;
        DataSlice *ds = data_slices[0];
        if (!bin_pool.Ready() || bin_pool.workgroups < workgroups)
            GR_CHECK(bin_pool.Allocate(static_cast<long long>(this->edges), workgroups, MARK_PREDECESSORS, d_overflow),
                     "BFSProblem bin pool allocation failed");
        if (!ds->d_fresh) {
            const size_t bytes = (static_cast<size_t>(this->nodes) + 1023) / 1024 * 1024 + 1024;  // FreshToBitmapKernel reads 1 KiB steps
            GR_CHECK(hipMalloc(&ds->d_fresh, bytes), "BFSProblem hipMalloc d_fresh failed");
            GR_CHECK(hipMemset(ds->d_fresh, 0, bytes), "BFSProblem hipMemset d_fresh failed");  // levels leave it zero again
            GR_CHECK(hipDeviceSynchronize(), "BFSProblem sync failed");  // (null-stream memset: the problem's stream is not ordered behind it)
        }
        if (!ds->d_frontier_mask[0])
            GR_CHECK(hipMalloc(&ds->d_frontier_mask[0], sizeof(unsigned) * static_cast<size_t>(MaskWords() + 2)),
                     "BFSProblem hipMalloc d_frontier_mask failed");
        return retval;
    }

    hipError_t InverseIsSelf(float alpha_ = 0.0f, float beta_ = 0.0f)
    {
        return SetInverseGraph(this->graph_slices[0]->d_row_offsets, this->graph_slices[0]->d_column_indices, alpha_, beta_);
    }

    hipError_t AllocData()
    {
        hipError_t retval = hipSuccess;
        data_slices = new DataSlice *[1];
        data_slices[0] = new DataSlice();
        DataSlice *ds = data_slices[0];
        const size_t n = static_cast<size_t>(this->nodes > 0 ? this->nodes : 1);
        GR_CHECK(hipMalloc(&ds->d_labels, sizeof(VertexId) * n), "BFSProblem hipMalloc d_labels failed");
        if (MARK_PREDECESSORS)
            GR_CHECK(hipMalloc(&ds->d_preds, sizeof(VertexId) * n), "BFSProblem hipMalloc d_preds failed");
        GR_CHECK(hipMalloc(&ds->d_visited_mask, sizeof(unsigned) * static_cast<size_t>(MaskWords() + 2)),
                 "BFSProblem hipMalloc d_visited_mask failed");
        return retval;
    }

    hipError_t Init(bool stream_from_host, const Csr<VertexId, Value, SizeT> &graph, int num_gpus = 1)
    {
        hipError_t retval = hipSuccess;
        if ((retval = Base::Init(stream_from_host, graph, num_gpus))) return retval;
        return AllocData();
    }

    hipError_t InitFromDevice(SizeT nodes, SizeT edges, SizeT *d_row_offsets, VertexId *d_column_indices)
    {
        hipError_t retval = hipSuccess;
        if ((retval = Base::InitFromDevice(nodes, edges, d_row_offsets, d_column_indices))) return retval;
        return AllocData();
    }

    // One fused kernel (labels = -1, preds = -2, visited = 0, source seeded, queue seeded) and one 8-byte read-back of
    // the source's row extent -- the reference issues 3 <<<128,128>>> memsets and 3-4 tiny H2D copies (:298-357).
    hipError_t Reset(VertexId src, FrontierType frontier_type, double queue_sizing)
    {
        hipError_t retval = hipSuccess;
        if ((retval = Base::Reset(frontier_type, queue_sizing))) return retval;
        GraphSlice<VertexId, SizeT, Value> *gs = this->graph_slices[0];
        DataSlice *ds = data_slices[0];
        hipStream_t stream = gs->stream;
        if (!h_src_box) {
            GR_CHECK(hipHostMalloc(reinterpret_cast<void **>(&h_src_box), sizeof(SourceBox), hipHostMallocMapped), "BFSProblem hipHostMalloc failed");
            h_src_box->seq = 0;
        }
        ds->iteration = 0;
        ds->defer_labels = 0;
        // direction-optimizing problems defer the labels of their vertex-ordered sweeps: no fill here, one pass at the end of Enact
        labels_deferred = direction_optimizing && defer_labels;
        level_masks.count = 0;
        mask_ring = 0;
        emit_current = false;
        const bool valid = src >= 0 && src < this->nodes;
        const long long work = (static_cast<long long>(this->nodes) + 3) / 4;
        long long grid = (work + 255) / 256;
        if (grid > 2048) grid = 2048;
        if (grid < 1) grid = 1;
        ++reset_seq;
        if (!util::WorkProgress::IsLive(arm_progress)) arm_progress = nullptr;  // (that enactor is gone)
        hipLaunchKernelGGL((BfsResetKernel<VertexId, SizeT, MARK_PREDECESSORS>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, stream,
                           ds->d_labels, ds->d_preds, ds->d_visited_mask, direction_optimizing ? ds->d_never_mask : nullptr,
                           direction_optimizing ? ds->d_snapshot : nullptr, labels_deferred ? 0 : 1, static_cast<long long>(this->nodes),
                           static_cast<long long>(MaskWords() + 2), valid ? src : static_cast<VertexId>(-1), gs->d_row_offsets,
                           gs->frontier_queues[0], h_src_box->row, &h_src_box->seq, reset_seq,
                           arm_progress ? arm_progress->d_tail : nullptr, arm_progress ? arm_progress->d_overflow : nullptr,
                           arm_progress ? arm_progress->d_wide : nullptr, arm_progress ? arm_progress->d_chain_log : nullptr);
        GR_CHECK(hipGetLastError(), "BfsResetKernel launch failed");
        armed_progress = arm_progress;  // (the enactor that finds its own words armed skips its arming kernel)
        // No synchronisation here: everything that follows runs on the same stream, and SourceDegree() waits for the one
        // host-visible result (the source's row, which the kernel writes first).
        src_row[0] = src_row[1] = 0;
        src_row_pending = valid;
        source = src;
        return retval;
    }

    // Degree of the source, known after Reset (seeds the first packed tail).
    SizeT SourceDegree()
    {
        if (src_row_pending) {
            volatile unsigned long long *flag = &h_src_box->seq;
            unsigned spins = 0, idle_checks = 0;
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != reset_seq) {
                if ((++spins & 0x3FFu) == 0) {
                    const hipError_t rc = hipStreamQuery(this->graph_slices[0]->stream);
                    if (rc != hipSuccess && rc != hipErrorNotReady) {  // the reset kernel failed: report a zero-degree source
                        util::GRError(rc, "BFSProblem Reset kernel failed", __FILE__, __LINE__);
                        break;
                    }
                    if (rc == hipSuccess && ++idle_checks > 1000) break;  // stream drained and still no word: give up (degree 0)
                }
            }
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == reset_seq) {
                src_row[0] = h_src_box->row[0];
                src_row[1] = h_src_box->row[1];
            }
            src_row_pending = false;
        }
        return src_row[1] - src_row[0];
    }

    hipError_t Extract(VertexId *h_labels, VertexId *h_preds)
    {
        hipError_t retval = hipSuccess;
        DataSlice *ds = data_slices[0];
        hipStream_t stream = this->graph_slices[0]->stream;
        GR_CHECK(EmitLabels(stream), "BFSProblem Extract: EmitLabels failed");  // (no-op after an Enact: it ends with this pass)
        GR_CHECK(hipStreamSynchronize(stream), "BFSProblem Extract sync failed");
        if (this->nodes > 0)
            GR_CHECK(hipMemcpy(h_labels, ds->d_labels, sizeof(VertexId) * static_cast<size_t>(this->nodes), hipMemcpyDeviceToHost),
                     "BFSProblem hipMemcpy d_labels failed");
        if (MARK_PREDECESSORS && h_preds && this->nodes > 0)
            GR_CHECK(hipMemcpy(h_preds, ds->d_preds, sizeof(VertexId) * static_cast<size_t>(this->nodes), hipMemcpyDeviceToHost),
                     "BFSProblem hipMemcpy d_preds failed");
        return retval;
    }

    // ---- deferred labels: the kept level bitmaps of the running search ----
    bool defer_labels = true;       // policy (off: every kernel labels at discovery, Reset fills -1)
    bool emit_current = false;      // state: a speculative emit pass has run and no vertex has been discovered since
    bool labels_deferred = false;   // state: d_labels is incomplete until EmitLabels has run (set by Reset, cleared by EmitLabels)
    int level_mask_limit = kLevelMasks;  // bitmaps of the pool a search may use (tests shrink it to force mid-search flushes)
    LevelMaskList<VertexId> level_masks; // (bitmap, label) of every kept level
    int mask_ring = 0;

    bool MaskKept(int idx) const
    {
        for (int k = 0; k < level_masks.count; ++k)
            if (level_masks.mask[k] == reinterpret_cast<const unsigned long long *>(data_slices[0]->d_frontier_mask[idx])) return true;
        return false;
    }
    // A bitmap of the pool that is neither kept nor one of the (up to two) the caller is still reading.  When every other one
    // is kept, the kept levels are flushed into d_labels first (partial stores -- the price the deferral normally avoids).
    hipError_t AcquireMask(hipStream_t stream, int &idx, int hold_a = -1, int hold_b = -1)
    {
        const int holds[2] = {hold_a, hold_b};
        bool got = false;
        const hipError_t rc = TryAcquireMask(stream, idx, holds, 2, got);
        if (rc) return rc;
        return got ? hipSuccess : util::GRError(hipErrorInvalidValue, "BFSProblem: no free frontier bitmap", __FILE__, __LINE__);
    }
    // the same with any number of bitmaps held; got = false when even a flush of the kept levels frees none
    hipError_t TryAcquireMask(hipStream_t stream, int &idx, const int *holds, int n_holds, bool &got)
    {
        hipError_t retval = hipSuccess;
        got = true;
        if (!data_slices[0]->d_frontier_mask[1]) {  // no pool (no inverse graph): the binned level's single bitmap
            idx = 0;
            return retval;
        }
        const int limit = level_mask_limit < 4 ? 4 : (level_mask_limit > kLevelMasks ? kLevelMasks : level_mask_limit);
        for (int round = 0; round < 2; ++round) {
            for (int t = 0; t < limit; ++t) {
                const int cand = (mask_ring + t) % limit;
                bool held = false;
                for (int h = 0; h < n_holds; ++h) held = held || holds[h] == cand;
                if (held || MaskKept(cand)) continue;
                idx = cand;
                mask_ring = (cand + 1) % limit;
                return retval;
            }
            if (level_masks.count == 0) break;  // nothing to flush: the holds alone fill the pool
            GR_CHECK(FlushLevelMasks(stream), "BFSProblem FlushLevelMasks failed");
        }
        got = false;
        return retval;
    }
    void KeepMask(int idx, VertexId label)
    {
        level_masks.mask[level_masks.count] = reinterpret_cast<const unsigned long long *>(data_slices[0]->d_frontier_mask[idx]);
        level_masks.label[level_masks.count] = label;
        ++level_masks.count;
    }
    template <bool FULL>
    hipError_t LaunchEmit(hipStream_t stream, VertexId src)
    {
        DataSlice *ds = data_slices[0];
        long long grid = ((static_cast<long long>(this->nodes) + 3) / 4 + 255) / 256;  // one quad of vertices per thread
        if (grid > (1 << 20)) grid = 1 << 20;
        if (grid < 1) grid = 1;
        LevelMaskList<VertexId> list = level_masks;
        // (one instantiation per list length up to 8: every extra bitmap is one more load per lane of a pass that is all loads and stores)
        const int kmax = list.count <= 8 ? list.count : kLevelMasks;
        for (int k = list.count; k < kmax; ++k) {  // padding: entry 0 again
            list.mask[k] = list.mask[0];
            list.label[k] = list.label[0];
        }
        const unsigned long long *vis = reinterpret_cast<const unsigned long long *>(ds->d_visited_mask);
        const unsigned long long *nev = reinterpret_cast<const unsigned long long *>(ds->d_never_mask);
        const long long n = static_cast<long long>(this->nodes);
        const dim3 g(static_cast<unsigned>(grid)), b(256);
#define GRX_EMIT_CASE(K) case K: hipLaunchKernelGGL((EmitLabelsKernel<VertexId, FULL, K>), g, b, 0, stream, list, vis, nev, n, src, ds->d_labels); break;
        switch (kmax) {
            GRX_EMIT_CASE(0) GRX_EMIT_CASE(1) GRX_EMIT_CASE(2) GRX_EMIT_CASE(3) GRX_EMIT_CASE(4) GRX_EMIT_CASE(5) GRX_EMIT_CASE(6) GRX_EMIT_CASE(7)
            GRX_EMIT_CASE(8)
            default: hipLaunchKernelGGL((EmitLabelsKernel<VertexId, FULL, kLevelMasks>), g, b, 0, stream, list, vis, nev, n, src, ds->d_labels);
        }
#undef GRX_EMIT_CASE
        level_masks.count = 0;
        return util::GRError(hipGetLastError(), "EmitLabelsKernel launch failed", __FILE__, __LINE__);
    }
    hipError_t FlushLevelMasks(hipStream_t stream)
    {
        if (level_masks.count == 0) return hipSuccess;
        return LaunchEmit<false>(stream, static_cast<VertexId>(-1));
    }
    // The closing pass of a search whose Reset left the labels unfilled.
    hipError_t EmitLabels(hipStream_t stream)
    {
        if (!labels_deferred) return hipSuccess;
        labels_deferred = false;
        if (emit_current) {  // a speculative pass already wrote them and nothing was discovered since
            emit_current = false;
            level_masks.count = 0;
            return hipSuccess;
        }
        return LaunchEmit<true>(stream, (source >= 0 && source < this->nodes) ? source : static_cast<VertexId>(-1));
    }
    // The same pass queued BEFORE the host knows that the search is over (behind the launch that usually ends it): the kept
    // bitmaps and the deferral stay as they are, so that a search that does go on simply ends with another pass -- the first
    // one wrote nothing wrong (labels of levels that existed), the second one covers the rest.  The caller clears emit_current
    // when anything is discovered after this.
    // ... and the same pass queued behind a CHAIN of sweeps and the closing levels that follow it on the device
    // (kernel.hpp ChainedPersistentLevelsKernel): the chain's bitmaps go in as candidates, the gate words say how many were filled
    hipError_t EmitLabelsGated(hipStream_t stream, const int *chain_masks, const VertexId *chain_labels, int chain_count, const int *d_gate)
    {
        if (!labels_deferred || !speculative_emit) return hipSuccess;
        if (level_masks.count + chain_count > kLevelMasks) return hipSuccess;  // (no room in the list: the ordinary pass will do)
        const LevelMaskList<VertexId> keep = level_masks;
        level_masks.chain_first = level_masks.count;
        for (int k = 0; k < chain_count; ++k) KeepMask(chain_masks[k], chain_labels[k]);
        level_masks.d_gate = d_gate;
        const hipError_t rc = LaunchEmit<true>(stream, (source >= 0 && source < this->nodes) ? source : static_cast<VertexId>(-1));
        level_masks = keep;
        return rc;
    }
    hipError_t EmitLabelsSpeculative(hipStream_t stream)
    {
        if (!labels_deferred || !speculative_emit) return hipSuccess;
        const LevelMaskList<VertexId> keep = level_masks;
        const hipError_t rc = LaunchEmit<true>(stream, (source >= 0 && source < this->nodes) ? source : static_cast<VertexId>(-1));
        level_masks = keep;
        emit_current = true;
        return rc;
    }

    // the enactor words Reset arms (set by the enactor at its first Enact on this problem; must outlive the problem's searches)
    util::WorkProgress *arm_progress = nullptr;
    util::WorkProgress *armed_progress = nullptr;  // whose words the last Reset armed (nullptr: nobody's)

    VertexId source = -1;
    SizeT src_row[2] = {0, 0};
    struct SourceBox {
        unsigned long long seq;
        SizeT row[2];
    };
    SourceBox *h_src_box = nullptr;  // pinned + mapped: written by BfsResetKernel
    unsigned long long reset_seq = 0;
    bool src_row_pending = false;
};

}  // namespace bfs
}  // namespace app
}  // namespace gunrock
