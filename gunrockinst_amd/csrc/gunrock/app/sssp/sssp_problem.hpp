// app/sssp/sssp_problem.hpp -- device data for single-source shortest paths.
//
// Same contract as the reference's SSSPProblem (gunrock/app/sssp/sssp_problem.cuh:35-387):
//   DataSlice { d_labels (unsigned distance), d_weights, d_preds, d_visit_lookup, d_delta }     (:51-59)
//   Init(stream_from_host, graph, num_gpus, delta_factor = 16)                                  (:185-288)
//   Reset(src, frontier_type, queue_sizing): labels = UINT_MAX, preds = iota, lookup = -1; src 0  (:299-377)
//   Extract(h_labels, h_preds)                                                                   (:144-177 pattern)
//   delta = average_edge_value * 32 / average_degree * delta_factor                              (:273, :379-383)
// Differences:
//   * the averages are actually computed here (the reference driver never calls GetAverageEdgeValue, so its
//     delta is 0 or NaN and every bucket has width 1 -- SURVEY 8(a) S1; distances do not depend on delta);
//   * with MARK_PATHS the distance and the predecessor of a vertex live in ONE 64-bit word
//     (distance << 32 | predecessor) updated by a single atomicMin, so the predecessor always belongs to the
//     stored distance.  The reference writes d_preds with a plain store after its 32-bit atomicMin
//     (sssp_functor.cuh:52-84): two racing relaxations can leave a predecessor that does not match the final
//     distance.
#pragma once

#include <hip/hip_runtime.h>
#include <climits>

#include <gunrock/app/problem_base.hpp>
#include <gunrock/graphio/device_sort.hpp>
#include <gunrock/util/memset_kernel.hpp>

namespace gunrock {
namespace app {
namespace sssp {

template <typename VertexId>
__global__ void InitDistPredKernel(unsigned long long *d_dist_pred, long long nodes)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < nodes; i += stride)
        d_dist_pred[i] = (0xFFFFFFFFull << 32) | static_cast<unsigned>(i);  // unreached, pred = own id (iota init)
}

static __global__ void SplitDistPredKernel(const unsigned long long *d_dist_pred, long long nodes, unsigned *d_labels,
                                           int *d_preds)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < nodes; i += stride) {
        const unsigned long long x = d_dist_pred[i];
        d_labels[i] = static_cast<unsigned>(x >> 32);
        d_preds[i] = static_cast<int>(static_cast<unsigned>(x));
    }
}

// frontier of every vertex that has in-edges, in vertex order: vertices without in-edges own no slots, so the exclusive degree
// prefix of entry v IS inv_row_offsets[v]
template <typename SizeT>
__global__ void HasInEdgesKernel(const SizeT *d_inv_row_offsets, long long nodes, unsigned *d_flags)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long v = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; v <= nodes; v += stride)
        d_flags[v] = (v < nodes && d_inv_row_offsets[v + 1] > d_inv_row_offsets[v]) ? 1u : 0u;
}
template <typename VertexId, typename SizeT>
__global__ void InFrontierKernel(const SizeT *d_inv_row_offsets, const unsigned *d_flags, const unsigned *d_pos, long long nodes,
                                 VertexId *d_v, SizeT *d_row_start, SizeT *d_scan)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long v = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; v < nodes; v += stride) {
        if (!d_flags[v]) continue;
        const unsigned i = d_pos[v];
        d_v[i] = static_cast<VertexId>(v);
        d_row_start[i] = d_inv_row_offsets[v];
        d_scan[i] = d_inv_row_offsets[v];
    }
}

template <typename _VertexId, typename _SizeT, typename _Value, bool _MARK_PATHS>
struct SSSPProblem : ProblemBase<_VertexId, _SizeT, _Value, false> {
    typedef ProblemBase<_VertexId, _SizeT, _Value, false> Base;
    typedef _VertexId VertexId;
    typedef _SizeT SizeT;
    typedef _Value Value;
    static constexpr bool MARK_PATHS = _MARK_PATHS;
    static constexpr bool MARK_PREDECESSORS = _MARK_PATHS;
    static constexpr bool ENABLE_IDEMPOTENCE = false;

    struct DataSlice {
        unsigned *d_labels = nullptr;               // distance (the only copy when !MARK_PATHS; extract target otherwise)
        unsigned long long *d_dist_pred = nullptr;  // MARK_PATHS: distance << 32 | predecessor
        const unsigned *d_weights = nullptr;        // per edge
        VertexId *d_preds = nullptr;                // MARK_PATHS: extract target
        int *d_visit_lookup = nullptr;              // de-duplication tag per vertex
        float delta = 0.0f;                         // bucket width (0 = one bucket per distance value)
        // pull iterations (dense levels): weights of the IN-edges in inverse-CSR order, and per vertex the best candidate the
        // reducing advance found -- a distance, or with MARK_PATHS the packed (distance << 32 | in-neighbour)
        const unsigned *d_inv_weights = nullptr;
        void *d_pull = nullptr;

        __device__ __forceinline__ unsigned Distance(VertexId v) const
        {
            if (MARK_PATHS) return reinterpret_cast<const unsigned *>(d_dist_pred)[2 * static_cast<size_t>(v) + 1];  // high half
            return d_labels[v];
        }
    };

    DataSlice **data_slices = nullptr;
    int delta_factor = 16;
    bool owns_weights = false;
    unsigned *d_weights_owned = nullptr;

    // far pile: vertices whose bucket lies beyond the current level, with the distance they had when parked
    VertexId *d_far_v[2] = {nullptr, nullptr};
    unsigned *d_far_d[2] = {nullptr, nullptr};
    VertexId *d_candidates = nullptr;  // advance output (ids, duplicates allowed)
    SizeT far_capacity = 0;
    SizeT candidate_capacity = 0;
    SizeT src_row[2] = {0, 0};
    // in-neighbour lists with their weights (SetInverseGraph / BuildInverse): enable pull iterations
    const SizeT *d_inv_row_offsets = nullptr;
    const VertexId *d_inv_column_indices = nullptr;
    SizeT *d_own_inv_row_offsets = nullptr;
    VertexId *d_own_inv_column_indices = nullptr;
    unsigned *d_own_inv_weights = nullptr;
    util::Frontier<VertexId, SizeT> inv_frontier;  // every vertex that has in-edges, with the degree prefix of the in-lists
    SizeT inv_frontier_len = 0;
    SizeT inv_frontier_edges = 0;
    // a level relaxes by PULL when its frontier has more than this many out-edges (-1: 3/4 of the edges; 0: never).  A pull
    // sweeps every in-edge once, without atomics; a push pays a memory-side atomicMin (~27 G/s) per edge that looks improvable.
    // Measured at R-MAT scale-22 (66 M edges, 16 MiB of distances -- four times an XCD's L2): a pull level costs 0.9 ms whatever
    // the frontier, because every in-edge gathers its neighbour's distance as a 64-byte sector from beyond L2 (4.2 GB per
    // level); a push level of 32 M frontier edges costs 0.64 ms.  So pulling pays only when the frontier holds nearly all edges.
    long long pull_min_edges = -1;
    bool HasInverse() const { return d_inv_row_offsets != nullptr && inv_frontier.v != nullptr; }
    long long PullMinEdges() const { return pull_min_edges >= 0 ? pull_min_edges : static_cast<long long>(this->edges) / 4 * 3 + 1; }

    ~SSSPProblem() override
    {
        if (data_slices) {
            DataSlice *ds = data_slices[0];
            if (ds) {
                if (ds->d_labels) hipFree(ds->d_labels);
                if (ds->d_dist_pred) hipFree(ds->d_dist_pred);
                if (ds->d_preds) hipFree(ds->d_preds);
                if (ds->d_visit_lookup) hipFree(ds->d_visit_lookup);
                if (ds->d_pull) hipFree(ds->d_pull);
                delete ds;
                data_slices[0] = nullptr;
            }
            delete[] data_slices;
        }
        for (int i = 0; i < 2; ++i) {
            if (d_far_v[i]) hipFree(d_far_v[i]);
            if (d_far_d[i]) hipFree(d_far_d[i]);
        }
        if (d_candidates) hipFree(d_candidates);
        if (d_weights_owned) hipFree(d_weights_owned);
        if (d_own_inv_row_offsets) hipFree(d_own_inv_row_offsets);
        if (d_own_inv_column_indices) hipFree(d_own_inv_column_indices);
        if (d_own_inv_weights) hipFree(d_own_inv_weights);
        if (inv_frontier.v) hipFree(inv_frontier.v);
        if (inv_frontier.row_start) hipFree(inv_frontier.row_start);
        if (inv_frontier.scan) hipFree(inv_frontier.scan);
    }

    hipError_t AllocData()
    {
        hipError_t retval = hipSuccess;
        data_slices = new DataSlice *[1];
        data_slices[0] = new DataSlice();
        DataSlice *ds = data_slices[0];
        const size_t n = static_cast<size_t>(this->nodes > 0 ? this->nodes : 1);
        GR_CHECK(hipMalloc(&ds->d_labels, sizeof(unsigned) * n), "SSSPProblem hipMalloc d_labels failed");
        if (MARK_PATHS) {
            GR_CHECK(hipMalloc(&ds->d_dist_pred, sizeof(unsigned long long) * n), "SSSPProblem hipMalloc d_dist_pred failed");
            GR_CHECK(hipMalloc(&ds->d_preds, sizeof(VertexId) * n), "SSSPProblem hipMalloc d_preds failed");
        }
        GR_CHECK(hipMalloc(&ds->d_visit_lookup, sizeof(int) * n), "SSSPProblem hipMalloc d_visit_lookup failed");
        return retval;
    }

    // reference sssp_problem.cuh:379-383 with the averages actually evaluated
    static float EstimatedDelta(double average_edge_value, double average_degree)
    {
        if (average_degree <= 0) return static_cast<float>(average_edge_value);
        return static_cast<float>(average_edge_value * 32 / average_degree);
    }

    hipError_t Init(bool stream_from_host, Csr<VertexId, Value, SizeT> &graph, int num_gpus = 1, int delta_factor_ = 16)
    {
        hipError_t retval = hipSuccess;
        if ((retval = Base::Init(stream_from_host, graph, num_gpus, true))) return retval;
        if ((retval = AllocData())) return retval;
        delta_factor = delta_factor_ > 0 ? delta_factor_ : 16;
        data_slices[0]->d_weights = reinterpret_cast<const unsigned *>(this->graph_slices[0]->d_edge_values);
        // averages over the unsigned weights (Csr<int,..> would average negative ints for weights >= 2^31)
        double mean_w = 0, cnt = 0;
        if (graph.edge_values)
            for (SizeT e = 0; e < graph.edges; ++e) {
                cnt += 1;
                mean_w += (static_cast<double>(static_cast<unsigned>(graph.edge_values[e])) - mean_w) / cnt;
            }
        data_slices[0]->delta = EstimatedDelta(static_cast<double>(static_cast<long long>(mean_w)),
                                               static_cast<double>(graph.GetAverageDegree())) * delta_factor;
        return retval;
    }

    hipError_t InitFromDevice(SizeT nodes, SizeT edges, SizeT *d_row_offsets, VertexId *d_column_indices,
                              const unsigned *d_weights, float delta)
    {
        hipError_t retval = hipSuccess;
        if ((retval = Base::InitFromDevice(nodes, edges, d_row_offsets, d_column_indices))) return retval;
        if ((retval = AllocData())) return retval;
        data_slices[0]->d_weights = d_weights;
        data_slices[0]->delta = delta;
        return retval;
    }

    hipError_t Reset(VertexId src, FrontierType frontier_type, double queue_sizing)
    {
        hipError_t retval = hipSuccess;
        if ((retval = Base::Reset(frontier_type, queue_sizing))) return retval;
        GraphSlice<VertexId, SizeT, Value> *gs = this->graph_slices[0];
        DataSlice *ds = data_slices[0];
        hipStream_t stream = gs->stream;
        // work queues: candidates and far piles hold one entry per successful relaxation (sssp_enactor.cuh:235-238
        // sizes the piles by the edge count)
        double want = (static_cast<double>(this->edges > this->nodes ? this->edges : this->nodes) + 1024) *
                      (queue_sizing > 0 ? queue_sizing : 1.0);
        if (want > 2147483000.0) want = 2147483000.0;
        const SizeT cap = static_cast<SizeT>(want);
        if (cap > far_capacity) {
            for (int i = 0; i < 2; ++i) {
                if (d_far_v[i]) hipFree(d_far_v[i]);
                if (d_far_d[i]) hipFree(d_far_d[i]);
                GR_CHECK(hipMalloc(&d_far_v[i], sizeof(VertexId) * static_cast<size_t>(cap)), "SSSPProblem hipMalloc far pile failed");
                GR_CHECK(hipMalloc(&d_far_d[i], sizeof(unsigned) * static_cast<size_t>(cap)), "SSSPProblem hipMalloc far pile failed");
            }
            if (d_candidates) hipFree(d_candidates);
            GR_CHECK(hipMalloc(&d_candidates, sizeof(VertexId) * static_cast<size_t>(cap)), "SSSPProblem hipMalloc candidates failed");
            far_capacity = cap;
            candidate_capacity = cap;
        }
        if (MARK_PATHS) {
            hipLaunchKernelGGL((InitDistPredKernel<VertexId>), dim3(util::MemsetGrid(this->nodes * 2LL)), dim3(256), 0, stream,
                               ds->d_dist_pred, static_cast<long long>(this->nodes));
        } else {
            util::Memset(ds->d_labels, 0xFFFFFFFFu, this->nodes, stream);
        }
        util::Memset(ds->d_visit_lookup, -1, this->nodes, stream);
        src_row[0] = src_row[1] = 0;
        if (src >= 0 && src < this->nodes) {
            GR_CHECK(hipMemcpyAsync(src_row, gs->d_row_offsets + src, 2 * sizeof(SizeT), hipMemcpyDeviceToHost, stream),
                     "SSSPProblem read source row failed");
            const unsigned zero = 0;
            const unsigned long long zero_self = static_cast<unsigned>(src);  // distance 0, pred = src (iota convention)
            if (MARK_PATHS)
                GR_CHECK(hipMemcpyAsync(ds->d_dist_pred + src, &zero_self, sizeof(zero_self), hipMemcpyHostToDevice, stream),
                         "SSSPProblem seed failed");
            else
                GR_CHECK(hipMemcpyAsync(ds->d_labels + src, &zero, sizeof(zero), hipMemcpyHostToDevice, stream),
                         "SSSPProblem seed failed");
            const SizeT zero_prefix = 0;
            GR_CHECK(hipMemcpyAsync(gs->frontier_queues[0].v, &src, sizeof(VertexId), hipMemcpyHostToDevice, stream),
                     "SSSPProblem seed queue failed");
            GR_CHECK(hipMemcpyAsync(gs->frontier_queues[0].scan, &zero_prefix, sizeof(SizeT), hipMemcpyHostToDevice, stream),
                     "SSSPProblem seed queue failed");
            GR_CHECK(hipStreamSynchronize(stream), "SSSPProblem Reset sync failed");
            GR_CHECK(hipMemcpyAsync(gs->frontier_queues[0].row_start, &src_row[0], sizeof(SizeT), hipMemcpyHostToDevice, stream),
                     "SSSPProblem seed queue failed");
        }
        GR_CHECK(hipStreamSynchronize(stream), "SSSPProblem Reset sync failed");
        return retval;
    }

    SizeT SourceDegree() const { return src_row[1] - src_row[0]; }

    // In-neighbour lists with the weight of every in-edge, in HBM (borrowed).  Enables pull iterations.
    hipError_t SetInverseGraph(const SizeT *d_iro, const VertexId *d_ici, const unsigned *d_iw)
    {
        hipError_t retval = hipSuccess;
        GraphSlice<VertexId, SizeT, Value> *gs = this->graph_slices[0];
        DataSlice *ds = data_slices[0];
        hipStream_t stream = gs->stream;
        const long long n = this->nodes;
        d_inv_row_offsets = d_iro;
        d_inv_column_indices = d_ici;
        ds->d_inv_weights = d_iw;
        if (!inv_frontier.v) {
            const size_t cap = static_cast<size_t>(n > 0 ? n : 1) + 1;
            GR_CHECK(hipMalloc(&inv_frontier.v, sizeof(VertexId) * cap), "SSSPProblem hipMalloc failed");
            GR_CHECK(hipMalloc(&inv_frontier.row_start, sizeof(SizeT) * cap), "SSSPProblem hipMalloc failed");
            GR_CHECK(hipMalloc(&inv_frontier.scan, sizeof(SizeT) * cap), "SSSPProblem hipMalloc failed");
            inv_frontier.capacity = static_cast<SizeT>(cap);
            GR_CHECK(hipMalloc(&ds->d_pull, (MARK_PATHS ? sizeof(unsigned long long) : sizeof(unsigned)) * cap), "SSSPProblem hipMalloc failed");
        }
        inv_frontier_len = 0;
        inv_frontier_edges = 0;
        if (n <= 0) return retval;
        unsigned *d_flags = nullptr, *d_pos = nullptr;
        unsigned long long *d_sums = nullptr;
        GR_CHECK(hipMalloc(&d_flags, sizeof(unsigned) * static_cast<size_t>(n + 1)), "SSSPProblem hipMalloc failed");
        GR_CHECK(hipMalloc(&d_pos, sizeof(unsigned) * static_cast<size_t>(n + 1)), "SSSPProblem hipMalloc failed");
        GR_CHECK(hipMalloc(&d_sums, sizeof(unsigned long long) * static_cast<size_t>(graphio::ScanScratchWords(n + 1))), "SSSPProblem hipMalloc failed");
        hipLaunchKernelGGL((HasInEdgesKernel<SizeT>), dim3(1024), dim3(256), 0, stream, d_iro, n, d_flags);
        GR_CHECK(hipGetLastError(), "HasInEdgesKernel launch failed");
        GR_CHECK(graphio::DeviceExclusiveScan<unsigned>(d_flags, d_pos, n + 1, d_sums, stream), "SSSPProblem scan failed");
        hipLaunchKernelGGL((InFrontierKernel<VertexId, SizeT>), dim3(1024), dim3(256), 0, stream, d_iro, d_flags, d_pos, n, inv_frontier.v,
                           inv_frontier.row_start, inv_frontier.scan);
        GR_CHECK(hipGetLastError(), "InFrontierKernel launch failed");
        unsigned total = 0;
        SizeT m_inv = 0;
        GR_CHECK(hipMemcpyAsync(&total, d_pos + n, sizeof(unsigned), hipMemcpyDeviceToHost, stream), "SSSPProblem read failed");
        GR_CHECK(hipMemcpyAsync(&m_inv, d_iro + n, sizeof(SizeT), hipMemcpyDeviceToHost, stream), "SSSPProblem read failed");
        GR_CHECK(hipStreamSynchronize(stream), "SSSPProblem sync failed");
        inv_frontier_len = static_cast<SizeT>(total);
        inv_frontier_edges = m_inv;
        GR_CHECK(hipFree(d_flags), "SSSPProblem hipFree failed");
        GR_CHECK(hipFree(d_pos), "SSSPProblem hipFree failed");
        GR_CHECK(hipFree(d_sums), "SSSPProblem hipFree failed");
        return retval;
    }
    // ... or built here: the transpose of the CSR, the weights travelling with their edges
    hipError_t BuildInverse()
    {
        hipError_t retval = hipSuccess;
        GraphSlice<VertexId, SizeT, Value> *gs = this->graph_slices[0];
        if (!d_own_inv_row_offsets) {
            const size_t m = static_cast<size_t>(this->edges > 0 ? this->edges : 1);
            GR_CHECK(hipMalloc(&d_own_inv_row_offsets, sizeof(SizeT) * (static_cast<size_t>(this->nodes) + 2)), "SSSPProblem hipMalloc failed");
            GR_CHECK(hipMalloc(&d_own_inv_column_indices, sizeof(VertexId) * m), "SSSPProblem hipMalloc failed");
            GR_CHECK(hipMalloc(&d_own_inv_weights, sizeof(unsigned) * m), "SSSPProblem hipMalloc failed");
        }
        GR_CHECK(graphio::DeviceTransposeCsr(this->nodes, this->edges, gs->d_row_offsets, gs->d_column_indices, d_own_inv_row_offsets,
                                             d_own_inv_column_indices, gs->stream, data_slices[0]->d_weights, d_own_inv_weights),
                 "SSSPProblem transpose failed");
        return SetInverseGraph(d_own_inv_row_offsets, d_own_inv_column_indices, d_own_inv_weights);
    }

    hipError_t Extract(unsigned *h_labels, VertexId *h_preds)
    {
        hipError_t retval = hipSuccess;
        DataSlice *ds = data_slices[0];
        hipStream_t stream = this->graph_slices[0]->stream;
        if (this->nodes <= 0) return retval;
        if (MARK_PATHS) {
            hipLaunchKernelGGL(SplitDistPredKernel, dim3(util::MemsetGrid(this->nodes * 2LL)), dim3(256), 0, stream, ds->d_dist_pred,
                               static_cast<long long>(this->nodes), ds->d_labels, reinterpret_cast<int *>(ds->d_preds));
        }
        GR_CHECK(hipStreamSynchronize(stream), "SSSPProblem Extract sync failed");
        GR_CHECK(hipMemcpy(h_labels, ds->d_labels, sizeof(unsigned) * static_cast<size_t>(this->nodes), hipMemcpyDeviceToHost),
                 "SSSPProblem hipMemcpy d_labels failed");
        if (MARK_PATHS && h_preds)
            GR_CHECK(hipMemcpy(h_preds, ds->d_preds, sizeof(VertexId) * static_cast<size_t>(this->nodes), hipMemcpyDeviceToHost),
                     "SSSPProblem hipMemcpy d_preds failed");
        return retval;
    }
};

}  // namespace sssp
}  // namespace app
}  // namespace gunrock
