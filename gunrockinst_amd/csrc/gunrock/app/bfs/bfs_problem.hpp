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
#include <gunrock/util/memset_kernel.hpp>

namespace gunrock {
namespace app {
namespace bfs {

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
        // direction-optimizing traversal (reference app/dobfs: d_frontier_map_in/out, dobfs_problem.cuh):
        unsigned *d_frontier_mask[2] = {nullptr, nullptr};  // 1 bit per vertex: current / next frontier
        const SizeT *d_inv_row_offsets = nullptr;           // in-neighbour CSR (CSC of the graph)
        const VertexId *d_inv_column_indices = nullptr;
    };

    // direction-optimizing switches, names from the reference's DOBFS driver (tests/dobfs/test_dobfs.cu:530-534)
    bool direction_optimizing = false;
    float alpha = 14.0f;  // top-down -> bottom-up when frontier_edges * alpha > unexplored_edges
    float beta = 24.0f;   // bottom-up -> top-down when frontier_vertices * beta < nodes

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
                for (int i = 0; i < 2; ++i)
                    if (ds->d_frontier_mask[i]) util::GRError(hipFree(ds->d_frontier_mask[i]), "BFSProblem hipFree d_frontier_mask failed", __FILE__, __LINE__);
                delete ds;
            }
            delete[] data_slices;
        }
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
        for (int i = 0; i < 2; ++i)
            if (!ds->d_frontier_mask[i])
                GR_CHECK(hipMalloc(&ds->d_frontier_mask[i], sizeof(unsigned) * static_cast<size_t>(MaskWords() + 2)),
                         "BFSProblem hipMalloc d_frontier_mask failed");
        if (alpha_ > 0) alpha = alpha_;
        if (beta_ > 0) beta = beta_;
        direction_optimizing = true;
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

    // Source degree / row start are read back from HBM (8 bytes) so Reset works for device-resident graphs.
    hipError_t Reset(VertexId src, FrontierType frontier_type, double queue_sizing)
    {
        hipError_t retval = hipSuccess;
        if ((retval = Base::Reset(frontier_type, queue_sizing))) return retval;
        GraphSlice<VertexId, SizeT, Value> *gs = this->graph_slices[0];
        DataSlice *ds = data_slices[0];
        hipStream_t stream = gs->stream;
        util::Memset(ds->d_labels, static_cast<VertexId>(-1), this->nodes, stream);
        if (MARK_PREDECESSORS) util::Memset(ds->d_preds, static_cast<VertexId>(-2), this->nodes, stream);
        util::Memset(ds->d_visited_mask, 0u, MaskWords() + 2, stream);
        ds->iteration = 0;
        src_row[0] = src_row[1] = 0;
        if (src >= 0 && src < this->nodes) {
            GR_CHECK(hipMemcpyAsync(src_row, gs->d_row_offsets + src, 2 * sizeof(SizeT), hipMemcpyDeviceToHost, stream),
                     "BFSProblem read source row failed");
            const VertexId zero = 0, minus_one = -1;
            const unsigned bit = 1u << (src & 31);
            GR_CHECK(hipMemcpyAsync(ds->d_labels + src, &zero, sizeof(VertexId), hipMemcpyHostToDevice, stream),
                     "BFSProblem seed label failed");
            if (MARK_PREDECESSORS)
                GR_CHECK(hipMemcpyAsync(ds->d_preds + src, &minus_one, sizeof(VertexId), hipMemcpyHostToDevice, stream),
                         "BFSProblem seed pred failed");
            GR_CHECK(hipMemcpyAsync(ds->d_visited_mask + (src >> 5), &bit, sizeof(unsigned), hipMemcpyHostToDevice, stream),
                     "BFSProblem seed mask failed");
            const SizeT zero_prefix = 0;
            GR_CHECK(hipMemcpyAsync(gs->frontier_queues[0].v, &src, sizeof(VertexId), hipMemcpyHostToDevice, stream),
                     "BFSProblem seed queue failed");
            GR_CHECK(hipMemcpyAsync(gs->frontier_queues[0].scan, &zero_prefix, sizeof(SizeT), hipMemcpyHostToDevice, stream),
                     "BFSProblem seed queue failed");
            GR_CHECK(hipStreamSynchronize(stream), "BFSProblem Reset sync failed");
            GR_CHECK(hipMemcpyAsync(gs->frontier_queues[0].row_start, &src_row[0], sizeof(SizeT), hipMemcpyHostToDevice, stream),
                     "BFSProblem seed queue failed");
        }
        GR_CHECK(hipStreamSynchronize(stream), "BFSProblem Reset sync failed");
        source = src;
        return retval;
    }

    // Degree of the source, known after Reset (seeds the first packed tail).
    SizeT SourceDegree() const { return src_row[1] - src_row[0]; }

    hipError_t Extract(VertexId *h_labels, VertexId *h_preds)
    {
        hipError_t retval = hipSuccess;
        DataSlice *ds = data_slices[0];
        hipStream_t stream = this->graph_slices[0]->stream;
        GR_CHECK(hipStreamSynchronize(stream), "BFSProblem Extract sync failed");
        if (this->nodes > 0)
            GR_CHECK(hipMemcpy(h_labels, ds->d_labels, sizeof(VertexId) * static_cast<size_t>(this->nodes), hipMemcpyDeviceToHost),
                     "BFSProblem hipMemcpy d_labels failed");
        if (MARK_PREDECESSORS && h_preds && this->nodes > 0)
            GR_CHECK(hipMemcpy(h_preds, ds->d_preds, sizeof(VertexId) * static_cast<size_t>(this->nodes), hipMemcpyDeviceToHost),
                     "BFSProblem hipMemcpy d_preds failed");
        return retval;
    }

    VertexId source = -1;
    SizeT src_row[2] = {0, 0};
};

}  // namespace bfs
}  // namespace app
}  // namespace gunrock
