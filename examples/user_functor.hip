// examples/user_functor.hip -- a user-written primitive on the operator API, with NO engine-specific hook.
//
// What "drop-in at the Problem / Enactor / Functor level" means: a functor that only has the four methods of the
// reference's functor shape -- CondEdge / ApplyEdge / CondFilter / ApplyFilter with the reference's argument lists
// (gunrock/app/bfs/bfs_functor.cuh:49-117) -- a problem derived from ProblemBase whose DataSlice travels to the kernels,
// and an enactor that alternates advance and filter until the frontier is empty, the loop of the reference's
// codesnaps/bfs/bfs_enactor.cuh:41-62.  None of the optional hooks of oprtr/advance/functor_hooks.hpp (ScreenEdge,
// IssueEdge / ResolveEdge, SourceData, ApplyEdgeWave, ReduceValue) is defined: the operators detect that at compile time.
//
// The functor is the reference's NON-idempotent, no-predecessor BFS rule: an edge claims its destination with
// atomicCAS(labels[d], -1, depth) and only the winner enqueues it (bfs_functor.cuh:56-58: atomicCAS(&d_labels[d_id], -1,
// s_id + 1) -- in that mode the reference's advance passes the current depth in place of s_id,
// edge_map_partitioned/kernel.cuh:401-403; here the depth travels in the DataSlice).
//
// usage: user_functor market <file.mtx> [--src=<vertex>] [--undirected]     prints CORRECT / INCORRECT like the drivers.
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <vector>

#include <gunrock/app/enactor_base.hpp>
#include <gunrock/app/problem_base.hpp>
#include <gunrock/graphio/market.hpp>
#include <gunrock/oprtr/advance/kernel.hpp>
#include <gunrock/oprtr/filter/kernel.hpp>
#include <gunrock/util/test_utils.hpp>

using namespace gunrock;
using namespace gunrock::app;

// ---- user problem: depth labels only ----
struct DepthProblem : ProblemBase<int, int, int, false> {
    static constexpr bool MARK_PREDECESSORS = false;
    static constexpr bool ENABLE_IDEMPOTENCE = false;
    struct DataSlice {
        int *d_labels = nullptr;
        int depth = 0;  // label the current advance hands out
    };
    DataSlice slice;

    ~DepthProblem() override
    {
        if (slice.d_labels) util::GRError(hipFree(slice.d_labels), "DepthProblem hipFree failed", __FILE__, __LINE__);
    }
    hipError_t Init(const Csr<int, int, int> &graph)
    {
        hipError_t retval = ProblemBase::Init(false, graph, 1);
        if (retval) return retval;
        GR_CHECK(hipMalloc(&slice.d_labels, sizeof(int) * static_cast<size_t>(nodes > 0 ? nodes : 1)), "DepthProblem hipMalloc failed");
        return retval;
    }
    // labels = -1, source = 0, queue[0] = {source} (bfs_problem.cuh:272-360)
    hipError_t Reset(int src, FrontierType ft, double queue_sizing, int &src_degree)
    {
        hipError_t retval = ProblemBase::Reset(ft, queue_sizing);
        if (retval) return retval;
        GraphSlice<int, int, int> *gs = graph_slices[0];
        GR_CHECK(hipMemsetAsync(slice.d_labels, 0xFF, sizeof(int) * static_cast<size_t>(nodes), gs->stream), "DepthProblem memset failed");
        int row[2] = {0, 0};
        GR_CHECK(hipMemcpyAsync(row, gs->d_row_offsets + src, sizeof(row), hipMemcpyDeviceToHost, gs->stream), "DepthProblem read failed");
        GR_CHECK(hipStreamSynchronize(gs->stream), "DepthProblem sync failed");
        const int zero = 0;
        GR_CHECK(hipMemcpy(slice.d_labels + src, &zero, sizeof(int), hipMemcpyHostToDevice), "DepthProblem seed failed");
        GR_CHECK(hipMemcpy(gs->frontier_queues[0].v, &src, sizeof(int), hipMemcpyHostToDevice), "DepthProblem seed failed");
        GR_CHECK(hipMemcpy(gs->frontier_queues[0].row_start, &row[0], sizeof(int), hipMemcpyHostToDevice), "DepthProblem seed failed");
        GR_CHECK(hipMemcpy(gs->frontier_queues[0].scan, &zero, sizeof(int), hipMemcpyHostToDevice), "DepthProblem seed failed");
        src_degree = row[1] - row[0];
        return retval;
    }
};

// ---- user functor: the four methods of the reference's functor shape, nothing else ----
struct DepthFunctor {
    typedef DepthProblem::DataSlice DataSlice;
    static __device__ __forceinline__ bool CondEdge(int /*s_id*/, int d_id, DataSlice *problem, int /*e_id*/ = 0, int /*e_id_in*/ = 0)
    {
        return atomicCAS(&problem->d_labels[d_id], -1, problem->depth) == -1;  // has anybody claimed d_id as a child yet?
    }
    static __device__ __forceinline__ void ApplyEdge(int, int, DataSlice *, int = 0, int = 0) {}  // the claim wrote the label
    static __device__ __forceinline__ bool CondFilter(int node, DataSlice *, int /*v*/ = 0, int /*nid*/ = 0) { return node != -1; }
    static __device__ __forceinline__ void ApplyFilter(int, DataSlice *, int = 0, int = 0) {}
};

// ---- user enactor: advance, filter, repeat (codesnaps/bfs/bfs_enactor.cuh:41-62) ----
class DepthEnactor : public EnactorBase {
   public:
    DepthEnactor() : EnactorBase(VERTEX_FRONTIERS, false) {}
    typedef oprtr::advance::KernelPolicy<256, 4, 8, oprtr::advance::LB> AdvancePolicy;
    typedef oprtr::filter::KernelPolicy<256, 4, 8> FilterPolicy;

    hipError_t Enact(DepthProblem *problem, int src_degree, long long &depth)
    {
        hipError_t retval = Setup(0, AdvancePolicy::MIN_BLOCKS, FilterPolicy::MIN_BLOCKS);
        if (retval) return retval;
        GraphSlice<int, int, int> *gs = problem->graph_slices[0];
        hipStream_t stream = gs->stream;
        // a third queue: the advance's raw output, which the filter compacts into the next input frontier
        util::Frontier<int, int> raw = gs->frontier_queues[1], next;
        next.capacity = raw.capacity;
        GR_CHECK(hipMalloc(&next.v, sizeof(int) * raw.capacity), "DepthEnactor hipMalloc failed");
        GR_CHECK(hipMalloc(&next.row_start, sizeof(int) * raw.capacity), "DepthEnactor hipMalloc failed");
        GR_CHECK(hipMalloc(&next.scan, sizeof(int) * raw.capacity), "DepthEnactor hipMalloc failed");
        util::Frontier<int, int> in = gs->frontier_queues[0];
        unsigned queue_length = src_degree > 0 ? 1u : 0u, queue_edges = static_cast<unsigned>(src_degree);
        GR_CHECK(work_progress.ResetWithTail(0, 0, 0, stream), "DepthEnactor arm failed");
        depth = 0;
        while (queue_length > 0) {
            problem->slice.depth = static_cast<int>(depth) + 1;
            oprtr::advance::AdvanceArgs<int, int> a;
            a.in = in;
            a.out = raw;
            a.in_len = static_cast<int>(queue_length);
            a.in_edges = static_cast<int>(queue_edges);
            a.d_row_offsets = gs->d_row_offsets;
            a.d_column_indices = gs->d_column_indices;
            a.d_tail_out = work_progress.d_tail + 1;
            a.d_tail_clear = work_progress.d_tail + 2;
            a.d_overflow = work_progress.d_overflow;
            if ((retval = oprtr::advance::LaunchKernel<AdvancePolicy, DepthProblem, DepthFunctor, /*OUT_WITH_DEGREES*/ false>(
                     a, problem->slice, enactor_stats.advance_grid_size, stream, oprtr::advance::V2V)))
                break;
            unsigned raw_len = 0, unused = 0;
            if ((retval = work_progress.GetTail(1, raw_len, unused, stream))) break;
            oprtr::filter::FilterArgs<int, int> f;
            f.d_in = raw.v;
            f.num_elements = static_cast<int>(raw_len);
            f.out = next;
            f.d_tail_out = work_progress.d_tail + 2;
            f.d_tail_clear = work_progress.d_tail + 1;
            f.d_overflow = work_progress.d_overflow;
            f.d_row_offsets = gs->d_row_offsets;
            if ((retval = oprtr::filter::LaunchKernel<FilterPolicy, DepthProblem, DepthFunctor, /*WITH_DEGREES*/ true>(
                     f, problem->slice, enactor_stats.filter_grid_size, stream)))
                break;
            if (raw_len == 0) {  // (the filter did not run, so nobody cleared slot 1 / wrote slot 2)
                queue_length = 0;
                break;
            }
            if ((retval = work_progress.GetTail(2, queue_length, queue_edges, stream))) break;
            const util::Frontier<int, int> t = in;
            in = next;
            next = t;
            ++depth;
        }
        bool overflow = false;
        if (!retval) retval = work_progress.CheckOverflow(overflow, stream);
        if (!retval && overflow)
            retval = util::GRError(hipErrorInvalidConfiguration, "Frontier queue overflow. Please increase queue-sizing factor.", __FILE__, __LINE__);
        // the problem owns queue 0's arrays; free whichever of the two extra-queue roles ended up on our allocation
        util::Frontier<int, int> mine = (in.v == gs->frontier_queues[0].v) ? next : in;
        hipFree(mine.v);
        hipFree(mine.row_start);
        hipFree(mine.scan);
        return retval;
    }
};

int main(int argc, char **argv)
{
    util::CommandLineArgs args(argc, argv);
    if (args.ParsedArgc() < 2 || args.Positional(0) != "market") {
        std::printf("usage: user_functor market <file.mtx> [--src=<vertex>] [--undirected]\n");
        return 1;
    }
    int src = 0;
    args.GetCmdLineArgument("src", src);
    const bool undirected = args.CheckCmdLineFlag("undirected");
    std::string file = args.Positional(1);
    Csr<int, int, int> csr(false);
    if (graphio::BuildMarketGraph<false>(const_cast<char *>(file.c_str()), csr, undirected, false) != 0) return 1;
    if (src < 0 || src >= csr.nodes) src = 0;
    std::printf("Graph: %d nodes, %d edges, source %d\n", csr.nodes, csr.edges, src);

    DepthProblem problem;
    DepthEnactor enactor;
    int src_degree = 0;
    long long depth = 0;
    if (util::GRError(problem.Init(csr), "Problem Initialization Failed", __FILE__, __LINE__)) return 1;
    if (util::GRError(problem.Reset(src, enactor.GetFrontierType(), 1.0, src_degree), "Problem Data Reset Failed", __FILE__, __LINE__)) return 1;
    if (util::GRError(enactor.Enact(&problem, src_degree, depth), "Problem Enact Failed", __FILE__, __LINE__)) return 1;
    std::vector<int> h_labels(csr.nodes), ref(csr.nodes, -1);
    if (util::GRError(hipMemcpy(h_labels.data(), problem.slice.d_labels, sizeof(int) * static_cast<size_t>(csr.nodes), hipMemcpyDeviceToHost),
                      "Problem Data Extraction Failed", __FILE__, __LINE__))
        return 1;

    ref[src] = 0;  // SimpleReferenceBfs (tests/bfs/test_bfs.cu:264-311)
    std::deque<int> fifo(1, src);
    int ref_depth = 0;
    while (!fifo.empty()) {
        const int u = fifo.front();
        fifo.pop_front();
        for (int e = csr.row_offsets[u]; e < csr.row_offsets[u + 1]; ++e) {
            const int w = csr.column_indices[e];
            if (ref[w] == -1) {
                ref[w] = ref[u] + 1;
                if (ref[w] > ref_depth) ref_depth = ref[w];
                fifo.push_back(w);
            }
        }
    }
    std::printf("search depth: GPU %lld, CPU %d\nLabel Validity: ", depth, ref_depth);
    const int errors = util::CompareResults(h_labels.data(), ref.data(), csr.nodes, true);
    std::printf("\n%s\n", errors == 0 ? "TEST PASSED" : "TEST FAILED");
    return errors == 0 ? 0 : 2;
}
