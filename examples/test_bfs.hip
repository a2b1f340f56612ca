// examples/test_bfs.hip -- command-line BFS driver on the C++ Problem / Enactor API.
//
// Same command line and console keys as the reference driver (tests/bfs/test_bfs.cu:58-89 usage, :343-507 RunTests,
// :174-235 DisplayStats): graph types `market <file>` and `rmat`; flags --undirected --src=<n|randomize|largestdegree>
// --idempotence=<0|1> --mark-pred --traversal-mode=<0|1|2> --queue-sizing= --quick=<0|1> --iteration-num= --v
// --instrumented --device=.  Prints "Label Validity: ... CORRECT", "elapsed: ... ms, rate: ... MiEdges/s".
// Differences: --traversal-mode=2 selects direction-optimizing traversal (needs --undirected); the iteration count is
// honoured (the fork hard-codes 100000, test_bfs.cu:406); predecessors ARE validated (valid-parent check).
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <string>

#include <gunrock/app/bfs/bfs_enactor.hpp>
#include <gunrock/app/bfs/bfs_problem.hpp>
#include <gunrock/graphio/market.hpp>
#include <gunrock/graphio/rmat.hpp>
#include <gunrock/util/test_utils.hpp>

using namespace gunrock;
using namespace gunrock::app;
using namespace gunrock::app::bfs;

static void Usage()
{
    std::printf(
        " test_bfs <graph type> <graph type args> [--device=<device_index>]\n"
        " [--undirected] [--src=<source_index>] [--idempotence=<0|1>] [--v]\n"
        " [--instrumented] [--iteration-num=<num>] [--traversal-mode=<0|1|2>]\n"
        " [--quick=<0|1>] [--mark-pred] [--queue-sizing=<scale factor>]\n"
        "Graph types and args:\n"
        "  market <file>   Matrix-Market coordinate file\n"
        "  rmat            2^10 vertices / 2^10 edges R-MAT (a=.55 b=.2 c=.2 d=.05) on the libc rand() stream\n"
        "  --src=<id|randomize|largestdegree>   [Default: 0]\n"
        "  --traversal-mode=<0|1|2>  0 load-balanced top-down, 1 reserved (TWC), 2 direction-optimizing (needs --undirected)\n"
        "                            [Default: 0 when the average degree exceeds 8, else 1]\n");
}

// serial FIFO breadth-first search on the host; depth labels, -1 = unreached (role of SimpleReferenceBfs, test_bfs.cu:258-322)
static int ReferenceBfs(const Csr<int, int, int> &g, int *labels, int *preds, int src)
{
    for (int i = 0; i < g.nodes; ++i) { labels[i] = -1; if (preds) preds[i] = -1; }
    labels[src] = 0;
    int depth = 0;
    std::deque<int> fifo(1, src);
    util::CpuTimer timer;
    timer.Start();
    while (!fifo.empty()) {
        const int u = fifo.front();
        fifo.pop_front();
        for (int e = g.row_offsets[u]; e < g.row_offsets[u + 1]; ++e) {
            const int w = g.column_indices[e];
            if (labels[w] != -1) continue;
            labels[w] = labels[u] + 1;
            if (preds) preds[w] = u;
            if (labels[w] > depth) depth = labels[w];
            fifo.push_back(w);
        }
    }
    timer.Stop();
    std::printf("CPU BFS finished in %lf msec. cpu_search_depth: %d\n", timer.ElapsedMillis(), depth + 1);
    return depth + 1;
}

static long long CheckParents(const Csr<int, int, int> &g, int src, const int *labels, const int *preds)
{
    long long bad = 0;
    for (int v = 0; v < g.nodes; ++v) {
        if (v == src) { bad += preds[v] != -1; continue; }
        if (labels[v] < 0) { bad += preds[v] != -2; continue; }
        const int p = preds[v];
        bool ok = p >= 0 && p < g.nodes && labels[p] == labels[v] - 1;
        if (ok) {
            ok = false;
            for (int e = g.row_offsets[p]; e < g.row_offsets[p + 1] && !ok; ++e) ok = g.column_indices[e] == v;
        }
        bad += !ok;
    }
    return bad;
}

template <bool INSTRUMENT, bool MARK_PRED, bool IDEMP>
static int RunTests(Csr<int, int, int> &graph, int src, int traversal_mode, double queue_sizing, int iterations, bool quick,
                    bool verbose, bool undirected, int device)
{
    typedef BFSProblem<int, int, int, MARK_PRED, IDEMP, (MARK_PRED && IDEMP)> Problem;
    util::DeviceContext context(device);
    Problem problem;
    BFSEnactor<INSTRUMENT> enactor(verbose);
    std::vector<int> h_labels(graph.nodes), h_preds(MARK_PRED ? graph.nodes : 0), ref_labels(quick ? 0 : graph.nodes);
    if (util::GRError(problem.Init(false, graph, 1), "Problem BFS Initialization Failed", __FILE__, __LINE__)) return 1;
    if (traversal_mode == 2) {
        if (!undirected) {
            std::fprintf(stderr, "--traversal-mode=2 needs --undirected (the CSR must be its own inverse)\n");
            return 1;
        }
        if (util::GRError(problem.InverseIsSelf(), "BFS SetInverseGraph Failed", __FILE__, __LINE__)) return 1;
    }
    double elapsed = 0;
    for (int it = 0; it < iterations; ++it) {
        if (util::GRError(problem.Reset(src, enactor.GetFrontierType(), queue_sizing), "BFS Problem Data Reset Failed", __FILE__, __LINE__)) return 1;
        util::GpuTimer timer;
        hipStream_t stream = problem.graph_slices[0]->stream;
        timer.Start(stream);
        if (util::GRError(enactor.template Enact<Problem>(context, &problem, src, 0, traversal_mode), "BFS Problem Enact Failed", __FILE__, __LINE__)) return 1;
        timer.Stop(stream);
        elapsed += timer.ElapsedMillis();
    }
    elapsed /= iterations;
    long long total_queued = 0, search_depth = 0;
    double avg_duty = 0;
    enactor.GetStatistics(total_queued, search_depth, avg_duty);
    if (util::GRError(problem.Extract(h_labels.data(), MARK_PRED ? h_preds.data() : nullptr), "BFS Problem Data Extraction Failed", __FILE__, __LINE__)) return 1;

    int errors = 0;
    if (!quick) {
        std::printf("Computing reference value ...\n");
        ReferenceBfs(graph, ref_labels.data(), nullptr, src);
        std::printf("\nLabel Validity: ");
        errors = util::CompareResults(h_labels.data(), ref_labels.data(), graph.nodes, true);
        if (errors > 0) std::printf("%d errors occurred.", errors);
        std::printf("\n");
        if (MARK_PRED) {
            const long long bad = CheckParents(graph, src, h_labels.data(), h_preds.data());
            std::printf("Predecessor Validity: %s", bad == 0 ? "CORRECT\n" : "INCORRECT\n");
            errors += bad != 0;
        }
    }
    if (verbose || graph.nodes <= 40) {
        std::printf("[");
        for (int i = 0; i < (graph.nodes < 40 ? graph.nodes : 40); ++i) {
            std::printf("%d:%d", i, h_labels[i]);
            if (MARK_PRED) std::printf(",%d", h_preds[i]);
            std::printf(" ");
        }
        std::printf("]\n");
    }
    // DisplayStats (test_bfs.cu:174-235)
    long long nodes_visited = 0, edges_visited = 0;
    for (int v = 0; v < graph.nodes; ++v)
        if (h_labels[v] > -1) { ++nodes_visited; edges_visited += graph.row_offsets[v + 1] - graph.row_offsets[v]; }
    std::printf("[BFS] finished. ");
    if (nodes_visited < 5) std::printf("Fewer than 5 vertices visited.\n");
    else {
        std::printf("\n elapsed: %.4f ms, rate: %.4f MiEdges/s", elapsed, (double)edges_visited / (elapsed * 1000.0));
        if (search_depth != 0) std::printf(", search_depth: %lld", search_depth);
        std::printf("\n src: %d, nodes_visited: %lld, edges_visited: %lld", src, nodes_visited, edges_visited);
        if (total_queued > 0) std::printf(", total queued: %lld", total_queued);
        std::printf("\n");
    }
    return errors;
}

int main(int argc, char **argv)
{
    util::CommandLineArgs args(argc, argv);
    if (argc < 2 || args.CheckCmdLineFlag("help") || args.ParsedArgc() < 1) { Usage(); return 1; }
    int device = 0;
    args.GetCmdLineArgument("device", device);
    const bool undirected = args.CheckCmdLineFlag("undirected");
    const bool mark_pred = args.CheckCmdLineFlag("mark-pred");
    const bool instrumented = args.CheckCmdLineFlag("instrumented");
    const bool verbose = args.CheckCmdLineFlag("v");
    int idempotence = 1, quick = 1, iterations = 1, traversal_mode = -1;
    double queue_sizing = 1.0;
    args.GetCmdLineArgument("idempotence", idempotence);
    args.GetCmdLineArgument("quick", quick);
    args.GetCmdLineArgument("iteration-num", iterations);
    args.GetCmdLineArgument("traversal-mode", traversal_mode);
    args.GetCmdLineArgument("queue-sizing", queue_sizing);
    if (iterations < 1) iterations = 1;
    if (hipSetDevice(device) != hipSuccess) { std::fprintf(stderr, "cannot select device %d\n", device); return 1; }

    Csr<int, int, int> csr(false);
    const std::string type = args.Positional(0);
    if (type == "market") {
        if (args.ParsedArgc() < 2) { Usage(); return 1; }
        std::string file = args.Positional(1);
        if (graphio::BuildMarketGraph<false>(const_cast<char *>(file.c_str()), csr, undirected, false, false) != 0) return 1;
    } else if (type == "rmat") {
        if (graphio::BuildRmatGraph<false>(1 << 10, 1 << 10, csr, undirected, 0.55, 0.20, 0.20, 0.05) != 0) return 1;  // test_bfs.cu:765-770
    } else {
        std::fprintf(stderr, "Unspecified graph type\n");
        return 1;
    }
    std::printf("Graph: %d nodes, %d edges\n", csr.nodes, csr.edges);

    int src = 0;
    std::string src_str;
    args.GetCmdLineArgument("src", src_str);
    if (src_str == "randomize") src = graphio::RandomNode(csr.nodes);
    else if (src_str == "largestdegree") { int md = 0; src = csr.GetNodeWithHighestDegree(md); std::printf("Using highest degree (%d) vertex: %d\n", md, src); }
    else if (!src_str.empty()) src = std::atoi(src_str.c_str());
    if (traversal_mode < 0) traversal_mode = csr.GetAverageDegree() > 8 ? 0 : 1;  // test_bfs.cu:563-566

    int rc;
#define GR_RUN(I, P, D) rc = RunTests<I, P, D>(csr, src, traversal_mode, queue_sizing, iterations, quick != 0, verbose, undirected, device)
    if (instrumented) {
        if (mark_pred) { if (idempotence) GR_RUN(true, true, true); else GR_RUN(true, true, false); }
        else { if (idempotence) GR_RUN(true, false, true); else GR_RUN(true, false, false); }
    } else {
        if (mark_pred) { if (idempotence) GR_RUN(false, true, true); else GR_RUN(false, true, false); }
        else { if (idempotence) GR_RUN(false, false, true); else GR_RUN(false, false, false); }
    }
#undef GR_RUN
    return rc == 0 ? 0 : 2;
}
