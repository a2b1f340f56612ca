// examples/simple_example.hip -- end-to-end self-check on a Matrix-Market file (default: the reference's bips98_606.mtx).
//
// Same flow as the reference's simple_example (simple_example/simple_example.cu:368-695, main :746-813), restricted to
// the primitives in scope: load the graph UNDIRECTED (:760), connected components on the GPU checked against a host
// union-find (component count, as the reference does with Boost :437-445, plus the labels), histogram, BFS from the root
// of the largest component in idempotent mode (:453-499) checked label by label against a host FIFO BFS (:566), then
// betweenness centrality from every vertex (:592-672: Reset + Enact per source, values halved) checked against a host Brandes
// pass with the reference's float tolerance, then "TEST PASSED" / "TEST FAILED" (:682-687).
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <deque>
#include <vector>

#include <gunrock/app/bc/bc_enactor.hpp>
#include <gunrock/app/bc/bc_problem.hpp>
#include <gunrock/app/bfs/bfs_enactor.hpp>
#include <gunrock/app/bfs/bfs_problem.hpp>
#include <gunrock/app/cc/cc_enactor.hpp>
#include <gunrock/app/cc/cc_problem.hpp>
#include <gunrock/graphio/market.hpp>
#include <gunrock/util/test_utils.hpp>

using namespace gunrock;
using namespace gunrock::app;

static int Find(std::vector<int> &p, int x)
{
    while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
    return x;
}

int main(int argc, char **argv)
{
    util::CommandLineArgs args(argc, argv);
    if (args.ParsedArgc() < 2 || args.Positional(0) != "market") {
        std::printf("usage: simple_example market <file.mtx> [--device=<n>]\n");
        return 1;
    }
    int device = 0;
    args.GetCmdLineArgument("device", device);
    util::DeviceContext context(device);
    std::string file = args.Positional(1);
    Csr<int, int, int> csr(false);
    if (graphio::BuildMarketGraph<false>(const_cast<char *>(file.c_str()), csr, /*undirected*/ true, false) != 0) return 1;
    std::printf("Graph: %d nodes, %d edges (undirected)\n", csr.nodes, csr.edges);
    int num_errors = 0;

    // ---- connected components ----
    typedef cc::CCProblem<int, int, int, true> CcProblem;
    CcProblem cc_problem;
    cc::CCEnactor<false> cc_enactor(false);
    std::vector<int> h_ids(csr.nodes);
    if (util::GRError(cc_problem.Init(false, csr, 1), "CC Problem Initialization Failed", __FILE__, __LINE__)) return 1;
    if (util::GRError(cc_problem.Reset(cc_enactor.GetFrontierType()), "CC Problem Data Reset Failed", __FILE__, __LINE__)) return 1;
    util::GpuTimer cc_timer;
    cc_timer.Start(cc_problem.graph_slices[0]->stream);
    if (util::GRError(cc_enactor.Enact(&cc_problem, 0), "CC Problem Enact Failed", __FILE__, __LINE__)) return 1;
    cc_timer.Stop(cc_problem.graph_slices[0]->stream);
    if (util::GRError(cc_problem.Extract(h_ids.data()), "CC Problem Data Extraction Failed", __FILE__, __LINE__)) return 1;
    std::printf("GPU Connected Component finished in %lf msec.\n", cc_timer.ElapsedMillis());
    std::vector<int> parent(csr.nodes);
    for (int v = 0; v < csr.nodes; ++v) parent[v] = v;
    for (int v = 0; v < csr.nodes; ++v)
        for (int e = csr.row_offsets[v]; e < csr.row_offsets[v + 1]; ++e) {
            int a = Find(parent, v), b = Find(parent, csr.column_indices[e]);
            if (a != b) { if (a < b) parent[b] = a; else parent[a] = b; }
        }
    unsigned ref_components = 0;
    for (int v = 0; v < csr.nodes; ++v) { parent[v] = Find(parent, v); ref_components += parent[v] == v; }
    std::printf("CPU components: %u, GPU components: %u\n", ref_components, cc_problem.num_components);
    if (ref_components != cc_problem.num_components) { std::printf("INCORRECT. Ref Component Count: %u, GPU Computed Component Count: %u\n", ref_components, cc_problem.num_components); ++num_errors; }
    else std::printf("CORRECT.\n");
    for (int v = 0; v < csr.nodes; ++v) num_errors += h_ids[v] != parent[v];

    // ---- histogram: BFS starts at the root of the largest component (simple_example.cu:453-480) ----
    std::vector<int> roots(cc_problem.num_components);
    std::vector<unsigned> histogram(cc_problem.num_components);
    cc_problem.ComputeCCHistogram(h_ids.data(), roots.data(), histogram.data());
    int src = 0;
    unsigned largest = 0;
    for (unsigned i = 0; i < cc_problem.num_components; ++i)
        if (histogram[i] > largest) { largest = histogram[i]; src = roots[i]; }
    std::printf("Largest component: root %d, %u vertices\n", src, largest);

    // ---- BFS, idempotent, no predecessors ----
    typedef bfs::BFSProblem<int, int, int, false, true, false> BfsProblem;
    BfsProblem bfs_problem;
    bfs::BFSEnactor<false> bfs_enactor(false);
    std::vector<int> h_labels(csr.nodes), ref_labels(csr.nodes, -1);
    if (util::GRError(bfs_problem.Init(false, csr, 1), "Problem BFS Initialization Failed", __FILE__, __LINE__)) return 1;
    if (util::GRError(bfs_problem.Reset(src, bfs_enactor.GetFrontierType(), 1.3), "BFS Problem Data Reset Failed", __FILE__, __LINE__)) return 1;
    util::GpuTimer bfs_timer;
    bfs_timer.Start(bfs_problem.graph_slices[0]->stream);
    if (util::GRError(bfs_enactor.Enact<BfsProblem>(context, &bfs_problem, src, 0), "BFS Problem Enact Failed", __FILE__, __LINE__)) return 1;
    bfs_timer.Stop(bfs_problem.graph_slices[0]->stream);
    if (util::GRError(bfs_problem.Extract(h_labels.data(), nullptr), "BFS Problem Data Extraction Failed", __FILE__, __LINE__)) return 1;
    std::printf("GPU BFS finished in %lf msec.\n", bfs_timer.ElapsedMillis());
    ref_labels[src] = 0;
    std::deque<int> fifo(1, src);
    while (!fifo.empty()) {
        const int u = fifo.front();
        fifo.pop_front();
        for (int e = csr.row_offsets[u]; e < csr.row_offsets[u + 1]; ++e) {
            const int w = csr.column_indices[e];
            if (ref_labels[w] == -1) { ref_labels[w] = ref_labels[u] + 1; fifo.push_back(w); }
        }
    }
    std::printf("Label Validity: ");
    num_errors += util::CompareResults(h_labels.data(), ref_labels.data(), csr.nodes, true);
    std::printf("\n");

    // ---- betweenness centrality, all sources (simple_example.cu:592-672) ----
    {
        typedef bc::BCProblem<int, int, float, true, false> BcProblem;
        Csr<int, float, int> fcsr(false);  // the same arrays behind the <int, float, int> graph type BC is instantiated on
        fcsr.nodes = csr.nodes;
        fcsr.edges = csr.edges;
        fcsr.row_offsets = csr.row_offsets;
        fcsr.column_indices = csr.column_indices;
        BcProblem bc_problem;
        bc::BCEnactor<false> bc_enactor(false);
        bool ok = !util::GRError(bc_problem.Init(false, fcsr, 1), "BC Problem Initialization Failed", __FILE__, __LINE__);
        util::GpuTimer bc_timer;
        if (ok) {
            ok = !util::GRError(bc_problem.ClearBcValues(), "BC clear failed", __FILE__, __LINE__);
            bc_timer.Start(bc_problem.graph_slices[0]->stream);
            for (int s = 0; ok && s < csr.nodes; ++s) {
                ok = !util::GRError(bc_problem.Reset(s, bc_enactor.GetFrontierType(), 1.3), "BC Problem Data Reset Failed", __FILE__, __LINE__) &&
                     !util::GRError(bc_enactor.Enact<BcProblem>(context, &bc_problem, s, 0), "BC Problem Enact Failed", __FILE__, __LINE__);
            }
            if (ok) ok = !util::GRError(bc_problem.ScaleBcValues(0.5f), "BC scale failed", __FILE__, __LINE__);
            bc_timer.Stop(bc_problem.graph_slices[0]->stream);
        }
        std::vector<float> h_bc(csr.nodes, 0.f), ref_bc(csr.nodes, 0.f);
        if (ok) ok = !util::GRError(bc_problem.Extract(nullptr, h_bc.data(), nullptr), "BC Problem Data Extraction Failed", __FILE__, __LINE__);
        fcsr.row_offsets = nullptr;
        fcsr.column_indices = nullptr;
        if (!ok) return 1;
        // host Brandes: BFS order + path counts forward, dependencies backward, halved like the GPU values
        {
            std::vector<double> acc(csr.nodes, 0.0), sigma(csr.nodes), delta(csr.nodes);
            std::vector<int> dist(csr.nodes), order;
            order.reserve(csr.nodes);
            for (int s = 0; s < csr.nodes; ++s) {
                std::fill(dist.begin(), dist.end(), -1);
                std::fill(sigma.begin(), sigma.end(), 0.0);
                std::fill(delta.begin(), delta.end(), 0.0);
                order.clear();
                dist[s] = 0;
                sigma[s] = 1.0;
                order.push_back(s);
                for (size_t head = 0; head < order.size(); ++head) {
                    const int u = order[head];
                    for (int e = csr.row_offsets[u]; e < csr.row_offsets[u + 1]; ++e) {
                        const int w = csr.column_indices[e];
                        if (dist[w] < 0) { dist[w] = dist[u] + 1; order.push_back(w); }
                        if (dist[w] == dist[u] + 1) sigma[w] += sigma[u];
                    }
                }
                for (size_t i = order.size(); i-- > 1;) {
                    const int w = order[i];
                    for (int e = csr.row_offsets[w]; e < csr.row_offsets[w + 1]; ++e) {
                        const int u = csr.column_indices[e];  // (undirected graph: predecessors are among the neighbours)
                        if (dist[u] == dist[w] - 1) delta[u] += sigma[u] / sigma[w] * (1.0 + delta[w]);
                    }
                    acc[w] += delta[w];
                }
            }
            for (int v = 0; v < csr.nodes; ++v) ref_bc[v] = static_cast<float>(0.5 * acc[v]);
        }
        std::printf("Validity BC Value: ");
        num_errors += util::CompareResults(h_bc.data(), ref_bc.data(), csr.nodes, true);
        std::printf("\nGPU BC finished in %lf msec.\n", bc_timer.ElapsedMillis());
    }

    if (num_errors == 0) std::printf("\nTEST PASSED\n");
    else std::printf("\nTEST FAILED: %d errors\n", num_errors);
    return num_errors == 0 ? 0 : 2;
}
