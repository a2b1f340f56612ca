// graphio/market.hpp -- MatrixMarket coordinate reader -> CSR (host).
//
// Behavioural contract taken from the reference's graphio::ReadMarketStream / BuildMarketGraph
// (gunrock/graphio/market.cuh:56-215, 249-339); results must be identical because the graph defines
// every downstream answer:
//   * '%' lines are comments; the first other line is "M N L" and must be square (:85-118)
//   * every entry is read "col row [val]", i.e. the edge goes from the SECOND number to the FIRST
//     (:139-141,166-167); `reversed` (directed only) swaps them back (:161-164)
//   * values are parsed as integers (%lld: "1.5e3" -> 1); a missing value is 1 (:146-148)
//   * `undirected` stores the mirrored tuple right after each entry and doubles the count (:108,173-184)
//   * the scanner reads one non-empty line then swallows all following white space, so blank lines in
//     the body are skipped and an empty FIRST line ends the parse (:81-83)
//   * the 4-argument BuildMarketGraph always loads edge values, whatever LOAD_VALUES says (:317,324,331)
// Not reproduced: the .<name>_{undirected,reversed,nonreversed}_csr cache files (:257-265,313-333).
#pragma once

#include <cstdio>
#include <cstdlib>

#include <gunrock/csr.hpp>

namespace gunrock {
namespace graphio {

template <bool LOAD_VALUES, typename VertexId, typename Value, typename SizeT>
int ReadMarketStream(FILE *f_in, char *output_file, Csr<VertexId, Value, SizeT> &csr_graph, bool undirected,
                     bool reversed, bool quiet = true)
{
    typedef Coo<VertexId, long long> Tuple;  // value kept at parse width until the CSR cast
    Tuple *coo = nullptr;
    long long declared = 0, stored = -1;
    SizeT nodes = 0;
    char line[1024];

    while (std::fscanf(f_in, "%1023[^\n]\n", line) > 0) {
        if (line[0] == '%') continue;
        if (stored < 0) {
            long long nx, ny, ne;
            if (std::sscanf(line, "%lld %lld %lld", &nx, &ny, &ne) != 3) {
                std::fprintf(stderr, "Error parsing MARKET graph: invalid problem description.\n");
                return -1;
            }
            if (nx != ny) {
                std::fprintf(stderr, "Error parsing MARKET graph: not square (%lld, %lld)\n", nx, ny);
                return -1;
            }
            nodes = static_cast<SizeT>(nx);
            declared = undirected ? 2 * ne : ne;
            coo = static_cast<Tuple *>(std::malloc(sizeof(Tuple) * static_cast<size_t>(declared > 0 ? declared : 1)));
            stored = 0;
            if (!quiet) std::printf(" (%lld nodes, %lld directed edges)... ", nx, ne);
            continue;
        }
        if (stored >= declared) {
            std::fprintf(stderr, "Error parsing MARKET graph: encountered more than %lld edges\n", declared);
            std::free(coo);
            return -1;
        }
        long long n1, n2, w = 1;
        int got = LOAD_VALUES ? std::sscanf(line, "%lld %lld %lld", &n1, &n2, &w)
                              : std::sscanf(line, "%lld %lld", &n1, &n2);
        if (got < 2) {
            std::fprintf(stderr, "Error parsing MARKET graph: badly formed edge\n");
            std::free(coo);
            return -1;
        }
        if (got == 2) w = 1;
        const VertexId from = static_cast<VertexId>(n2 - 1);  // second number = source row
        const VertexId to = static_cast<VertexId>(n1 - 1);    // first number  = destination column
        if (reversed && !undirected) coo[stored++] = Tuple(to, from, w);
        else coo[stored++] = Tuple(from, to, w);
        if (undirected) coo[stored++] = Tuple(to, from, w);
    }

    if (!coo) {
        std::fprintf(stderr, "No graph found\n");
        return -1;
    }
    if (stored != declared) {
        std::fprintf(stderr, "Error parsing MARKET graph: only %lld/%lld edges read\n", stored, declared);
        std::free(coo);
        return -1;
    }
    csr_graph.template FromCoo<LOAD_VALUES>(output_file, coo, nodes, static_cast<SizeT>(declared), false,
                                            undirected, reversed, quiet);
    std::free(coo);
    return 0;
}

template <bool LOAD_VALUES, typename VertexId, typename Value, typename SizeT>
int BuildMarketGraph(char *mm_filename, char *output_file, Csr<VertexId, Value, SizeT> &csr_graph,
                     bool undirected, bool reversed, bool quiet = true)
{
    if (mm_filename == nullptr) {
        if (!quiet) std::printf("Reading from stdin:\n");
        return ReadMarketStream<LOAD_VALUES>(stdin, output_file, csr_graph, undirected, reversed, quiet);
    }
    FILE *f_in = std::fopen(mm_filename, "r");
    if (!f_in) {
        std::perror("Unable to open file");
        return -1;
    }
    if (!quiet) std::printf("Reading from %s:\n", mm_filename);
    int rc = ReadMarketStream<LOAD_VALUES>(f_in, output_file, csr_graph, undirected, reversed, quiet);
    std::fclose(f_in);
    return rc;
}

// 4-argument form used by every driver (market.cuh:296-339): values are always loaded.
template <bool LOAD_VALUES, typename VertexId, typename Value, typename SizeT>
int BuildMarketGraph(char *file_in, Csr<VertexId, Value, SizeT> &graph, bool undirected, bool reversed,
                     bool quiet = true)
{
    return BuildMarketGraph<true>(file_in, nullptr, graph, undirected, reversed && !undirected, quiet) != 0 ? 1 : 0;
}

}  // namespace graphio
}  // namespace gunrock
