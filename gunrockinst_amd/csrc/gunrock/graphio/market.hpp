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
// The .<name>_{undirected,reversed,nonreversed}_csr cache files (:257-265,313-333) exist only behind BuildMarketGraphCached:
// binary, stamped with the source's size and mtime (csr.hpp), never consulted by the plain BuildMarketGraph.
#pragma once

#include <sys/stat.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <gunrock/csr.hpp>

namespace gunrock {
namespace graphio {

// One "%lld" of the reference's sscanf calls: skip white space, optional sign, at least one digit.  1 = converted, 0 = no
// digits there (matching failure), -1 = the line ended first.
inline int ScanLongLong(const char *&p, const char *end, long long &out)
{
    while (p < end && std::isspace(static_cast<unsigned char>(*p))) ++p;
    if (p == end) return -1;
    const char *q = p;
    bool negative = false;
    if (*q == '+' || *q == '-') negative = (*q++ == '-');
    if (q == end || *q < '0' || *q > '9') return 0;
    unsigned long long acc = 0;
    const unsigned long long limit = negative ? 9223372036854775808ull : 9223372036854775807ull;
    while (q < end && *q >= '0' && *q <= '9') {
        const unsigned digit = static_cast<unsigned>(*q - '0');
        acc = (acc > (limit - digit) / 10) ? limit : acc * 10 + digit;  // (clamps like strtoll)
        ++q;
    }
    out = negative ? static_cast<long long>(0ull - acc) : static_cast<long long>(acc);
    p = q;
    return 1;
}
// sscanf(line, "%lld %lld %lld"): how many numbers were converted before the first failure (0 also stands for EOF)
inline int ScanNumbers(const char *p, const char *end, long long *v, int want)
{
    int got = 0;
    while (got < want && ScanLongLong(p, end, v[got]) == 1) ++got;
    return got;
}

// The whole file is in memory; lines are cut exactly as the reference's loop `fscanf("%1023[^\n]\n", line)` cuts them
// (market.cuh:81-83): up to 1023 characters that are not a newline (a longer line continues as the NEXT "line"), then ALL
// following white space is swallowed -- so blank lines inside the body vanish, and a newline at the very start of the
// input ends the parse.  The numbers of a line are scanned by hand (ScanLongLong = "%lld"): the fscanf + sscanf pair of
// the reference costs ~0.2 us per line, this loop a tenth of it.
template <bool LOAD_VALUES, typename VertexId, typename Value, typename SizeT>
int ReadMarketBuffer(const char *buf, size_t len, char *output_file, Csr<VertexId, Value, SizeT> &csr_graph, bool undirected,
                     bool reversed, bool quiet = true)
{
    typedef Coo<VertexId, long long> Tuple;  // value kept at parse width until the CSR cast
    Tuple *coo = nullptr;
    long long declared = 0, stored = -1;
    SizeT nodes = 0;
    const char *at = buf, *const end = buf + len;

    while (at < end && *at != '\n') {
        const char *line = at;
        const char *stop = (end - at > 1023) ? at + 1023 : end;
        while (at < stop && *at != '\n') ++at;
        const char *line_end = at;
        while (at < end && std::isspace(static_cast<unsigned char>(*at))) ++at;
        if (line[0] == '%') continue;
        if (stored < 0) {
            long long dims[3];
            if (ScanNumbers(line, line_end, dims, 3) != 3) {
                std::fprintf(stderr, "Error parsing MARKET graph: invalid problem description.\n");
                return -1;
            }
            if (dims[0] != dims[1]) {
                std::fprintf(stderr, "Error parsing MARKET graph: not square (%lld, %lld)\n", dims[0], dims[1]);
                return -1;
            }
            nodes = static_cast<SizeT>(dims[0]);
            declared = undirected ? 2 * dims[2] : dims[2];
            coo = static_cast<Tuple *>(std::malloc(sizeof(Tuple) * static_cast<size_t>(declared > 0 ? declared : 1)));
            stored = 0;
            if (!quiet) std::printf(" (%lld nodes, %lld directed edges)... ", dims[0], dims[2]);
            continue;
        }
        if (stored >= declared) {
            std::fprintf(stderr, "Error parsing MARKET graph: encountered more than %lld edges\n", declared);
            std::free(coo);
            return -1;
        }
        long long num[3] = {0, 0, 1};
        const int got = ScanNumbers(line, line_end, num, LOAD_VALUES ? 3 : 2);
        if (got < 2) {
            std::fprintf(stderr, "Error parsing MARKET graph: badly formed edge\n");
            std::free(coo);
            return -1;
        }
        const long long w = (got == 2) ? 1 : num[2];
        const VertexId from = static_cast<VertexId>(num[1] - 1);  // second number = source row
        const VertexId to = static_cast<VertexId>(num[0] - 1);    // first number  = destination column
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
int ReadMarketStream(FILE *f_in, char *output_file, Csr<VertexId, Value, SizeT> &csr_graph, bool undirected,
                     bool reversed, bool quiet = true)
{
    std::vector<char> text;
    size_t used = 0;
    for (;;) {  // (works for stdin too: no seeking)
        if (text.size() - used < (1u << 20)) text.resize(text.size() < (1u << 22) ? (1u << 22) : text.size() * 2);
        const size_t got = std::fread(text.data() + used, 1, text.size() - used, f_in);
        used += got;
        if (got == 0) break;
    }
    return ReadMarketBuffer<LOAD_VALUES>(text.data(), used, output_file, csr_graph, undirected, reversed, quiet);
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

// BuildMarketGraph with the reference's cache rule (market.cuh:296-339: "<dir>/.<name>_undirected_csr" etc., written after the
// first parse and preferred afterwards), made safe: the cache is binary, lives at "<that name>.bin" and is a hit only when it
// was written for a source file of exactly this size and modification time (Csr::FromBinary).  *cache_hit (optional) tells
// which way the graph came.  An unwritable directory is not an error.
template <bool LOAD_VALUES, typename VertexId, typename Value, typename SizeT>
int BuildMarketGraphCached(char *file_in, Csr<VertexId, Value, SizeT> &graph, bool undirected, bool reversed, int *cache_hit = nullptr,
                           bool quiet = true)
{
    if (cache_hit) *cache_hit = 0;
    if (!file_in) return BuildMarketGraph<LOAD_VALUES>(file_in, graph, undirected, reversed, quiet);
    struct stat st;
    if (::stat(file_in, &st) != 0) {
        std::perror("Unable to open file");
        return 1;
    }
    const bool rev = reversed && !undirected;
    typename Csr<VertexId, Value, SizeT>::CacheStamp stamp;
    stamp.source_size = static_cast<long long>(st.st_size);
    stamp.source_mtime_ns = static_cast<long long>(st.st_mtim.tv_sec) * 1000000000ll + st.st_mtim.tv_nsec;
    stamp.undirected = undirected ? 1u : 0u;
    stamp.reversed = rev ? 1u : 0u;
    const std::string path(file_in);
    const size_t slash = path.find_last_of('/');
    const std::string dir = slash == std::string::npos ? std::string(".") : path.substr(0, slash == 0 ? 1 : slash);
    const std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
    const std::string cache = dir + "/." + name + (undirected ? "_undirected_csr" : rev ? "_reversed_csr" : "_nonreversed_csr") + ".bin";
    if (graph.template FromBinary<true>(cache.c_str(), stamp)) {
        if (cache_hit) *cache_hit = 1;
        if (!quiet) std::printf("  Reading directly from previously stored CSR arrays ...\n");
        return 0;
    }
    if (BuildMarketGraph<LOAD_VALUES>(file_in, graph, undirected, reversed, quiet) != 0) return 1;
    graph.WriteBinary(cache.c_str(), stamp);
    return 0;
}

}  // namespace graphio
}  // namespace gunrock
