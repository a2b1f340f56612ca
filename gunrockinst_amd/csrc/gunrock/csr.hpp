// csr.hpp -- host-side COO tuple and CSR container.
//
// Public fields and method names follow the reference's gunrock::Coo (gunrock/coo.cuh:32-58) and
// gunrock::Csr (gunrock/csr.cuh:38-80) so drivers written against them keep compiling, and
// FromCoo reproduces the reference's graph-defining behaviour (csr.cuh:247-340): stable
// (row, col) sort, self-loop removal, consecutive-duplicate removal keeping the first value.
// Differences by design: the reference's cache files (WriteToFile / FromCsr, csr.cuh:140-232) are whitespace-separated
// TEXT keyed by file name only -- a stale cache silently overrides the input.  Here the cache is
// BINARY (WriteBinary / FromBinary: header + the three arrays, read with three fread calls) and stamped with the source
// file's size and modification time plus the element widths, so a changed input or a foreign file is simply not a hit;
// nothing is cached unless the caller asks (graphio::BuildMarketGraphCached).  Pinned allocation goes through hipHostMalloc.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <gunrock/util/error_utils.hpp>

namespace gunrock {

template <typename VertexId, typename Value>
struct Coo {
    VertexId row;
    VertexId col;
    Value val;

    Coo() {}
    Coo(VertexId r, VertexId c, Value v) : row(r), col(c), val(v) {}
    void Val(Value &value) { value = val; }
};

// (row, col) lexicographic order -- coo.cuh:71-85
template <typename Tuple>
inline bool RowFirstTupleCompare(const Tuple &x, const Tuple &y)
{
    return (x.row < y.row) || (x.row == y.row && x.col < y.col);
}

template <typename VertexId, typename Value, typename SizeT>
struct Csr {
    SizeT nodes = 0;
    SizeT edges = 0;
    SizeT out_nodes = -1;
    SizeT average_degree = 0;

    VertexId *column_indices = nullptr;
    SizeT *row_offsets = nullptr;
    Value *edge_values = nullptr;
    Value *node_values = nullptr;

    Value average_edge_value = 0;
    Value average_node_value = 0;

    bool pinned = false;

    explicit Csr(bool pinned_ = false) : pinned(pinned_) {}
    Csr(const Csr &) = delete;
    Csr &operator=(const Csr &) = delete;
    ~Csr() { Free(); }

    template <typename T>
    T *Alloc(size_t count)
    {
        if (count == 0) count = 1;
        if (pinned) {
            void *p = nullptr;
            if (util::GRError(hipHostMalloc(&p, sizeof(T) * count, hipHostMallocMapped),
                              "Csr hipHostMalloc failed", __FILE__, __LINE__))
                std::exit(1);
            return static_cast<T *>(p);
        }
        return static_cast<T *>(std::malloc(sizeof(T) * count));
    }
    template <typename T>
    void Release(T *&p)
    {
        if (!p) return;
        if (pinned) util::GRError(hipHostFree(p), "Csr hipHostFree failed", __FILE__, __LINE__);
        else std::free(p);
        p = nullptr;
    }

    template <bool LOAD_EDGE_VALUES, bool LOAD_NODE_VALUES>
    void FromScratch(SizeT nodes_, SizeT edges_)
    {
        Free();
        nodes = nodes_;
        edges = edges_;
        row_offsets = Alloc<SizeT>(static_cast<size_t>(nodes) + 1);
        column_indices = Alloc<VertexId>(static_cast<size_t>(edges));
        node_values = LOAD_NODE_VALUES ? Alloc<Value>(static_cast<size_t>(nodes)) : nullptr;
        edge_values = LOAD_EDGE_VALUES ? Alloc<Value>(static_cast<size_t>(edges)) : nullptr;
    }

    // ---- binary cache (role of WriteToFile / FromCsr, csr.cuh:140-232) ----
    struct CacheStamp {
        long long source_size = 0;      // bytes of the file the graph was parsed from
        long long source_mtime_ns = 0;  // its modification time
        unsigned undirected = 0, reversed = 0;
    };
    struct CacheHeader {
        char magic[8];                  // "GRXCSR1"
        unsigned sizeof_vertex, sizeof_size, sizeof_value, has_values;
        unsigned undirected, reversed, reserved0, reserved1;
        long long nodes, edges, source_size, source_mtime_ns;
    };
    static void FillHeader(CacheHeader &h, const CacheStamp &stamp, long long nodes_, long long edges_, bool with_values)
    {
        std::memset(&h, 0, sizeof(h));
        std::memcpy(h.magic, "GRXCSR1", 8);
        h.sizeof_vertex = sizeof(VertexId);
        h.sizeof_size = sizeof(SizeT);
        h.sizeof_value = sizeof(Value);
        h.has_values = with_values ? 1u : 0u;
        h.undirected = stamp.undirected;
        h.reversed = stamp.reversed;
        h.nodes = nodes_;
        h.edges = edges_;
        h.source_size = stamp.source_size;
        h.source_mtime_ns = stamp.source_mtime_ns;
    }
    // Written to "<path>.tmp" and renamed, so a reader never sees half a file.  false = could not write (not an error for
    // the caller: the reference ignores an unwritable cache too, csr.cuh:171-174).
    bool WriteBinary(const char *path, const CacheStamp &stamp) const
    {
        if (!path || !row_offsets || !column_indices) return false;
        std::string tmp = std::string(path) + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f) return false;
        CacheHeader h;
        FillHeader(h, stamp, nodes, edges, edge_values != nullptr);
        bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1;
        ok = ok && std::fwrite(row_offsets, sizeof(SizeT), static_cast<size_t>(nodes) + 1, f) == static_cast<size_t>(nodes) + 1;
        ok = ok && (edges == 0 || std::fwrite(column_indices, sizeof(VertexId), static_cast<size_t>(edges), f) == static_cast<size_t>(edges));
        if (edge_values) ok = ok && (edges == 0 || std::fwrite(edge_values, sizeof(Value), static_cast<size_t>(edges), f) == static_cast<size_t>(edges));
        ok = (std::fclose(f) == 0) && ok;
        if (ok) ok = std::rename(tmp.c_str(), path) == 0;
        if (!ok) std::remove(tmp.c_str());
        return ok;
    }
    // true = the file exists, carries exactly this stamp and these element widths, is complete, and its offsets are a
    // non-decreasing prefix array ending at `edges`; the graph has been loaded.  Anything else: false, the graph is untouched.
    template <bool LOAD_EDGE_VALUES>
    bool FromBinary(const char *path, const CacheStamp &stamp)
    {
        FILE *f = path ? std::fopen(path, "rb") : nullptr;
        if (!f) return false;
        CacheHeader h, want;
        bool ok = std::fread(&h, sizeof(h), 1, f) == 1;
        if (ok) {
            FillHeader(want, stamp, h.nodes, h.edges, h.has_values != 0);
            ok = std::memcmp(&h, &want, sizeof(h)) == 0 && h.nodes >= 0 && h.edges >= 0 && (!LOAD_EDGE_VALUES || h.has_values);
        }
        if (ok) {  // the length must be exactly what the header promises
            const long long body = static_cast<long long>(sizeof(SizeT)) * (h.nodes + 1) + static_cast<long long>(sizeof(VertexId)) * h.edges +
                                   (h.has_values ? static_cast<long long>(sizeof(Value)) * h.edges : 0);
            ok = std::fseek(f, 0, SEEK_END) == 0 && std::ftell(f) == static_cast<long>(sizeof(h)) + body &&
                 std::fseek(f, static_cast<long>(sizeof(h)), SEEK_SET) == 0;
        }
        if (!ok) {
            std::fclose(f);
            return false;
        }
        Csr staged(pinned);
        staged.template FromScratch<LOAD_EDGE_VALUES, false>(static_cast<SizeT>(h.nodes), static_cast<SizeT>(h.edges));
        const size_t n1 = static_cast<size_t>(h.nodes) + 1, m = static_cast<size_t>(h.edges);
        ok = std::fread(staged.row_offsets, sizeof(SizeT), n1, f) == n1;
        ok = ok && (m == 0 || std::fread(staged.column_indices, sizeof(VertexId), m, f) == m);
        if (LOAD_EDGE_VALUES) ok = ok && (m == 0 || std::fread(staged.edge_values, sizeof(Value), m, f) == m);
        std::fclose(f);
        ok = ok && staged.row_offsets[0] == 0 && static_cast<long long>(staged.row_offsets[h.nodes]) == h.edges;
        SizeT with_out_edges = 0;
        for (SizeT v = 0; ok && v < staged.nodes; ++v) {
            ok = staged.row_offsets[v + 1] >= staged.row_offsets[v];
            with_out_edges += (staged.row_offsets[v + 1] > staged.row_offsets[v]);
        }
        for (size_t e = 0; ok && e < m; ++e)  // (a column outside the graph would fault on the GPU)
            ok = static_cast<unsigned long long>(staged.column_indices[e]) < static_cast<unsigned long long>(h.nodes);
        if (!ok) return false;
        Free();
        nodes = staged.nodes;
        edges = staged.edges;
        row_offsets = staged.row_offsets;
        column_indices = staged.column_indices;
        edge_values = staged.edge_values;
        staged.row_offsets = nullptr;
        staged.column_indices = nullptr;
        staged.edge_values = nullptr;
        out_nodes = with_out_edges;
        return true;
    }

    // csr.cuh:247-340.  `coo` is reordered in place.
    template <bool LOAD_EDGE_VALUES, typename Tuple>
    void FromCoo(char * /*output_file: cache files intentionally not written*/, Tuple *coo, SizeT coo_nodes,
                 SizeT coo_edges, bool ordered_rows = false, bool /*undirected*/ = false,
                 bool /*reversed*/ = false, bool quiet = true)
    {
        if (!quiet) {
            std::printf("  Converting %lld vertices, %lld directed edges (%s tuples) to CSR format...\n",
                        (long long)coo_nodes, (long long)coo_edges, ordered_rows ? "ordered" : "unordered");
        }
        FromScratch<LOAD_EDGE_VALUES, false>(coo_nodes, coo_edges);
        if (!ordered_rows && coo_edges > 1)
            std::stable_sort(coo, coo + coo_edges, RowFirstTupleCompare<Tuple>);

        SizeT kept = 0;
        VertexId filled_row = -1;  // last row whose offset has been written
        for (SizeT i = 0; i < coo_edges; ++i) {
            const Tuple &t = coo[i];
            if (t.row == t.col) continue;                                            // self loop
            if (i > 0 && t.row == coo[i - 1].row && t.col == coo[i - 1].col) continue;  // repeat of predecessor
            while (filled_row < t.row) row_offsets[++filled_row] = kept;
            column_indices[kept] = t.col;
            if (LOAD_EDGE_VALUES) edge_values[kept] = static_cast<Value>(t.val);
            ++kept;
        }
        while (filled_row < static_cast<VertexId>(nodes)) row_offsets[++filled_row] = kept;
        edges = kept;

        SizeT with_out_edges = 0;
        for (SizeT v = 0; v < nodes; ++v) with_out_edges += (row_offsets[v + 1] > row_offsets[v]);
        out_nodes = with_out_edges;
    }

    // first vertex of maximal out-degree (strict '>' scan, csr.cuh:442-455)
    int GetNodeWithHighestDegree(int &max_degree)
    {
        int best = 0, best_deg = 0;
        for (SizeT v = 0; v < nodes; ++v) {
            int deg = static_cast<int>(row_offsets[v + 1] - row_offsets[v]);
            if (deg > best_deg) { best_deg = deg; best = static_cast<int>(v); }
        }
        max_degree = best_deg;
        return best;
    }

    // truncated running mean (csr.cuh:475-485); drives the LB/TWC choice (tests/bfs/test_bfs.cu:563-566)
    SizeT GetAverageDegree()
    {
        if (average_degree == 0) {
            double mean = 0, count = 0;
            for (SizeT v = 0; v < nodes; ++v) {
                count += 1;
                mean += (row_offsets[v + 1] - row_offsets[v] - mean) / count;
            }
            average_degree = static_cast<SizeT>(mean);
        }
        return average_degree;
    }

    // running mean over edge values, skipping UINT_MAX sentinels (csr.cuh:505-517)
    Value GetAverageEdgeValue()
    {
        if (average_edge_value == 0 && edge_values) {
            double mean = 0, count = 0;
            for (SizeT e = 0; e < edges; ++e) {
                if (static_cast<unsigned long long>(edge_values[e]) < 0xFFFFFFFFull) {
                    count += 1;
                    mean += (edge_values[e] - mean) / count;
                }
            }
            average_edge_value = static_cast<Value>(mean);
        }
        return average_edge_value;
    }

    void DisplayGraph(const char *name = "", SizeT limit = 40) const
    {
        SizeT shown = nodes < limit ? nodes : limit;
        std::printf("%s: first %lld nodes (of %lld nodes, %lld edges)\n", name, (long long)shown,
                    (long long)nodes, (long long)edges);
        for (SizeT v = 0; v < shown; ++v) {
            std::printf("%lld:", (long long)v);
            for (SizeT e = row_offsets[v]; e < row_offsets[v + 1]; ++e)
                std::printf(" %lld", (long long)column_indices[e]);
            std::printf("\n");
        }
    }

    void Free()
    {
        Release(row_offsets);
        Release(column_indices);
        Release(edge_values);
        Release(node_values);
        nodes = 0;
        edges = 0;
    }
};

}  // namespace gunrock
