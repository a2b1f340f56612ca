// graphio/rmat.hpp -- R-MAT generators (host).
//
// BuildRmatGraph: the reference generator (gunrock/graphio/rmat.cuh:27-91, utils.cuh:47-130) on the
// libc rand() stream, kept so `test_bfs rmat` reproduces the reference's 2^10 graph.
// BuildSeededRmatGraph: the benchmark generator of SURVEY 8(d): identical quadrant/noise rules, but
// every draw is splitmix64(seed, edge, level, k), so any slice of the edge list can be produced
// independently (host threads here, one lane per edge in graphio/rmat_device.hpp).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include <gunrock/csr.hpp>
#include <gunrock/graphio/utils.hpp>

namespace gunrock {
namespace graphio {

template <bool WITH_VALUES, typename VertexId, typename Value, typename SizeT>
int BuildRmatGraph(SizeT nodes, SizeT edges, Csr<VertexId, Value, SizeT> &graph, bool undirected,
                   double a0 = 0.55, double b0 = 0.2, double c0 = 0.2, double d0 = 0.05)
{
    typedef Coo<VertexId, Value> Tuple;
    if (nodes < 0 || edges < 0) {
        std::fprintf(stderr, "Invalid graph size: nodes=%lld, edges=%lld", (long long)nodes, (long long)edges);
        return -1;
    }
    const SizeT total = undirected ? edges * 2 : edges;
    Tuple *coo = static_cast<Tuple *>(std::malloc(sizeof(Tuple) * static_cast<size_t>(total > 0 ? total : 1)));
    const double noise = 0.05;
    for (SizeT i = 0; i < edges; ++i) {
        double p[4] = {a0, b0, c0, d0};
        VertexId u = 1, v = 1;
        for (VertexId step = nodes / 2; step >= 1; step /= 2) {
            ChoosePartition(&u, &v, step, p[0], p[1], p[2], p[3], Sprng());
            for (int k = 0; k < 4; ++k) {            // Flip() first, then Sprng(), per parameter
                if (Flip()) p[k] += p[k] * noise * Sprng();
                else p[k] -= p[k] * noise * Sprng();
            }
            const double s = p[0] + p[1] + p[2] + p[3];
            for (int k = 0; k < 4; ++k) p[k] = p[k] / s;
        }
        coo[i] = Tuple(u - 1, v - 1, 1);
        if (undirected) coo[edges + i] = Tuple(v - 1, u - 1, 1);
    }
    graph.template FromCoo<WITH_VALUES>(nullptr, coo, nodes, total);
    std::free(coo);
    return 0;
}

// ---- seeded counter-based stream (shared spec with the device generator and the test oracle) ----
struct SeededRmat {
    static inline uint64_t Mix(uint64_t x)
    {
        x += 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        return x ^ (x >> 31);
    }
    static inline uint64_t Draw(uint64_t seed, uint64_t edge, unsigned level, unsigned k)
    {
        return Mix(seed ^ Mix((edge << 10) | (uint64_t(level) << 4) | k));
    }
    static inline double Unit(uint64_t r) { return double(r >> 11) * (1.0 / 9007199254740992.0); }

    template <typename VertexId>
    static inline void Edge(int scale, uint64_t seed, uint64_t e, double a, double b, double c, double d,
                            VertexId &u_out, VertexId &v_out)
    {
        VertexId u = 0, v = 0;
        for (int level = 0; level < scale; ++level) {
            const VertexId step = VertexId(1) << (scale - 1 - level);
            ChoosePartition(&u, &v, step, a, b, c, d, Unit(Draw(seed, e, level, 0)));
            const uint64_t flips = Draw(seed, e, level, 1);
            const double ta = (a * 0.05) * Unit(Draw(seed, e, level, 2));
            const double tb = (b * 0.05) * Unit(Draw(seed, e, level, 3));
            const double tc = (c * 0.05) * Unit(Draw(seed, e, level, 4));
            const double td = (d * 0.05) * Unit(Draw(seed, e, level, 5));
            a = (flips & 1) ? a + ta : a - ta;
            b = (flips & 2) ? b + tb : b - tb;
            c = (flips & 4) ? c + tc : c - tc;
            d = (flips & 8) ? d + td : d - td;
            const double s = ((a + b) + c) + d;
            a = a / s; b = b / s; c = c / s; d = d / s;
        }
        u_out = u;
        v_out = v;
    }
};

template <bool WITH_VALUES, typename VertexId, typename Value, typename SizeT>
int BuildSeededRmatGraph(int scale, long long pairs, uint64_t seed, Csr<VertexId, Value, SizeT> &graph,
                         bool undirected, double a = 0.55, double b = 0.2, double c = 0.2, double d = 0.05)
{
    typedef Coo<VertexId, Value> Tuple;
    const long long total = undirected ? 2 * pairs : pairs;
    if (scale < 1 || scale > 30 || pairs < 0 || total > 0x7fffffffLL) {
        std::fprintf(stderr, "Invalid seeded R-MAT size: scale=%d pairs=%lld\n", scale, pairs);
        return -1;
    }
    Tuple *coo = static_cast<Tuple *>(std::malloc(sizeof(Tuple) * static_cast<size_t>(total > 0 ? total : 1)));
    for (long long i = 0; i < pairs; ++i) {
        VertexId u, v;
        SeededRmat::Edge(scale, seed, static_cast<uint64_t>(i), a, b, c, d, u, v);
        coo[i] = Tuple(u, v, 1);
        if (undirected) coo[pairs + i] = Tuple(v, u, 1);
    }
    graph.template FromCoo<WITH_VALUES>(nullptr, coo, SizeT(1) << scale, static_cast<SizeT>(total));
    std::free(coo);
    return 0;
}

}  // namespace graphio
}  // namespace gunrock
