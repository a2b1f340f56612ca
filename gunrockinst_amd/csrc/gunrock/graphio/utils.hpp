// graphio/utils.hpp -- random helpers shared by the graph builders.
//
// RandomNode / RandomBits keep the reference's libc rand() behaviour (graphio/utils.cuh:38-45,
// util/random_bits.h:45-75: each key byte is rand()>>7) so `src_mode = randomize` picks the same
// vertex as the reference would in a fresh process.  The R-MAT helpers follow utils.cuh:47-130.
#pragma once

#include <cstdlib>
#include <cstring>

namespace gunrock {
namespace util {

template <typename K>
inline void RandomBits(K &key, int entropy_reduction = 0, int lower_key_bits = sizeof(K) * 8)
{
    unsigned char bytes[sizeof(K)];
    do {
        for (size_t j = 0; j < sizeof(K); ++j) {
            unsigned char q = 0xff;
            for (int i = 0; i <= entropy_reduction; ++i) q &= static_cast<unsigned char>(std::rand() >> 7);
            bytes[j] = q;
        }
        if (lower_key_bits < static_cast<int>(sizeof(K) * 8)) {
            unsigned long long base = 0;
            std::memcpy(&base, bytes, sizeof(K));
            base &= (1ull << lower_key_bits) - 1;
            std::memcpy(bytes, &base, sizeof(K));
        }
        std::memcpy(&key, bytes, sizeof(K));
    } while (key != key);
}

}  // namespace util

namespace graphio {

template <typename SizeT>
inline SizeT RandomNode(SizeT num_nodes)
{
    SizeT id;
    util::RandomBits(id);
    if (id < 0) id *= -1;
    return id % num_nodes;
}

inline double Sprng() { return double(std::rand()) / RAND_MAX; }
inline bool Flip() { return std::rand() >= RAND_MAX / 2; }

// quadrant choice with the reference's strict comparisons (a draw that lands exactly on a
// boundary selects no move) -- utils.cuh:58-82
template <typename VertexId>
inline void ChoosePartition(VertexId *u, VertexId *v, VertexId step, double a, double b, double c, double d,
                            double p)
{
    if (p < a) return;
    if (a < p && p < a + b) { *v += step; return; }
    if (a + b < p && p < a + b + c) { *u += step; return; }
    if (a + b + c < p && p < a + b + c + d) { *u += step; *v += step; }
}

}  // namespace graphio
}  // namespace gunrock
