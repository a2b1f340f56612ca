// graphio/rmat_device.hpp -- seeded R-MAT tuple generation on the GPU, one lane per generated edge.
//
// Same draw stream and arithmetic as graphio::SeededRmat (rmat.hpp) so host and device tuples are
// bit-identical: every double operation uses the round-to-nearest intrinsics (no FMA contraction),
// and the quadrant rule keeps the reference's strict comparisons (graphio/utils.cuh:58-82).
// First row of SURVEY 8(f) "graph ingest on device"; needed now because the reference generator's
// serial rand() stream (~10 calls per level per edge) cannot produce scale-24+ inputs in bench time.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#include <gunrock/util/error_utils.hpp>

namespace gunrock {
namespace graphio {

__device__ __forceinline__ uint64_t RmatMix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t RmatDraw(uint64_t seed, uint64_t edge, unsigned level, unsigned k)
{
    return RmatMix(seed ^ RmatMix((edge << 10) | (uint64_t(level) << 4) | k));
}
__device__ __forceinline__ double RmatUnit(uint64_t r)
{
    return __dmul_rn(static_cast<double>(r >> 11), 1.0 / 9007199254740992.0);
}

__global__ void SeededRmatKernel(int scale, long long first, long long count, uint64_t seed, double a0, double b0,
                                 double c0, double d0, int *d_rows, int *d_cols)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long t = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; t < count; t += stride) {
        const uint64_t e = static_cast<uint64_t>(first + t);
        double a = a0, b = b0, c = c0, d = d0;
        int u = 0, v = 0;
        for (int level = 0; level < scale; ++level) {
            const int step = 1 << (scale - 1 - level);
            const double p = RmatUnit(RmatDraw(seed, e, level, 0));
            const double ab = __dadd_rn(a, b);
            const double abc = __dadd_rn(ab, c);
            const double abcd = __dadd_rn(abc, d);
            if (p < a) {
            } else if (a < p && p < ab) {
                v += step;
            } else if (ab < p && p < abc) {
                u += step;
            } else if (abc < p && p < abcd) {
                u += step;
                v += step;
            }
            const uint64_t flips = RmatDraw(seed, e, level, 1);
            const double ta = __dmul_rn(__dmul_rn(a, 0.05), RmatUnit(RmatDraw(seed, e, level, 2)));
            const double tb = __dmul_rn(__dmul_rn(b, 0.05), RmatUnit(RmatDraw(seed, e, level, 3)));
            const double tc = __dmul_rn(__dmul_rn(c, 0.05), RmatUnit(RmatDraw(seed, e, level, 4)));
            const double td = __dmul_rn(__dmul_rn(d, 0.05), RmatUnit(RmatDraw(seed, e, level, 5)));
            a = (flips & 1) ? __dadd_rn(a, ta) : __dsub_rn(a, ta);
            b = (flips & 2) ? __dadd_rn(b, tb) : __dsub_rn(b, tb);
            c = (flips & 4) ? __dadd_rn(c, tc) : __dsub_rn(c, tc);
            d = (flips & 8) ? __dadd_rn(d, td) : __dsub_rn(d, td);
            const double s = __dadd_rn(__dadd_rn(__dadd_rn(a, b), c), d);
            a = __ddiv_rn(a, s);
            b = __ddiv_rn(b, s);
            c = __ddiv_rn(c, s);
            d = __ddiv_rn(d, s);
        }
        d_rows[t] = u;
        d_cols[t] = v;
    }
}

inline hipError_t SeededRmatDevice(int scale, long long first, long long count, uint64_t seed, double a, double b,
                                   double c, double d, int *d_rows, int *d_cols, hipStream_t stream)
{
    if (count <= 0) return hipSuccess;
    long long blocks = (count + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(SeededRmatKernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, scale, first, count,
                       seed, a, b, c, d, d_rows, d_cols);
    return util::GRError("SeededRmatKernel launch failed", __FILE__, __LINE__);
}

}  // namespace graphio
}  // namespace gunrock
