// graphio/symmetry.hpp -- is a CSR in HBM its own inverse (every edge has its mirror, with the same multiplicity)?
//
// Callers use the answer for CORRECTNESS (gunrock_bfs_func / gunrock_pr_func take the out-lists as in-lists when it is
// "yes"), so the test is exact and conservative:
//   * every edge (f, t) with f != t -- BOTH orientations, not only f < t: a graph whose unmirrored edges all point from a
//     higher to a lower id (the smallest case is {1 -> 0}) must be reported directed -- is looked up in row t by binary search;
//   * a row that is not STRICTLY increasing (unsorted, or holding a duplicate) makes the answer "no": the search needs sorted
//     rows, and with duplicates equal multiplicities (which a pulled PageRank needs) would have to be counted.  Csr::FromCoo
//     (csr.cuh:263-311) builds sorted, duplicate-free rows, so its graphs never take this exit.
// "No" is always the safe answer: the caller then builds the inverse graph (graphio::DeviceTransposeCsr) or stays top-down.
// No edges-sized scratch: the source of an edge is the row the lane (or wave) is walking.
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/util/device_intrinsics.hpp>
#include <gunrock/util/error_utils.hpp>

namespace gunrock {
namespace graphio {

__device__ __forceinline__ bool RowHolds(const int *d_row_offsets, const int *d_cols, int row, int wanted)
{
    int lo = d_row_offsets[row];
    const int end = d_row_offsets[row + 1];
    int hi = end;
    while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        if (d_cols[mid] < wanted) lo = mid + 1; else hi = mid;
    }
    return lo < end && d_cols[lo] == wanted;
}

// One wave per 64 rows: a short row is checked by its lane, a long one by the whole wave.
static __global__ void SymmetryCheckKernel(const int *d_row_offsets, const int *d_cols, long long nodes, int *d_missing)
{
    const unsigned lane = util::LaneId();
    const long long wave0 = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) / util::kWaveSize;
    const long long nwaves = static_cast<long long>(gridDim.x) * blockDim.x / util::kWaveSize;
    const long long groups = (nodes + 63) / 64;
    bool missing = false;
    for (long long g = wave0; g < groups; g += nwaves) {
        if (__ballot(__hip_atomic_load(d_missing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) break;  // (wave-uniform) somebody already said no
        const long long v = g * 64 + lane;
        int b = 0, e = 0;
        if (v < nodes) { b = d_row_offsets[v]; e = d_row_offsets[v + 1]; }
        const bool long_row = (e - b) > 16;
        if (!long_row) {
            for (int i = b; i < e; ++i) {
                const int t = d_cols[i];
                if (i > b && d_cols[i - 1] >= t) missing = true;                     // unsorted or duplicate
                if (t < 0 || t >= nodes) missing = true;                             // not a vertex
                else if (t != static_cast<int>(v) && !RowHolds(d_row_offsets, d_cols, t, static_cast<int>(v))) missing = true;
            }
        }
        unsigned long long todo = __ballot(long_row);
        while (todo) {
            const int leader = __ffsll(static_cast<long long>(todo)) - 1;
            const int lb = __shfl(b, leader, util::kWaveSize), le = __shfl(e, leader, util::kWaveSize);
            const int lv = static_cast<int>(g * 64 + leader);
            for (int i = lb + static_cast<int>(lane); i < le; i += util::kWaveSize) {
                const int t = d_cols[i];
                if (i > lb && d_cols[i - 1] >= t) missing = true;
                if (t < 0 || t >= nodes) missing = true;
                else if (t != lv && !RowHolds(d_row_offsets, d_cols, t, lv)) missing = true;
            }
            todo &= todo - 1;
        }
    }
    if (__ballot(missing) && lane == 0) __hip_atomic_store(d_missing, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

inline hipError_t DeviceIsSymmetric(int nodes, long long edges, const int *d_row_offsets, const int *d_column_indices, hipStream_t stream,
                                    bool &symmetric)
{
    symmetric = false;
    if (nodes <= 0 || edges <= 0) return hipSuccess;
    int *d_missing = nullptr;
    hipError_t rc = util::GRError(hipMalloc(&d_missing, sizeof(int)), "DeviceIsSymmetric hipMalloc failed", __FILE__, __LINE__);
    if (rc) return rc;
    int missing = 1;
    rc = util::GRError(hipMemsetAsync(d_missing, 0, sizeof(int), stream), "DeviceIsSymmetric memset failed", __FILE__, __LINE__);
    if (!rc) {
        long long grid = ((static_cast<long long>(nodes) + 63) / 64 + 3) / 4;
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(SymmetryCheckKernel, dim3(static_cast<unsigned>(grid)), dim3(256), 0, stream, d_row_offsets, d_column_indices,
                           static_cast<long long>(nodes), d_missing);
        rc = util::GRError(hipGetLastError(), "SymmetryCheckKernel launch failed", __FILE__, __LINE__);
    }
    if (!rc) rc = util::GRError(hipMemcpyAsync(&missing, d_missing, sizeof(int), hipMemcpyDeviceToHost, stream), "DeviceIsSymmetric read failed", __FILE__, __LINE__);
    if (!rc) rc = util::GRError(hipStreamSynchronize(stream), "DeviceIsSymmetric sync failed", __FILE__, __LINE__);
    const hipError_t freed = util::GRError(hipFree(d_missing), "DeviceIsSymmetric hipFree failed", __FILE__, __LINE__);  // on every path
    if (rc) return rc;
    if (freed) return freed;
    symmetric = missing == 0;
    return hipSuccess;
}

}  // namespace graphio
}  // namespace gunrock
