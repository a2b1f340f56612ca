// graphio/device_sort.hpp -- device-wide helpers built on the ingest kernels of device_csr.hpp:
//   * DeviceKeySort     LSD radix sort of 64-bit keys (the reference sorts (value, id) pairs with cub::DeviceRadixSort
//                       behind util::CUBRadixSort, gunrock/util/sort_utils.cuh:31-158 -- PageRank's final ranking,
//                       pr_enactor.cuh:513-516, and TopK's degree ranking, topk_enactor.cuh:262-272).  Callers pack
//                       (sort value << 32 | id): equal values then order by id, which is what a stable pair sort gives.
//   * DeviceTransposeCsr CSR -> CSR of the reversed edges (CSC) by counting sort on the destination: in-degree histogram,
//                       exclusive scan, scatter through per-row cursors.  Every edge is kept (no de-duplication); the
//                       order of a row's entries is the arrival order of the scatter, i.e. not deterministic.
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/graphio/device_csr.hpp>

namespace gunrock {
namespace graphio {

struct DeviceKeySort {
    unsigned long long *d_keys[2] = {nullptr, nullptr};
    unsigned *d_hist = nullptr;
    unsigned long long *d_offsets = nullptr;
    unsigned long long *d_sums = nullptr;
    long long capacity = 0;

    void Release()
    {
        for (int i = 0; i < 2; ++i)
            if (d_keys[i]) util::GRError(hipFree(d_keys[i]), "DeviceKeySort hipFree failed", __FILE__, __LINE__);
        if (d_hist) util::GRError(hipFree(d_hist), "DeviceKeySort hipFree failed", __FILE__, __LINE__);
        if (d_offsets) util::GRError(hipFree(d_offsets), "DeviceKeySort hipFree failed", __FILE__, __LINE__);
        if (d_sums) util::GRError(hipFree(d_sums), "DeviceKeySort hipFree failed", __FILE__, __LINE__);
        d_keys[0] = d_keys[1] = nullptr;
        d_hist = nullptr;
        d_offsets = nullptr;
        d_sums = nullptr;
        capacity = 0;
    }
    ~DeviceKeySort() { Release(); }

    // room for n keys; the caller writes them to Keys()
    hipError_t Reserve(long long n)
    {
        hipError_t retval = hipSuccess;
        if (n <= capacity) return retval;
        Release();
        const long long tiles = (n + kSortTile - 1) / kSortTile;
        const long long hist_words = 256 * (tiles > 0 ? tiles : 1);
        for (int i = 0; i < 2; ++i)
            GR_CHECK(hipMalloc(&d_keys[i], sizeof(unsigned long long) * static_cast<size_t>(n > 0 ? n : 1)), "DeviceKeySort hipMalloc failed");
        GR_CHECK(hipMalloc(&d_hist, sizeof(unsigned) * static_cast<size_t>(hist_words)), "DeviceKeySort hipMalloc failed");
        GR_CHECK(hipMalloc(&d_offsets, sizeof(unsigned long long) * static_cast<size_t>(hist_words)), "DeviceKeySort hipMalloc failed");
        GR_CHECK(hipMalloc(&d_sums, sizeof(unsigned long long) * static_cast<size_t>(ScanScratchWords(hist_words))), "DeviceKeySort hipMalloc failed");
        capacity = n;
        return retval;
    }
    unsigned long long *Keys() { return d_keys[0]; }

    // ascending sort of the n keys in Keys() over their low `key_bits` bits; *sorted receives the buffer that holds the result
    hipError_t Sort(long long n, int key_bits, hipStream_t stream, unsigned long long **sorted)
    {
        hipError_t retval = hipSuccess;
        int cur = 0;
        if (n > 1) {
            const long long tiles = (n + kSortTile - 1) / kSortTile;
            const long long hist_words = 256 * tiles;
            for (int shift = 0; shift < key_bits; shift += 8) {
                hipLaunchKernelGGL(RadixHistogramKernel, dim3(static_cast<unsigned>(tiles)), dim3(kSortThreads), 0, stream, d_keys[cur], n, shift,
                                   tiles, d_hist);
                GR_CHECK(hipGetLastError(), "RadixHistogramKernel launch failed");
                GR_CHECK(DeviceExclusiveScan<unsigned long long>(d_hist, d_offsets, hist_words, d_sums, stream), "radix scan failed");
                hipLaunchKernelGGL(RadixScatterKernel, dim3(static_cast<unsigned>(tiles)), dim3(kSortThreads), 0, stream, d_keys[cur], n, shift,
                                   tiles, d_offsets, d_keys[cur ^ 1]);
                GR_CHECK(hipGetLastError(), "RadixScatterKernel launch failed");
                cur ^= 1;
            }
        }
        *sorted = d_keys[cur];
        return retval;
    }
};

// ---- transpose ----
static __global__ void InDegreeKernel(const int *d_cols, long long m, unsigned *d_counts)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < m; e += stride)
        atomicAdd(d_counts + d_cols[e], 1u);
}

// one wave per 64 rows of the forward graph (short rows by their lane, long rows by the wave): edge (v, u) lands in row u
static __global__ void TransposeScatterKernel(const int *d_row_offsets, const int *d_cols, int nodes, const int *d_inv_row_offsets,
                                              unsigned *d_cursor, int *d_inv_cols, const unsigned *d_vals = nullptr,
                                              unsigned *d_inv_vals = nullptr)
{
    const unsigned lane = util::LaneId();
    const long long wave0 = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) / util::kWaveSize;
    const long long nwaves = static_cast<long long>(gridDim.x) * blockDim.x / util::kWaveSize;
    const long long groups = (static_cast<long long>(nodes) + 63) / 64;
    for (long long g = wave0; g < groups; g += nwaves) {
        const long long v = g * 64 + lane;
        int b = 0, e = 0;
        if (v < nodes) { b = d_row_offsets[v]; e = d_row_offsets[v + 1]; }
        const bool long_row = (e - b) > 16;
        if (!long_row)
            for (int i = b; i < e; ++i) {
                const int u = d_cols[i];
                const int at = d_inv_row_offsets[u] + static_cast<int>(atomicAdd(d_cursor + u, 1u));
                d_inv_cols[at] = static_cast<int>(v);
                if (d_inv_vals) d_inv_vals[at] = d_vals[i];
            }
        unsigned long long todo = __ballot(long_row);
        while (todo) {
            const int leader = __ffsll(static_cast<long long>(todo)) - 1;
            const int lb = __shfl(b, leader, util::kWaveSize), le = __shfl(e, leader, util::kWaveSize);
            const int lv = static_cast<int>(g * 64 + leader);
            for (int i = lb + static_cast<int>(lane); i < le; i += util::kWaveSize) {
                const int u = d_cols[i];
                const int at = d_inv_row_offsets[u] + static_cast<int>(atomicAdd(d_cursor + u, 1u));
                d_inv_cols[at] = lv;
                if (d_inv_vals) d_inv_vals[at] = d_vals[i];
            }
            todo &= todo - 1;
        }
    }
}

// d_inv_row_offsets[nodes + 1], d_inv_cols[edges]: caller-owned device arrays
// d_vals / d_inv_vals (optional): one 32-bit value per edge that travels with it (SSSP weights)
inline hipError_t DeviceTransposeCsr(int nodes, long long edges, const int *d_row_offsets, const int *d_cols, int *d_inv_row_offsets,
                                     int *d_inv_cols, hipStream_t stream, const unsigned *d_vals = nullptr, unsigned *d_inv_vals = nullptr)
{
    hipError_t retval = hipSuccess;
    unsigned *d_counts = nullptr;
    unsigned long long *d_sums = nullptr;
    const long long words = static_cast<long long>(nodes) + 1;
    GR_CHECK(hipMalloc(&d_counts, sizeof(unsigned) * static_cast<size_t>(words)), "DeviceTransposeCsr hipMalloc failed");
    GR_CHECK(hipMalloc(&d_sums, sizeof(unsigned long long) * static_cast<size_t>(ScanScratchWords(words))), "DeviceTransposeCsr hipMalloc failed");
    GR_CHECK(hipMemsetAsync(d_counts, 0, sizeof(unsigned) * static_cast<size_t>(words), stream), "DeviceTransposeCsr memset failed");
    if (edges > 0) {
        hipLaunchKernelGGL(InDegreeKernel, dim3(2048), dim3(256), 0, stream, d_cols, edges, d_counts);
        GR_CHECK(hipGetLastError(), "InDegreeKernel launch failed");
    }
    GR_CHECK(DeviceExclusiveScan<int>(d_counts, d_inv_row_offsets, words, d_sums, stream), "DeviceTransposeCsr scan failed");
    GR_CHECK(hipMemsetAsync(d_counts, 0, sizeof(unsigned) * static_cast<size_t>(words), stream), "DeviceTransposeCsr memset failed");
    if (edges > 0) {
        hipLaunchKernelGGL(TransposeScatterKernel, dim3(2048), dim3(256), 0, stream, d_row_offsets, d_cols, nodes, d_inv_row_offsets, d_counts,
                           d_inv_cols, d_vals, d_inv_vals);
        GR_CHECK(hipGetLastError(), "TransposeScatterKernel launch failed");
    }
    GR_CHECK(hipStreamSynchronize(stream), "DeviceTransposeCsr sync failed");
    GR_CHECK(hipFree(d_counts), "DeviceTransposeCsr hipFree failed");
    GR_CHECK(hipFree(d_sums), "DeviceTransposeCsr hipFree failed");
    return retval;
}

}  // namespace graphio
}  // namespace gunrock
