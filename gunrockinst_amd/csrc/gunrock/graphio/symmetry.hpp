// graphio/symmetry.hpp -- is a CSR in HBM its own inverse (every edge has its mirror)?
// One binary search per edge with from < to (rows sorted by column, as Csr::FromCoo leaves them, csr.cuh:263-311; an unsorted
// row can only make the answer "no", which every caller treats as "build or do without the inverse").
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/app/cc/cc_problem.hpp>  // ExpandRowsKernel, MirrorCheckKernel

namespace gunrock {
namespace graphio {

inline hipError_t DeviceIsSymmetric(int nodes, long long edges, const int *d_row_offsets, const int *d_column_indices, hipStream_t stream,
                                    bool &symmetric)
{
    hipError_t retval = hipSuccess;
    symmetric = false;
    if (nodes <= 0 || edges <= 0) return retval;
    int *d_froms = nullptr, *d_missing = nullptr;
    GR_CHECK(hipMalloc(&d_froms, sizeof(int) * static_cast<size_t>(edges)), "DeviceIsSymmetric hipMalloc failed");
    GR_CHECK(hipMalloc(&d_missing, sizeof(int)), "DeviceIsSymmetric hipMalloc failed");
    GR_CHECK(hipMemsetAsync(d_missing, 0, sizeof(int), stream), "DeviceIsSymmetric memset failed");
    hipLaunchKernelGGL((app::cc::ExpandRowsKernel<int, int>), dim3(2048), dim3(256), 0, stream, d_row_offsets, nodes, d_froms);
    hipLaunchKernelGGL((app::cc::MirrorCheckKernel<int, int>), dim3(4096), dim3(256), 0, stream, d_row_offsets, d_froms, d_column_indices,
                       edges, d_missing);
    GR_CHECK(hipGetLastError(), "MirrorCheckKernel launch failed");
    int missing = 1;
    GR_CHECK(hipMemcpyAsync(&missing, d_missing, sizeof(int), hipMemcpyDeviceToHost, stream), "DeviceIsSymmetric read failed");
    GR_CHECK(hipStreamSynchronize(stream), "DeviceIsSymmetric sync failed");
    GR_CHECK(hipFree(d_froms), "DeviceIsSymmetric hipFree failed");
    GR_CHECK(hipFree(d_missing), "DeviceIsSymmetric hipFree failed");
    symmetric = missing == 0;
    return retval;
}

}  // namespace graphio
}  // namespace gunrock
