// memset_kernel.hpp -- fill / iota kernels.
//
// Same role as the reference's util::MemsetKernel / MemsetIdxKernel
// (gunrock/util/memset_kernel.cuh:43-67), but launched with enough workgroups to fill 256 CUs
// instead of the reference's fixed <<<128,128>>> grid (bfs_problem.cuh:298-316) and with 16-byte
// stores on the aligned body.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace gunrock {
namespace util {

template <typename T>
__global__ void MemsetKernel(T *d_out, T value, long long length)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if constexpr (sizeof(T) == 4) {
        // vector body: the base pointer comes from hipMalloc (256-B aligned)
        using V4 = __attribute__((ext_vector_type(4))) unsigned;
        unsigned bits = __builtin_bit_cast(unsigned, value);
        V4 v = {bits, bits, bits, bits};
        const long long nvec = length / 4;
        V4 *out4 = reinterpret_cast<V4 *>(d_out);
        for (long long k = i; k < nvec; k += stride) out4[k] = v;
        for (long long k = nvec * 4 + i; k < length; k += stride) d_out[k] = value;
    } else {
        for (; i < length; i += stride) d_out[i] = value;
    }
}

template <typename T>
__global__ void MemsetIdxKernel(T *d_out, long long length)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < length; i += stride)
        d_out[i] = static_cast<T>(i);
}

inline int MemsetGrid(long long length, int threads = 256)
{
    long long blocks = (length / 4 + threads - 1) / threads;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;  // 256 CUs x 8 workgroups, grid-stride the rest
    return static_cast<int>(blocks);
}

template <typename T>
inline void Memset(T *d_out, T value, long long length, hipStream_t stream = 0)
{
    if (length <= 0) return;
    hipLaunchKernelGGL(MemsetKernel<T>, dim3(MemsetGrid(length)), dim3(256), 0, stream, d_out, value, length);
}

template <typename T>
inline void MemsetIdx(T *d_out, long long length, hipStream_t stream = 0)
{
    if (length <= 0) return;
    hipLaunchKernelGGL(MemsetIdxKernel<T>, dim3(MemsetGrid(length * 4)), dim3(256), 0, stream, d_out, length);
}

}  // namespace util
}  // namespace gunrock
