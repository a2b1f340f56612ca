// Micro-benchmark: cost of N scattered one-byte stores into a 16 MiB flag array followed by a full scan of the array,
//  (a) destinations uniform over the whole array from every workgroup;
//  (b) every workgroup stores only into the 2 MiB slice that belongs to the XCD it runs on (HW_REG_XCC_ID).
// Build: hipcc -O3 --offload-arch=gfx950 tools/xcd_store_bench.hip -o /tmp/xcd_store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); std::exit(1); } } while (0)

__device__ __forceinline__ unsigned XccId() { return __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11)) & 7u; }

__device__ __forceinline__ unsigned Mix(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return static_cast<unsigned>(x ^ (x >> 31));
}

template <int MODE>  // 0 uniform, 1 slice by XCC id, 2 slice by blockIdx % 8
__global__ void StoreKernel(unsigned char *flags, unsigned n_log2, long long stores, unsigned *xcc_hist)
{
    const unsigned xcc = XccId();
    if (threadIdx.x == 0 && xcc_hist) atomicAdd(xcc_hist + xcc * 8 + (blockIdx.x & 7), 1u);
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < stores; i += stride) {
        unsigned d = Mix(i) & ((1u << n_log2) - 1);
        if (MODE == 1) d = (d >> 3) | (xcc << (n_log2 - 3));
        if (MODE == 2) d = (d >> 3) | ((blockIdx.x & 7u) << (n_log2 - 3));
        flags[d] = 1;
    }
}

__global__ void ScanKernel(const uint4 *flags, long long n16, unsigned long long *out)
{
    unsigned count = 0;
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 f = flags[i];
        count += __popc(f.x) + __popc(f.y) + __popc(f.z) + __popc(f.w);
    }
    for (int o = 32; o; o >>= 1) count += __shfl_xor(count, o, 64);
    if ((threadIdx.x & 63) == 0 && count) atomicAdd(out, static_cast<unsigned long long>(count));
}

int main()
{
    const unsigned n_log2 = 24;
    const long long n = 1ll << n_log2;
    unsigned char *flags; unsigned long long *out; unsigned *hist;
    CK(hipMalloc(&flags, n)); CK(hipMalloc(&out, 8)); CK(hipMalloc(&hist, 64 * 4));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    for (long long stores : {1000000ll, 3400000ll, 13000000ll}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best_a = 1e9f, best_b = 1e9f;
            unsigned long long h_out = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemset(flags, 0, n)); CK(hipMemset(out, 0, 8)); CK(hipMemset(hist, 0, 256)); CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(StoreKernel<0>, dim3(2048), dim3(256), 0, 0, flags, n_log2, stores, hist);
                if (mode == 1) hipLaunchKernelGGL(StoreKernel<1>, dim3(2048), dim3(256), 0, 0, flags, n_log2, stores, hist);
                if (mode == 2) hipLaunchKernelGGL(StoreKernel<2>, dim3(2048), dim3(256), 0, 0, flags, n_log2, stores, hist);
                CK(hipEventRecord(e1));
                hipLaunchKernelGGL(ScanKernel, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const uint4 *>(flags), n / 16, out);
                CK(hipEventRecord(e2)); CK(hipDeviceSynchronize());
                float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
                if (a + b < best_a + best_b) { best_a = a; best_b = b; }
                CK(hipMemcpy(&h_out, out, 8, hipMemcpyDeviceToHost));
            }
            std::printf("stores %9lld mode %d: store kernel %7.1f us, scan kernel %7.1f us, total %7.1f us (distinct %llu)\n", stores, mode,
                        best_a * 1e3, best_b * 1e3, (best_a + best_b) * 1e3, h_out);
        }
    }
    unsigned h[64]; CK(hipMemcpy(h, hist, 256, hipMemcpyDeviceToHost));
    std::printf("workgroups by (XCC id row, blockIdx %% 8 column):\n");
    for (int x = 0; x < 8; ++x) { for (int b = 0; b < 8; ++b) std::printf("%5u", h[x * 8 + b]); std::printf("\n"); }
    return 0;
}
