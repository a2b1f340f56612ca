// Microbenchmark: scattered updates of a 2 MiB bitmap (the BFS status structure) by every CU.
//   mode 0: atomicOr agent scope, result used     mode 1: agent scope, fire-and-forget
//   mode 2: atomicOr WORKGROUP scope, fire-and-forget, one private bitmap per XCD (HW_REG_XCC_ID)
//   mode 3: plain byte stores into a 16 MiB byte map          mode 4: plain byte stores, per-XCD 2 MiB region (aliasing test)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ inline unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf; }  // HW_REG_XCC_ID

template <int MODE>
__global__ void k(unsigned *bits, unsigned char *bytes, unsigned *sink, int words, long long n, unsigned seed)
{
    unsigned acc = 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned *mine = bits;
    if (MODE == 2) mine = bits + (size_t)xcc_id() * words;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        const unsigned v = x & (words * 32 - 1);
        if (MODE == 0) acc += atomicOr(bits + (v >> 5), 1u << (v & 31));
        if (MODE == 1) atomicOr(bits + (v >> 5), 1u << (v & 31));
        if (MODE == 2) __hip_atomic_fetch_or(mine + (v >> 5), 1u << (v & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 3) bytes[v] = 1;
        if (MODE == 4) bytes[(size_t)xcc_id() * (words * 4) + (v >> 3)] = 1;
    }
    if (MODE == 0 && acc == 0x12345) *sink = acc;
}

int main()
{
    const int words = 1 << 19;  // 2 MiB bitmap = 16.7 M bits
    const long long n = 8 << 20;
    unsigned *bits, *sink;
    unsigned char *bytes;
    hipMalloc(&bits, (size_t)words * 4 * 16);
    hipMalloc(&bytes, (size_t)words * 32 + (1 << 26));
    hipMalloc(&sink, 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipMemset(bits, 0, (size_t)words * 4 * 16);
            hipDeviceSynchronize();
            hipEventRecord(a);
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, bits, bytes, sink, words, n, rep); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, bits, bytes, sink, words, n, rep); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, bits, bytes, sink, words, n, rep); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(2048), dim3(256), 0, 0, bits, bytes, sink, words, n, rep); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(2048), dim3(256), 0, 0, bits, bytes, sink, words, n, rep); break;
            }
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        // verify mode 2: OR of the 8 copies must have ~ n distinct bits set
        long long setbits = -1;
        if (mode == 2 || mode == 1) {
            std::vector<unsigned> h((size_t)words * 16);
            hipMemcpy(h.data(), bits, h.size() * 4, hipMemcpyDeviceToHost);
            setbits = 0;
            for (int w = 0; w < words; ++w) { unsigned o = 0; for (int c = 0; c < 16; ++c) o |= h[(size_t)c * words + w]; setbits += __builtin_popcount(o); }
        }
        printf("mode %d: %.3f ms  -> %.2f G updates/s  setbits %lld\n", mode, best, n / best / 1e6, setbits);
    }
    return 0;
}
