// device_intrinsics.hpp -- wave64 building blocks for gfx950 (CDNA4).
//
// Replaces the reference's warp-32 / inline-PTX helpers (gunrock/util/device_intrinsics.cuh:24-92,
// util/scan/warp_scan.cuh, util/scan/cooperative_scan.cuh:152-173) with native wave64 forms:
// 64-bit ballots, v_mbcnt lane ranks, shuffle-based wave scans and a workgroup scan that
// combines wave totals through LDS.
#pragma once

#include <hip/hip_runtime.h>

#include <map>

namespace gunrock {
namespace util {

constexpr int kWaveSize = 64;

__device__ __forceinline__ unsigned LaneId()
{
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// Number of set bits of `mask` strictly below the calling lane.
__device__ __forceinline__ unsigned RankInMask(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), 0u));
}

// Workgroup barrier for LDS hand-offs only: waits for this wave's LDS operations, NOT for its outstanding global stores.
// __syncthreads() drains vmcnt too (s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier), i.e. a wave that has just issued scattered
// global stores sits at the barrier for their full round trip (~3 us under load on MI355X) although nobody in the workgroup
// will read them.  Use only where the barrier orders LDS traffic (and register-consumed loads) and nothing else.
__device__ __forceinline__ void LdsBarrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Inclusive prefix sum across the 64 lanes of a wave (all lanes must call).
template <typename T>
__device__ __forceinline__ T WaveInclusiveSum(T x)
{
    const unsigned lane = LaneId();
#pragma unroll
    for (int d = 1; d < kWaveSize; d <<= 1) {
        T y = __shfl_up(x, d, kWaveSize);
        if (lane >= static_cast<unsigned>(d)) x += y;
    }
    return x;
}

// ---- wave64 scans on the DPP data path (no LDS crossbar traffic, ~6 VALU instructions) ----
// row_shr:1,2,4,8 scan the four 16-lane rows, row_bcast:15 / row_bcast:31 carry the row totals across (gfx9 DPP controls;
// lanes without a source keep `identity`).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned DppFrom(unsigned identity, unsigned x)
{
    return static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(identity), static_cast<int>(x), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ unsigned WaveInclusiveSumDpp(unsigned x)
{
    x += DppFrom<0x111, 0xF>(0u, x);
    x += DppFrom<0x112, 0xF>(0u, x);
    x += DppFrom<0x114, 0xF>(0u, x);
    x += DppFrom<0x118, 0xF>(0u, x);
    x += DppFrom<0x142, 0xA>(0u, x);
    x += DppFrom<0x143, 0xC>(0u, x);
    return x;
}
__device__ __forceinline__ unsigned WaveInclusiveMaxDpp(unsigned x)
{
    x = max(x, DppFrom<0x111, 0xF>(0u, x));
    x = max(x, DppFrom<0x112, 0xF>(0u, x));
    x = max(x, DppFrom<0x114, 0xF>(0u, x));
    x = max(x, DppFrom<0x118, 0xF>(0u, x));
    x = max(x, DppFrom<0x142, 0xA>(0u, x));
    x = max(x, DppFrom<0x143, 0xC>(0u, x));
    return x;
}

// DPP move of a 32- or 64-bit value (lanes without a source get `identity`)
template <int CTRL, int ROW_MASK, typename T>
__device__ __forceinline__ T DppMove(T identity, T x)
{
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- or 64-bit values");
    if constexpr (sizeof(T) == 4) {
        unsigned xi, ii;
        __builtin_memcpy(&xi, &x, 4);
        __builtin_memcpy(&ii, &identity, 4);
        const unsigned r = DppFrom<CTRL, ROW_MASK>(ii, xi);
        T out;
        __builtin_memcpy(&out, &r, 4);
        return out;
    } else {
        unsigned long long xi, ii;
        __builtin_memcpy(&xi, &x, 8);
        __builtin_memcpy(&ii, &identity, 8);
        const unsigned lo = DppFrom<CTRL, ROW_MASK>(static_cast<unsigned>(ii), static_cast<unsigned>(xi));
        const unsigned hi = DppFrom<CTRL, ROW_MASK>(static_cast<unsigned>(ii >> 32), static_cast<unsigned>(xi >> 32));
        const unsigned long long r = (static_cast<unsigned long long>(hi) << 32) | lo;
        T out;
        __builtin_memcpy(&out, &r, 8);
        return out;
    }
}

// Inclusive SEGMENTED scan across the wave on the DPP path: `head` marks the first lane of a segment; lane i ends up with
// Combine over its segment's lanes up to i.  The pair operator (f_a, v_a) + (f_b, v_b) = (f_a | f_b, f_b ? v_b : Combine(v_a, v_b))
// is associative, so the usual scan network (row_shr 1, 2, 4, 8; row_bcast 15, 31) applies.  Six steps of three or four VALU
// instructions; the shuffle form (two ds_bpermute per step, each a ~100-cycle LDS-pipe round trip) dominated the reducing advance.
template <typename Ops, typename T>
__device__ __forceinline__ T WaveSegmentedScanDpp(bool head, T v)
{
    unsigned f = head ? 1u : 0u;
    const T id = Ops::Identity();
#define GRX_SEG_STEP(CTRL, MASK)                                   \
    {                                                              \
        const unsigned fo = DppFrom<CTRL, MASK>(0u, f);            \
        const T vo = DppMove<CTRL, MASK, T>(id, v);                \
        v = f ? v : Ops::Combine(vo, v);                           \
        f |= fo;                                                   \
    }
    GRX_SEG_STEP(0x111, 0xF)
    GRX_SEG_STEP(0x112, 0xF)
    GRX_SEG_STEP(0x114, 0xF)
    GRX_SEG_STEP(0x118, 0xF)
    GRX_SEG_STEP(0x142, 0xA)
    GRX_SEG_STEP(0x143, 0xC)
#undef GRX_SEG_STEP
    return v;
}

template <typename T>
__device__ __forceinline__ T WaveSum(T x)
{
#pragma unroll
    for (int d = kWaveSize / 2; d >= 1; d >>= 1) x += __shfl_xor(x, d, kWaveSize);
    return x;
}

// Workgroup-wide exclusive prefix sum; every thread of the THREADS-sized block must call.
// `total` receives the block total in every thread.  Two barriers per call.
template <int THREADS, typename T>
struct BlockScan {
    static constexpr int WAVES = THREADS / kWaveSize;
    struct Storage {
        T wave_total[WAVES];
    };

    static __device__ __forceinline__ T ExclusiveSum(T x, T &total, Storage &st)
    {
        const unsigned lane = LaneId();
        const unsigned wave = threadIdx.x / kWaveSize;
        T incl = WaveInclusiveSum(x);
        if (lane == kWaveSize - 1) st.wave_total[wave] = incl;
        __syncthreads();
        T base = 0, sum = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            T t = st.wave_total[w];
            if (static_cast<unsigned>(w) < wave) base += t;
            sum += t;
        }
        __syncthreads();  // storage may be reused right away
        total = sum;
        return base + incl - x;
    }
};

// Agent-scope relaxed accessors for words other workgroups update inside the same launch
// (per-XCD L2s are not coherent; plain loads may be served from a stale L1/L2 line).
__device__ __forceinline__ unsigned LoadAgent(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int LoadAgent(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Workgroups of `kernel` that fit on the device at once (occupancy API x CU count), cached per kernel instantiation.
// Grid-stride kernels launched with exactly this many workgroups run as ONE wave of blocks: a grid of
// CUs x "hoped-for blocks per CU" that exceeds residency leaves a second, mostly empty round (a 1.3x tail at 6 of 8).
template <typename Kernel>
inline int ResidentGrid(Kernel kernel, int threads)
{
    static std::map<const void *, int> cache;  // keyed by the kernel's address (the template is keyed by its TYPE only)
    int &cached = cache[reinterpret_cast<const void *>(kernel)];
    if (cached == 0) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        int cus = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
        cached = per_cu * cus;
    }
    return cached;
}

}  // namespace util
}  // namespace gunrock
