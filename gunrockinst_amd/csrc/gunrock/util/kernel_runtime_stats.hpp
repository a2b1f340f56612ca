// util/kernel_runtime_stats.hpp -- per-workgroup runtime stamps of instrumented operator launches.
//
// Role of the reference's KernelRuntimeStats (gunrock/util/kernel_runtime_stats.cuh:51-291): with INSTRUMENT, thread 0 of every
// CTA records clock() at entry and exit (:79-109); the host accumulates "avg CTA duty" = sum of CTA runtimes / (longest CTA
// runtime x grid size) over all launches (:226-279; reported by Enactor::GetStatistics, bfs_enactor.cuh:173-186).
// Here a launch gets 32 lines of three device words -- sum of workgroup runtimes, longest runtime, workgroups that reported --
// filled with one atomic each per workgroup from the constant-rate wall clock (s_memrealtime, 100 MHz).  32 lines, 128 bytes
// apart, because atomics on ONE address retire at ~88 per microsecond: a single triple would add ~50 us to a 1500-workgroup
// launch and spoil the very times the instrumented enactor reports.  The enactor reads all launches' lines back once, at the
// end of Enact.  A null slot (every non-instrumented enactor) costs one scalar compare.
#pragma once

#include <hip/hip_runtime.h>

namespace gunrock {
namespace util {

constexpr int kDutyLines = 32;
constexpr int kDutyLineWords = 16;                         // 128 bytes
constexpr int kDutyWords = kDutyLines * kDutyLineWords;    // per launch

// Stamps the workgroup's lifetime: construct at kernel entry; the destructor (any exit path) reports.
struct DutyStamp {
    unsigned long long *slot;
    unsigned long long t0;
    __device__ __forceinline__ explicit DutyStamp(unsigned long long *s)
        : slot(s ? s + (blockIdx.x & (kDutyLines - 1)) * kDutyLineWords : nullptr), t0(s ? wall_clock64() : 0ull)
    {
    }
    __device__ __forceinline__ ~DutyStamp()
    {
        if (slot && threadIdx.x == 0) {
            const unsigned long long dt = wall_clock64() - t0;
            atomicAdd(slot, dt);
            atomicMax(slot + 1, dt);
            atomicAdd(slot + 2, 1ull);
        }
    }
};

}  // namespace util
}  // namespace gunrock
