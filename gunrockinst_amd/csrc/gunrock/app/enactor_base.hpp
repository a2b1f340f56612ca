// app/enactor_base.hpp -- host-side state shared by all enactors.
//
// Roles of the reference's EnactorBase / EnactorStats / FrontierAttribute
// (gunrock/app/enactor_base.cuh:36-68, 136-189): grid sizing, the work-progress counters and the
// per-run statistics an enactor reports through GetStatistics().  Grid sizing is for MI355X: 256 CUs,
// and operators that loop over tiles are launched with CUs x blocks-per-CU workgroups
// (enactor_base.cuh:147,182-188 used SMs x occupancy).
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

#include <gunrock/app/problem_base.hpp>
#include <gunrock/util/error_utils.hpp>
#include <gunrock/util/frontier.hpp>
#include <gunrock/util/kernel_runtime_stats.hpp>

namespace gunrock {
namespace app {

struct EnactorStats {
    long long iteration = 0;
    long long total_queued = 0;     // vertices dequeued over the whole run
    long long total_edges_queued = 0;
    int advance_grid_size = 0;
    int filter_grid_size = 0;
    double total_runtimes = 0;      // per-workgroup clock sums when INSTRUMENT (KernelRuntimeStats role)
    double total_lifetimes = 0;
    long long kernel_launches = 0;  // INSTRUMENT: operator kernels launched by the last Enact
    double kernel_ms = 0;           // INSTRUMENT: their summed HIP-event durations
    hipError_t retval = hipSuccess;
    // INSTRUMENT: one record per BSP iteration of the last Enact
    struct LevelRecord { long long frontier; long long edges; double ms; int kind; };
    std::vector<LevelRecord> levels;
};

template <typename SizeT, typename VertexId>
struct FrontierAttribute {
    SizeT queue_length = 0;
    SizeT queue_edges = 0;
    int selector = 0;
    int queue_index = 0;
    bool queue_reset = false;
};

class EnactorBase {
   protected:
    int cu_count = 256;
    FrontierType frontier_type;
    EnactorStats enactor_stats;
    util::WorkProgress work_progress;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;  // INSTRUMENT: brackets one operator launch
    // INSTRUMENT: runtime-stamp words of the operator launches of one Enact (KernelRuntimeStats role)
    static constexpr int kDutyLaunches = 512;
    unsigned long long *d_duty = nullptr;
    int duty_used = 0;

    EnactorBase(FrontierType ft, bool debug) : frontier_type(ft), DEBUG(debug) {}

    virtual ~EnactorBase()
    {
        work_progress.Release();
        if (d_duty) util::GRError(hipFree(d_duty), "EnactorBase hipFree failed", __FILE__, __LINE__);
        if (ev_begin) util::GRError(hipEventDestroy(ev_begin), "EnactorBase hipEventDestroy failed", __FILE__, __LINE__);
        if (ev_end) util::GRError(hipEventDestroy(ev_end), "EnactorBase hipEventDestroy failed", __FILE__, __LINE__);
    }

    // INSTRUMENT support: the reference samples clock() per CTA (util/kernel_runtime_stats.cuh:79-109);
    // here an operator launch is bracketed by two HIP events on the launch stream.
    hipError_t InstrumentBegin(hipStream_t stream)
    {
        hipError_t retval = hipSuccess;
        if (!ev_begin) {
            GR_CHECK(hipEventCreate(&ev_begin), "EnactorBase hipEventCreate failed");
            GR_CHECK(hipEventCreate(&ev_end), "EnactorBase hipEventCreate failed");
        }
        return util::GRError(hipEventRecord(ev_begin, stream), "EnactorBase hipEventRecord failed", __FILE__, __LINE__);
    }
    hipError_t InstrumentEnd(hipStream_t stream)
    {
        return util::GRError(hipEventRecord(ev_end, stream), "EnactorBase hipEventRecord failed", __FILE__, __LINE__);
    }
    // call after the stream has been synchronised
    // start of an instrumented Enact: clear the stamp words
    hipError_t DutyBegin(hipStream_t stream)
    {
        hipError_t retval = hipSuccess;
        if (!d_duty) GR_CHECK(hipMalloc(&d_duty, sizeof(unsigned long long) * util::kDutyWords * kDutyLaunches), "EnactorBase hipMalloc failed");
        duty_used = 0;
        enactor_stats.total_runtimes = 0;
        enactor_stats.total_lifetimes = 0;
        return util::GRError(hipMemsetAsync(d_duty, 0, sizeof(unsigned long long) * util::kDutyWords * kDutyLaunches, stream),
                             "EnactorBase memset failed", __FILE__, __LINE__);
    }
    // stamp words for the next operator launch (nullptr once the table is full: that launch goes unmeasured)
    unsigned long long *DutySlot()
    {
        if (!d_duty || duty_used >= kDutyLaunches) return nullptr;
        return d_duty + static_cast<size_t>(util::kDutyWords) * duty_used++;
    }
    // end of Enact: total_runtimes = sum of workgroup runtimes, total_lifetimes = sum over launches of (longest runtime x
    // workgroups), so that total_runtimes / total_lifetimes is the reference's "avg CTA duty" (kernel_runtime_stats.cuh:226-279)
    hipError_t DutyCollect(hipStream_t stream)
    {
        hipError_t retval = hipSuccess;
        if (!d_duty || duty_used == 0) return retval;
        std::vector<unsigned long long> h(static_cast<size_t>(util::kDutyWords) * duty_used);
        GR_CHECK(hipMemcpyAsync(h.data(), d_duty, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, stream), "EnactorBase copy failed");
        GR_CHECK(hipStreamSynchronize(stream), "EnactorBase sync failed");
        for (int i = 0; i < duty_used; ++i) {
            double sum = 0, longest = 0, groups = 0;
            for (int l = 0; l < util::kDutyLines; ++l) {
                const unsigned long long *w = h.data() + static_cast<size_t>(util::kDutyWords) * i + static_cast<size_t>(util::kDutyLineWords) * l;
                sum += static_cast<double>(w[0]);
                if (static_cast<double>(w[1]) > longest) longest = static_cast<double>(w[1]);
                groups += static_cast<double>(w[2]);
            }
            enactor_stats.total_runtimes += sum;
            enactor_stats.total_lifetimes += longest * groups;
        }
        return retval;
    }
    void InstrumentCollect(long long frontier = 0, long long edges = 0, int kind = 0)
    {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev_begin, ev_end) == hipSuccess) {
            enactor_stats.kernel_ms += ms;
            enactor_stats.kernel_launches += 1;
            enactor_stats.levels.push_back({frontier, edges, ms, kind});
        }
    }

    hipError_t Setup(int max_grid_size, int advance_blocks_per_cu, int filter_blocks_per_cu)
    {
        hipError_t retval = hipSuccess;
        int dev = 0;
        hipDeviceProp_t prop;
        GR_CHECK(hipGetDevice(&dev), "EnactorBase hipGetDevice failed");
        GR_CHECK(hipGetDeviceProperties(&prop, dev), "EnactorBase hipGetDeviceProperties failed");
        cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        enactor_stats.advance_grid_size = max_grid_size > 0 ? max_grid_size : cu_count * advance_blocks_per_cu;
        enactor_stats.filter_grid_size = max_grid_size > 0 ? max_grid_size : cu_count * filter_blocks_per_cu;
        if ((retval = work_progress.Init())) return retval;
        enactor_stats.iteration = 0;
        enactor_stats.total_queued = 0;
        enactor_stats.total_edges_queued = 0;
        enactor_stats.kernel_launches = 0;
        enactor_stats.kernel_ms = 0;
        enactor_stats.levels.clear();
        return retval;
    }

   public:
    bool DEBUG;
    FrontierType GetFrontierType() { return frontier_type; }
};

}  // namespace app
}  // namespace gunrock
