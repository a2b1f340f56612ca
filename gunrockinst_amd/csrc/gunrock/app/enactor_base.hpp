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

    EnactorBase(FrontierType ft, bool debug) : frontier_type(ft), DEBUG(debug) {}

    virtual ~EnactorBase()
    {
        work_progress.Release();
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
