// oprtr/advance/sweep_chain.hpp -- the direction rules of a bottom-up level as one host/device definition, and the
// bookkeeping of a CHAIN of sweeps that apply them on the device (bottom_up.hpp BottomUpAutoKernel; kernel.hpp
// ChainedPersistentLevelsKernel; enactor: app/bfs/bfs_enactor.hpp run_sweep_chain).
//
// Reference: the host-side switches of app/dobfs/dobfs_enactor.cuh:397,569 (one blocking read-back per level).
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/util/device_intrinsics.hpp>
#include <gunrock/util/frontier.hpp>

namespace gunrock {
namespace oprtr {
namespace advance {

// ---- chained sweeps: the direction rules of a bottom-up level, evaluated on the device ----
// A bottom-up level ends with a host round trip (publish kernel, PCIe write, host spin, next launch: ~12 us) only so that the
// host can apply four rules to the number of vertices the level found: stop (none), return to top-down (few), dense or
// compacting sweep, and whether the compacting sweep also emits a queue.  BottomUpAutoKernel applies the same rules itself:
// every sweep of a CHAIN adds its finds to its own set of wide counters, sweep k reads the sets of the sweeps before it (they
// are complete: kernel boundary) and replays their decisions and its own.  The host queues several sweeps back to back and
// makes ONE round trip for all of them; a sweep that finds the chain already over (stop / switch) exits at once (~3 us), and
// the host replays the same rules on the published sums (SweepRule is the one definition of them) to learn what ran.
enum SweepAction { kSweepLeft = -1, kSweepStop = 0, kSweepSwitch = 1, kSweepDense = 2, kSweepSparse = 3, kSweepSparseEmit = 4 };

struct SweepRule {
    long long with_in_edges = 0, nodes = 0;
    double beta = 0, emit_factor = 0;
    int sparse_div = 0;
    // in_count: vertices of the level's input frontier; total: vertices queued so far, this frontier included
    __host__ __device__ __forceinline__ int Decide(long long in_count, long long total, bool may_switch) const
    {
        if (in_count == 0) return kSweepStop;
        if (may_switch && static_cast<double>(in_count) * beta < static_cast<double>(nodes)) return kSweepSwitch;
        const long long open = with_in_edges - total;
        if (sparse_div > 0 && open * sparse_div <= nodes)
            return (static_cast<double>(in_count) * beta < emit_factor * static_cast<double>(nodes)) ? kSweepSparseEmit : kSweepSparse;
        return kSweepDense;
    }
};

constexpr int kChainMax = 6;        // sweeps per chain at most
constexpr int kWideSetWords = 32 * 16;  // one set of wide counters (util::WorkProgress: 32 lines, 128 bytes apart)

struct SweepChain {
    const unsigned long long *d_sets = nullptr;  // set j: finds of the chain's sweep j; set 0: of whatever produced the first frontier
    long long base_total = 0;   // vertices queued before the chain's first frontier
    long long first_in = -1;    // size of the first frontier when the host knows it, else -1: the sum of set 0
    SweepRule rule;
    int index = 1;              // this sweep's position in the chain, from 1
    int first_may_switch = 0;   // 0: the first sweep runs whatever the switch-back rule says (the host has just turned bottom-up)
    int *d_log = nullptr;       // [kChainMax + 1]: the action every sweep took
};

// Whole wave; the result is wave-uniform and the same in every wave of the grid.
__device__ __forceinline__ int ChainAction(const SweepChain &c, unsigned lane)
{
    unsigned long long w[kChainMax];
#pragma unroll
    for (int j = 0; j < kChainMax; ++j)  // (all sets the decision needs, in flight together)
        w[j] = (j < c.index && lane < 32) ? c.d_sets[static_cast<size_t>(j) * kWideSetWords + lane * 16] : 0ull;
    long long total = c.base_total;
    int action = kSweepStop;
#pragma unroll
    for (int j = 1; j <= kChainMax; ++j) {
        if (j <= c.index) {
            const long long in_j = (j == 1 && c.first_in >= 0) ? c.first_in : static_cast<long long>(util::TailCount(util::WaveSum(w[j - 1])));
            total += in_j;
            action = c.rule.Decide(in_j, total, j > 1 || c.first_may_switch != 0);
            if (action <= kSweepSwitch && j < c.index) return kSweepLeft;  // an earlier sweep already ended the chain
            if (action <= kSweepSwitch) return action;
        }
    }
    return action;
}


// How a chain of `sweeps` queued sweeps ended: end = index (from 1) of the first sweep that did NOT run -- its action is
// `action` (kSweepStop / kSweepSwitch), the action of the sweep before it `prev_action` -- or sweeps + 1 when all of them ran
// (then `action` is what a further sweep would do: the rules applied to the frontier the last one produced).
struct ChainEnd {
    int end = 1;
    int action = kSweepStop;
    int prev_action = kSweepLeft;
    long long next_in = 0;  // size of the frontier the chain left
};

// Whole wave (device) -- needs the counter sets 0 .. sweeps.
__device__ __forceinline__ ChainEnd ChainOutcome(const SweepChain &c, int sweeps, unsigned lane)
{
    unsigned long long w[kChainMax + 1];
#pragma unroll
    for (int j = 0; j <= kChainMax; ++j)
        w[j] = (j <= sweeps && lane < 32) ? c.d_sets[static_cast<size_t>(j) * kWideSetWords + lane * 16] : 0ull;
    ChainEnd out;
    long long total = c.base_total;
#pragma unroll
    for (int j = 1; j <= kChainMax + 1; ++j) {
        if (j <= sweeps + 1) {
            const long long in_j = (j == 1 && c.first_in >= 0) ? c.first_in : static_cast<long long>(util::TailCount(util::WaveSum(w[j - 1])));
            total += in_j;
            const int action = c.rule.Decide(in_j, total, j > 1 || c.first_may_switch != 0);
            out.end = j;
            out.action = action;
            out.next_in = in_j;
            if (action <= kSweepSwitch || j == sweeps + 1) return out;
            out.prev_action = action;
        }
    }
    return out;
}

}  // namespace advance
}  // namespace oprtr
}  // namespace gunrock
