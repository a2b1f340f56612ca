// priority_queue/kernel.hpp -- near/far split of a vertex queue by priority bucket (delta-stepping).
//
// Role of the reference's priority_queue::Bisect and its kernels MarkVisit / MarkNF / Compact
// (gunrock/priority_queue/kernel.cuh:114-234, 404-450; near_far_pile.cuh:38-141): remove duplicates from the
// advance output, send vertices whose bucket (Functor::ComputePriorityScore) is within the current level to the
// NEAR queue -- the next advance frontier -- and park the rest in the FAR pile.
// The reference needs 3 kernels, 2 device-wide moderngpu scans and 2 blocking reads per call, and keeps four
// (m+1)-int scratch arrays.  Here it is ONE kernel:
//   * de-duplication: atomicExch of a per-call tag into d_visit_lookup[v]; the first arrival keeps the vertex
//     (the reference keeps the LAST index written, kernel.cuh:114-132 -- which copy survives is irrelevant);
//   * far-pile entries carry the distance they were parked with; when the pile is re-split at a later level,
//     entries whose vertex has improved since are stale and dropped (the reference re-expands them);
//   * near vertices go through FrontierWriter (degree prefix for the next load-balanced advance), far
//     vertices are appended with one global atomic per tile.
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/oprtr/frontier_writer.hpp>
#include <gunrock/util/device_intrinsics.hpp>
#include <gunrock/util/error_utils.hpp>
#include <gunrock/util/frontier.hpp>

namespace gunrock {
namespace priority_queue {

template <typename VertexId, typename SizeT>
struct BisectArgs {
    const VertexId *d_in;        // queue to split
    const unsigned *d_in_dist;   // distance at parking time (far pile input) or NULL (advance output)
    SizeT num_elements;
    // when set: the element count is the low word of this packed device tail (the advance that produced d_in wrote it; the
    // split then follows the advance on the stream without a host round trip in between) and num_elements is ignored
    const unsigned long long *d_num_elements = nullptr;
    unsigned level;              // buckets <= level are near
    int tag;                     // unique per call, for de-duplication
    util::Frontier<VertexId, SizeT> near;   // output frontier
    unsigned long long *d_near_tail;
    VertexId *d_far_v;           // far pile output (appended at *d_far_tail)
    unsigned *d_far_d;
    SizeT far_capacity;
    unsigned long long *d_far_tail;
    unsigned *d_far_min;         // smallest bucket parked by this call (atomicMin; caller presets UINT_MAX)
    int *d_overflow;
    const SizeT *d_row_offsets;
};

template <int THREADS, int ITEMS, typename ProblemData, typename Functor>
__global__ __launch_bounds__(THREADS) void BisectKernel(BisectArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> a,
                                                        typename ProblemData::DataSlice slice)
{
    typedef typename ProblemData::VertexId VertexId;
    typedef typename ProblemData::SizeT SizeT;
    constexpr int TILE = THREADS * ITEMS;
    typedef oprtr::FrontierWriter<THREADS, 2 * TILE, VertexId, SizeT> Writer;
    typedef util::BlockScan<THREADS, int> Scan;
    __shared__ typename Writer::Storage s_writer;
    __shared__ typename Scan::Storage s_scan;
    __shared__ unsigned long long s_far_base;

    Writer::Init(s_writer);
    __syncthreads();

    unsigned far_min = 0xFFFFFFFFu;
    const long long num_elements = a.d_num_elements ? static_cast<long long>(util::TailCount(*a.d_num_elements)) : static_cast<long long>(a.num_elements);
    const long long tiles = (num_elements + TILE - 1) / TILE;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int pending = Writer::Count(s_writer);
        __syncthreads();
        if (pending > TILE) Writer::template Flush<true>(s_writer, pending, a.near, a.d_near_tail, a.d_overflow, a.d_row_offsets);

        VertexId v[ITEMS];
        unsigned dist[ITEMS];
        int kind[ITEMS];  // 0 drop, 1 near, 2 far
        int near_mine = 0, far_mine = 0;
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const long long i = tile * TILE + k * THREADS + threadIdx.x;
            v[k] = -1;
            if (i < num_elements) v[k] = a.d_in[i];
        }
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            kind[k] = 0;
            if (v[k] < 0) continue;
            dist[k] = slice.Distance(v[k]);
            if (a.d_in_dist) {  // far-pile entry: stale if the vertex improved after it was parked
                const long long i = tile * TILE + k * THREADS + threadIdx.x;
                if (a.d_in_dist[i] != dist[k]) continue;
            }
            if (atomicExch(slice.d_visit_lookup + v[k], a.tag) == a.tag) continue;  // duplicate within this call
            const unsigned bucket = Functor::ComputePriorityScore(v[k], &slice);
            kind[k] = bucket <= a.level ? 1 : 2;
            if (kind[k] == 2 && bucket < far_min) far_min = bucket;
            near_mine += kind[k] == 1;
            far_mine += kind[k] == 2;
        }
        int pos = Writer::Reserve(s_writer, near_mine);
#pragma unroll
        for (int k = 0; k < ITEMS; ++k)
            if (kind[k] == 1) s_writer.buf[pos++] = v[k];

        int far_total;
        const int far_rank = Scan::ExclusiveSum(far_mine, far_total, s_scan);
        if (threadIdx.x == 0 && far_total > 0) s_far_base = atomicAdd(a.d_far_tail, static_cast<unsigned long long>(far_total));
        __syncthreads();
        if (far_total > 0) {
            const unsigned long long base = s_far_base;
            if (base + far_total > static_cast<unsigned long long>(a.far_capacity)) {
                if (threadIdx.x == 0) *a.d_overflow = 1;
            } else {
                unsigned long long at = base + far_rank;
#pragma unroll
                for (int k = 0; k < ITEMS; ++k)
                    if (kind[k] == 2) {
                        a.d_far_v[at] = v[k];
                        a.d_far_d[at] = dist[k];
                        ++at;
                    }
            }
        }
        __syncthreads();
    }
    const int rest = Writer::Count(s_writer);
    __syncthreads();
    Writer::template Flush<true>(s_writer, rest, a.near, a.d_near_tail, a.d_overflow, a.d_row_offsets);
#pragma unroll
    for (int d = util::kWaveSize / 2; d >= 1; d >>= 1) {
        const unsigned other = __shfl_xor(far_min, d, util::kWaveSize);
        far_min = other < far_min ? other : far_min;
    }
    if (util::LaneId() == 0 && far_min != 0xFFFFFFFFu) atomicMin(a.d_far_min, far_min);
}

template <int THREADS, int ITEMS, typename ProblemData, typename Functor>
hipError_t Bisect(const BisectArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> &args,
                  const typename ProblemData::DataSlice &slice, int max_grid_size, hipStream_t stream)
{
    if (!args.d_num_elements && args.num_elements <= 0) return hipSuccess;
    // (count on the device: num_elements is then the caller's upper bound for sizing the grid)
    const long long tiles = (static_cast<long long>(args.num_elements) + THREADS * ITEMS - 1) / (THREADS * ITEMS);
    long long grid = tiles < max_grid_size ? tiles : max_grid_size;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((BisectKernel<THREADS, ITEMS, ProblemData, Functor>), dim3(static_cast<unsigned>(grid)), dim3(THREADS), 0, stream, args,
                       slice);
    return util::GRError("priority_queue::BisectKernel launch failed", __FILE__, __LINE__);
}

}  // namespace priority_queue
}  // namespace gunrock
