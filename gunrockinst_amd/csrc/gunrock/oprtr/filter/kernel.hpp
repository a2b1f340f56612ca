// oprtr/filter/kernel.hpp -- the FILTER (vertex-compact) operator for gfx950.
//
// Public surface kept from the reference (gunrock/oprtr/filter/kernel.cuh:738-792, kernel_policy.cuh:74-180):
// namespace gunrock::oprtr::filter, a KernelPolicy, and kernels that run the user functor's
// CondFilter / ApplyFilter over a queue of element ids:
//   * Kernel       -- compacting: elements that are -1 or fail CondFilter are dropped, survivors get
//                     ApplyFilter and are enqueued (reference filter::Kernel, cta.cuh:467-544);
//   * ApplyKernel  -- `filtering_flag = false`: ApplyFilter on every element, optional copy-through, no
//                     compaction (reference filter::Kernel2, kernel.cuh:302-383) -- what CC's hook / pointer-jump
//                     sweeps use.
// Implementation is new: tile load with ITEMS independent loads per lane, wave64 ballot ranks, LDS staging
// and one packed global atomic per flush through FrontierWriter (the reference reserves output with one
// atomicAdd per 512-element tile after a raking-grid CTA scan, scan/cooperative_scan.cuh:289-323).
// A NULL input queue means the identity queue 0..num_elements-1 (the reference materialises iota queues,
// cc_problem.cuh:386-408, and reads them back every sweep).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include <gunrock/oprtr/frontier_writer.hpp>
#include <gunrock/util/device_intrinsics.hpp>
#include <gunrock/util/error_utils.hpp>
#include <gunrock/util/frontier.hpp>

namespace gunrock {
namespace oprtr {
namespace filter {

// Optional split of CondFilter for functors whose test is a returning atomic (same idea as the advance's IssueEdge /
// ResolveEdge, oprtr/advance/functor_hooks.hpp): all elements of a thread are issued before any result is examined.
//   `static bool ScreenFilter(node, problem, value, nid)`   side-effect free, evaluated for every element (node may be any valid id)
//   `static T    IssueFilter(node, problem, value, nid)`     the atomic, result handed back unexamined
//   `static bool ResolveFilter(T, node, problem, value, nid)`
// CondFilter must stay equivalent to Screen && Resolve(Issue).
template <typename Functor, typename VertexId, typename DataSlice, typename Value, typename SizeT, typename = void>
struct HasIssueFilter : std::false_type {};
template <typename Functor, typename VertexId, typename DataSlice, typename Value, typename SizeT>
struct HasIssueFilter<Functor, VertexId, DataSlice, Value, SizeT,
                      std::void_t<decltype(Functor::IssueFilter(VertexId(), static_cast<DataSlice *>(nullptr), Value(), SizeT()))>> : std::true_type {};

template <int _THREADS, int _ITEMS_PER_THREAD, int _MIN_BLOCKS_PER_CU>
struct KernelPolicy {
    static constexpr int THREADS = _THREADS;
    static constexpr int ITEMS = _ITEMS_PER_THREAD;
    static constexpr int TILE = THREADS * ITEMS;
    static constexpr int MIN_BLOCKS = _MIN_BLOCKS_PER_CU;
    static constexpr int STAGE_CAPACITY = 2 * TILE;
};

template <typename VertexId, typename SizeT>
struct FilterArgs {
    const VertexId *d_in;          // NULL = identity queue
    SizeT num_elements;
    util::Frontier<VertexId, SizeT> out;   // WITH_DEGREES: full frontier; else only out.v / out.capacity are used
    unsigned long long *d_tail_out;
    unsigned long long *d_tail_clear;
    int *d_overflow;
    const SizeT *d_row_offsets;    // WITH_DEGREES only
};

// Compacting filter.  WITH_DEGREES = the output feeds a load-balanced advance (vertex frontier with degree
// prefix); otherwise ids only.
template <typename KernelPolicy, typename ProblemData, typename Functor, bool WITH_DEGREES>
__global__ __launch_bounds__(KernelPolicy::THREADS) void Kernel(
    FilterArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> a, typename ProblemData::DataSlice slice)
{
    typedef typename ProblemData::VertexId VertexId;
    typedef typename ProblemData::SizeT SizeT;
    typedef typename ProblemData::Value Value;
    constexpr int THREADS = KernelPolicy::THREADS;
    constexpr int ITEMS = KernelPolicy::ITEMS;
    constexpr int TILE = KernelPolicy::TILE;
    typedef FrontierWriter<THREADS, KernelPolicy::STAGE_CAPACITY, VertexId, SizeT> Writer;
    __shared__ typename Writer::Storage s_writer;

    if (blockIdx.x == 0 && threadIdx.x == 0 && a.d_tail_clear) *a.d_tail_clear = 0ull;
    Writer::Init(s_writer);
    __syncthreads();

    const long long tiles = (static_cast<long long>(a.num_elements) + TILE - 1) / TILE;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int pending = Writer::Count(s_writer);
        __syncthreads();
        if (pending > KernelPolicy::STAGE_CAPACITY - TILE) {
            if (WITH_DEGREES) Writer::template Flush<true>(s_writer, pending, a.out, a.d_tail_out, a.d_overflow, a.d_row_offsets);
            else Writer::FlushIds(s_writer, pending, a.out.v, a.out.capacity, a.d_tail_out, a.d_overflow);
        }
        VertexId node[ITEMS];
        bool keep[ITEMS];
        SizeT idx[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            idx[k] = static_cast<SizeT>(tile * TILE + k * THREADS + threadIdx.x);
            node[k] = -1;
            if (idx[k] < a.num_elements) node[k] = a.d_in ? a.d_in[idx[k]] : static_cast<VertexId>(idx[k]);
        }
        int mine = 0;
        if constexpr (HasIssueFilter<Functor, VertexId, typename ProblemData::DataSlice, Value, SizeT>::value) {
            typedef decltype(Functor::IssueFilter(node[0], &slice, Value(0), idx[0])) Token;
            Token token[ITEMS];
#pragma unroll
            for (int k = 0; k < ITEMS; ++k)  // branch-free screens: their loads are in flight together
                keep[k] = (node[k] != -1) & Functor::ScreenFilter(node[k] != -1 ? node[k] : static_cast<VertexId>(0), &slice, Value(0), idx[k]);
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) {
                token[k] = Token();
                if (keep[k]) token[k] = Functor::IssueFilter(node[k], &slice, Value(0), idx[k]);
            }
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) {
                keep[k] = keep[k] && Functor::ResolveFilter(token[k], node[k], &slice, Value(0), idx[k]);
                if (keep[k]) {
                    Functor::ApplyFilter(node[k], &slice, Value(0), idx[k]);
                    ++mine;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) {
                keep[k] = node[k] != -1 && Functor::CondFilter(node[k], &slice, Value(0), idx[k]);
                if (keep[k]) {
                    Functor::ApplyFilter(node[k], &slice, Value(0), idx[k]);
                    ++mine;
                }
            }
        }
        int pos = Writer::Reserve(s_writer, mine);
#pragma unroll
        for (int k = 0; k < ITEMS; ++k)
            if (keep[k]) s_writer.buf[pos++] = node[k];
        __syncthreads();
    }
    const int rest = Writer::Count(s_writer);
    __syncthreads();
    if (WITH_DEGREES) Writer::template Flush<true>(s_writer, rest, a.out, a.d_tail_out, a.d_overflow, a.d_row_offsets);
    else Writer::FlushIds(s_writer, rest, a.out.v, a.out.capacity, a.d_tail_out, a.d_overflow);
}

// Non-compacting sweep (reference Kernel2): ApplyFilter(element) for every element with CondFilter true.
template <typename KernelPolicy, typename ProblemData, typename Functor>
__global__ __launch_bounds__(KernelPolicy::THREADS) void ApplyKernel(
    const typename ProblemData::VertexId *d_in, typename ProblemData::SizeT num_elements,
    typename ProblemData::VertexId *d_out, typename ProblemData::DataSlice slice)
{
    typedef typename ProblemData::VertexId VertexId;
    typedef typename ProblemData::SizeT SizeT;
    typedef typename ProblemData::Value Value;
    const long long stride = static_cast<long long>(gridDim.x) * KernelPolicy::THREADS;
    for (long long i = static_cast<long long>(blockIdx.x) * KernelPolicy::THREADS + threadIdx.x; i < num_elements;
         i += stride) {
        const VertexId node = d_in ? d_in[i] : static_cast<VertexId>(i);
        if (node != -1 && Functor::CondFilter(node, &slice, Value(0), static_cast<SizeT>(i)))
            Functor::ApplyFilter(node, &slice, Value(0), static_cast<SizeT>(i));
        if (d_out) d_out[i] = node;
    }
}

inline int SweepGrid(long long num_elements, int threads, int max_grid)
{
    long long blocks = (num_elements + threads - 1) / threads;
    if (blocks < 1) blocks = 1;
    if (blocks > max_grid) blocks = max_grid;
    return static_cast<int>(blocks);
}

template <typename KernelPolicy, typename ProblemData, typename Functor>
hipError_t LaunchApply(const typename ProblemData::VertexId *d_in, typename ProblemData::SizeT num_elements,
                       typename ProblemData::VertexId *d_out, const typename ProblemData::DataSlice &slice,
                       int max_grid_size, hipStream_t stream)
{
    if (num_elements <= 0) return hipSuccess;
    hipLaunchKernelGGL((ApplyKernel<KernelPolicy, ProblemData, Functor>),
                       dim3(SweepGrid(num_elements, KernelPolicy::THREADS, max_grid_size)), dim3(KernelPolicy::THREADS), 0,
                       stream, d_in, num_elements, d_out, slice);
    return util::GRError("filter::ApplyKernel launch failed", __FILE__, __LINE__);
}

// ---- non-compacting sweep over the identity queue 0 .. n-1 that SKIPS elements whose byte in d_done is set ----
// (CC: an edge that is marked done stays done, and after the first hooking sweep most are.)  The test runs on 16 flag bytes
// at a time -- one 16-byte load per lane, 1 KiB per wave instruction -- and only the elements still open reach the functor;
// a sweep that reads one flag BYTE per lane spends its time issuing 64-byte wave loads (measured: 270 us for the 265 MB of
// flags of a scale-24 graph in which every edge is done).  d_done must be 16-byte aligned (hipMalloc) and padded to 16.
template <typename KernelPolicy, typename ProblemData, typename Functor>
__global__ __launch_bounds__(KernelPolicy::THREADS) void ApplySkipKernel(long long num_elements, const unsigned char *d_done,
                                                                         typename ProblemData::DataSlice slice)
{
    typedef typename ProblemData::VertexId VertexId;
    typedef typename ProblemData::SizeT SizeT;
    typedef typename ProblemData::Value Value;
    const long long groups = (num_elements + 15) / 16;
    const long long stride = static_cast<long long>(gridDim.x) * KernelPolicy::THREADS;
    for (long long g = static_cast<long long>(blockIdx.x) * KernelPolicy::THREADS + threadIdx.x; g < groups; g += stride) {
        const uint4 f = reinterpret_cast<const uint4 *>(d_done)[g];
        const unsigned w[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // (flag bytes are 0 or 1) a dword of four set flags is 0x01010101: nothing to do
            if (w[q] == 0x01010101u) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const long long i = g * 16 + q * 4 + b;
                if (((w[q] >> (8 * b)) & 0xFFu) == 0u && i < num_elements) {
                    const VertexId node = static_cast<VertexId>(i);
                    if (Functor::CondFilter(node, &slice, Value(0), static_cast<SizeT>(i))) Functor::ApplyFilter(node, &slice, Value(0), static_cast<SizeT>(i));
                }
            }
        }
    }
}

template <typename KernelPolicy, typename ProblemData, typename Functor>
hipError_t LaunchApplySkip(typename ProblemData::SizeT num_elements, const unsigned char *d_done, const typename ProblemData::DataSlice &slice,
                           int max_grid_size, hipStream_t stream)
{
    if (num_elements <= 0) return hipSuccess;
    const long long groups = (static_cast<long long>(num_elements) + 15) / 16;
    hipLaunchKernelGGL((ApplySkipKernel<KernelPolicy, ProblemData, Functor>), dim3(SweepGrid(groups, KernelPolicy::THREADS, max_grid_size)),
                       dim3(KernelPolicy::THREADS), 0, stream, static_cast<long long>(num_elements), d_done, slice);
    return util::GRError("filter::ApplySkipKernel launch failed", __FILE__, __LINE__);
}

template <typename KernelPolicy, typename ProblemData, typename Functor, bool WITH_DEGREES>
hipError_t LaunchKernel(const FilterArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> &args,
                        const typename ProblemData::DataSlice &slice, int max_grid_size, hipStream_t stream)
{
    if (args.num_elements <= 0) return hipSuccess;
    const long long tiles = (static_cast<long long>(args.num_elements) + KernelPolicy::TILE - 1) / KernelPolicy::TILE;
    hipLaunchKernelGGL((Kernel<KernelPolicy, ProblemData, Functor, WITH_DEGREES>),
                       dim3(static_cast<unsigned>(tiles < max_grid_size ? tiles : max_grid_size)),
                       dim3(KernelPolicy::THREADS), 0, stream, args, slice);
    return util::GRError("filter::Kernel launch failed", __FILE__, __LINE__);
}

}  // namespace filter
}  // namespace oprtr
}  // namespace gunrock
