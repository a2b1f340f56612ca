// app/cc/cc_problem.hpp -- device data for connected components (Soman hook + pointer-jump).
//
// Same contract as the reference's CCProblem (gunrock/app/cc/cc_problem.cuh:36-440):
//   DataSlice { d_component_ids, d_masks, d_marks, d_froms, d_tos, d_vertex_flag, d_edge_flag }   (:48-57)
//   Init(stream_from_host, graph, num_gpus): uploads the CSR and the edge list                     (:221-345)
//   Reset(frontier_type): component ids = iota, masks = 0, marks = false                            (:361-440)
//   Extract(h_component_ids): copies ids, counts roots into num_components                          (:144-175)
//   ComputeCCHistogram(ids, roots, histogram)                                                        (:185-210)
// MI355X-first differences: the edge list is not a host-built copy (cc_problem.cuh:262-272 builds and uploads
// 2 x m ints): d_tos IS the CSR column array and d_froms is expanded on the GPU from the row offsets; the
// iota "queues" the reference stores and re-reads every sweep (:386-408) do not exist (the filter operator
// takes a NULL = identity queue); ComputeCCHistogram is O(n) instead of O(n x components).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>
#include <vector>

#include <gunrock/app/problem_base.hpp>
#include <gunrock/graphio/device_csr.hpp>
#include <gunrock/graphio/symmetry.hpp>
#include <gunrock/util/device_intrinsics.hpp>
#include <gunrock/util/memset_kernel.hpp>

namespace gunrock {
namespace app {
namespace cc {

// froms[e] = row of edge e.  One wave per 64 rows: short rows are written by their lane, long rows by the wave.
template <typename VertexId, typename SizeT>
__global__ void ExpandRowsKernel(const SizeT *d_row_offsets, SizeT nodes, VertexId *d_froms)
{
    const unsigned lane = util::LaneId();
    const long long wave0 = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) / util::kWaveSize;
    const long long nwaves = static_cast<long long>(gridDim.x) * blockDim.x / util::kWaveSize;
    const long long groups = (static_cast<long long>(nodes) + 63) / 64;
    for (long long g = wave0; g < groups; g += nwaves) {
        const long long v = g * 64 + lane;
        SizeT b = 0, e = 0;
        if (v < nodes) { b = d_row_offsets[v]; e = d_row_offsets[v + 1]; }
        const bool long_row = (e - b) > 16;
        if (!long_row) for (SizeT i = b; i < e; ++i) d_froms[i] = static_cast<VertexId>(v);
        unsigned long long todo = __ballot(long_row);
        while (todo) {
            const int leader = __ffsll(static_cast<long long>(todo)) - 1;
            const SizeT lb = __shfl(b, leader, util::kWaveSize), le = __shfl(e, leader, util::kWaveSize);
            const VertexId lv = static_cast<VertexId>(g * 64 + leader);
            for (SizeT i = lb + static_cast<SizeT>(lane); i < le; i += util::kWaveSize) d_froms[i] = lv;
            todo &= todo - 1;
        }
    }
}

// Does every edge (f, t) with f < t have its mirror (t, f)?  One lane per such edge, binary search of f in row t (rows sorted
// ascending, as Csr::FromCoo builds them; an unsorted row can only produce a false "no", which is the safe answer).  That is
// all the hooking sweeps need to skip the f < t orientation: the mirror, an edge with from > to, is always processed.
template <typename VertexId, typename SizeT>
__global__ void MirrorCheckKernel(const SizeT *d_row_offsets, const VertexId *d_froms, const VertexId *d_tos, long long edges,
                                  int *d_missing)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    bool missing = false;
    for (long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < edges; e += stride) {
        const VertexId f = d_froms[e], t = d_tos[e];
        if (f >= t) continue;
        SizeT lo = d_row_offsets[t], hi = d_row_offsets[t + 1];
        while (lo < hi) {
            const SizeT mid = lo + (hi - lo) / 2;
            if (d_tos[mid] < f) lo = mid + 1; else hi = mid;
        }
        if (lo >= d_row_offsets[t + 1] || d_tos[lo] != f) missing = true;
    }
    if (__ballot(missing) && util::LaneId() == 0) *d_missing = 1;
}

// ---- mirrored input: only the orientation with from > to ever hooks (HookInit, HookMax), so only it is materialised ----
// d_count[v] = entries of row v below v (rows are sorted ascending: a binary search); d_count[nodes] = 0 closes the scan
template <typename VertexId, typename SizeT>
__global__ void LowerCountKernel(const SizeT *d_row_offsets, const VertexId *d_cols, long long nodes, unsigned *d_count)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long v = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; v <= nodes; v += stride) {
        unsigned below = 0;
        if (v < nodes) {
            const SizeT b = d_row_offsets[v];
            SizeT lo = b, hi = d_row_offsets[v + 1];
            while (lo < hi) {
                const SizeT mid = lo + (hi - lo) / 2;
                if (d_cols[mid] < static_cast<VertexId>(v)) lo = mid + 1; else hi = mid;
            }
            below = static_cast<unsigned>(lo - b);
        }
        d_count[v] = below;
    }
}
// the (from, to) pairs with to < from, row by row (one wave per 64 rows; long rows by the whole wave)
template <typename VertexId, typename SizeT>
__global__ void LowerFillKernel(const SizeT *d_row_offsets, const VertexId *d_cols, const SizeT *d_low_offsets, long long nodes,
                                VertexId *d_low_froms, VertexId *d_low_tos)
{
    const unsigned lane = util::LaneId();
    const long long wave0 = (static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x) / util::kWaveSize;
    const long long nwaves = static_cast<long long>(gridDim.x) * blockDim.x / util::kWaveSize;
    const long long groups = (nodes + 63) / 64;
    for (long long g = wave0; g < groups; g += nwaves) {
        const long long v = g * 64 + lane;
        SizeT src = 0, dst = 0, len = 0;
        if (v < nodes) {
            src = d_row_offsets[v];
            dst = d_low_offsets[v];
            len = d_low_offsets[v + 1] - dst;
        }
        const bool long_row = len > 16;
        if (!long_row)
            for (SizeT i = 0; i < len; ++i) {
                d_low_froms[dst + i] = static_cast<VertexId>(v);
                d_low_tos[dst + i] = d_cols[src + i];
            }
        unsigned long long todo = __ballot(long_row);
        while (todo) {
            const int leader = __ffsll(static_cast<long long>(todo)) - 1;
            const SizeT ls = __shfl(src, leader, util::kWaveSize), ld = __shfl(dst, leader, util::kWaveSize), ll = __shfl(len, leader, util::kWaveSize);
            const VertexId lv = static_cast<VertexId>(g * 64 + leader);
            for (SizeT i = static_cast<SizeT>(lane); i < ll; i += util::kWaveSize) {
                d_low_froms[ld + i] = lv;
                d_low_tos[ld + i] = d_cols[ls + i];
            }
            todo &= todo - 1;
        }
    }
}

// The component most of a sample of vertices belongs to (one workgroup of kGiantSamples threads; the sweeps before it left every vertex
// pointing at its root): d_giant[0] = that root, [1] = samples holding it, [2] = 0.  Isolated vertices are their own roots and
// never form a majority, so on a graph without a big component [1] stays small and the caller keeps the edge-form sweeps.
constexpr int kGiantSamples = 256;
template <typename VertexId>
__global__ __launch_bounds__(kGiantSamples) void PickGiantKernel(const VertexId *d_component_ids, long long nodes, VertexId *d_giant)
{
    __shared__ VertexId s_root[kGiantSamples];
    __shared__ int s_best_count;
    const int t = threadIdx.x;
    const long long stride = nodes >= kGiantSamples ? nodes / kGiantSamples : 1;
    const long long v = static_cast<long long>(t) * stride;
    VertexId r = v < nodes ? d_component_ids[v] : static_cast<VertexId>(-1);
    if (r >= 0) r = d_component_ids[r];  // (one more hop costs nothing and tolerates a vertex one jump short of its root)
    s_root[t] = r;
    if (t == 0) s_best_count = 0;
    __syncthreads();
    int count = 0;
    if (r >= 0)
        for (int j = 0; j < kGiantSamples; ++j) count += s_root[j] == r;
    atomicMax(&s_best_count, count);
    __syncthreads();
    if (r >= 0 && count == s_best_count) {  // (several threads may hold the winning root: they write the same values)
        d_giant[0] = r;
        d_giant[1] = static_cast<VertexId>(count);
    }
    if (t == 0) {
        d_giant[2] = 0;
        if (s_best_count == 0) { d_giant[0] = static_cast<VertexId>(-1); d_giant[1] = 0; }
    }
}

template <typename VertexId, typename SizeT>
__global__ void FirstLowerKernel(const SizeT *d_low_offsets, const VertexId *d_low_tos, long long nodes, VertexId *d_first)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long v = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; v < nodes; v += stride) {
        const SizeT b = d_low_offsets[v];
        d_first[v] = d_low_offsets[v + 1] > b ? d_low_tos[b] : static_cast<VertexId>(v);
    }
}

template <typename _VertexId, typename _SizeT, typename _Value, bool _USE_DOUBLE_BUFFER>
struct CCProblem : ProblemBase<_VertexId, _SizeT, _Value, _USE_DOUBLE_BUFFER> {
    typedef ProblemBase<_VertexId, _SizeT, _Value, _USE_DOUBLE_BUFFER> Base;
    typedef _VertexId VertexId;
    typedef _SizeT SizeT;
    typedef _Value Value;
    static constexpr bool ENABLE_IDEMPOTENCE = false;
    static constexpr bool MARK_PREDECESSORS = false;

    struct DataSlice {
        VertexId *d_component_ids = nullptr;  // parent pointer per vertex; converges to the component's min id
        int *d_masks = nullptr;               // 0 = root candidate, 1 = non-root, -1 = settled root
        unsigned char *d_marks = nullptr;     // per edge: both ends already in one tree
        VertexId *d_froms = nullptr;          // per edge: source vertex
        const VertexId *d_tos = nullptr;      // per edge: destination vertex (= CSR column_indices)
        int symmetric = 0;                    // every edge has its mirror: hooking sweeps need one orientation only
        const VertexId *d_first_lower = nullptr;  // compact (mirrored) layout: per vertex its smallest neighbour below it, or itself
        // row form of the hooking sweep (mirrored input; cc_functor.hpp HookMaxRowFunctor): the whole CSR and three device words --
        // [0] root of the sampled giant component (-1: none), [1] how many of the samples it holds, [2] raised by a vertex whose
        // row is too long for one lane
        const SizeT *d_row_offsets = nullptr;
        const VertexId *d_columns = nullptr;
        VertexId *d_giant = nullptr;
        SizeT row_form_limit = 256;
        const SizeT *d_low_offsets = nullptr;     // compact layout: row extents of the from > to list (the neighbour rounds read them)
        int neighbour_round = 0;                  // which lower neighbour of every vertex a neighbour round hooks (0 = the smallest)
        int *d_vertex_flag = nullptr;         // cleared by a pointer-jump sweep that changed something
        int *d_edge_flag = nullptr;           // cleared by a hook sweep that hooked something
    };

    DataSlice **data_slices = nullptr;
    unsigned int num_components = 0;
    int *h_flags = nullptr;  // pinned: [0] vertex flag, [1] edge flag
    // edges the hooking sweeps run over: all of them, or -- mirrored input -- the from > to orientation only (AllocData)
    _SizeT sweep_edges = 0;
    bool compact_mirrored = true;      // policy (tests switch it off to run the parking path)
    bool row_form = true;              // policy: hooking sweeps in row form when a sampled component dominates (cc_functor.hpp)
    _VertexId *d_owned_first = nullptr;
    _SizeT *d_owned_low_offsets = nullptr;
    int neighbour_rounds = 1;          // policy: neighbour rounds in front of the row-form sweeps (0: first hooking sweep in edge form).
                                       // Scale-24 R-MAT, whole CC: 0 -> 1.51 ms, 1 -> 0.84, 2 -> 1.11, 3 -> 1.38 (a round costs ~0.3 ms:
                                       // 16 M random parent gathers and the jumps; the second one no longer shortens what follows)
    _VertexId *d_owned_tos = nullptr;  // the compact `to` array when the problem owns one (d_tos otherwise aliases the CSR's columns)

    ~CCProblem() override
    {
        if (data_slices) {
            DataSlice *ds = data_slices[0];
            if (ds) {
                if (ds->d_component_ids) util::GRError(hipFree(ds->d_component_ids), "CCProblem hipFree failed", __FILE__, __LINE__);
                if (ds->d_masks) util::GRError(hipFree(ds->d_masks), "CCProblem hipFree failed", __FILE__, __LINE__);
                if (ds->d_marks) util::GRError(hipFree(ds->d_marks), "CCProblem hipFree failed", __FILE__, __LINE__);
                if (ds->d_froms) util::GRError(hipFree(ds->d_froms), "CCProblem hipFree failed", __FILE__, __LINE__);
                if (ds->d_vertex_flag) util::GRError(hipFree(ds->d_vertex_flag), "CCProblem hipFree failed", __FILE__, __LINE__);
                delete ds;
            }
            delete[] data_slices;
        }
        if (h_flags) util::GRError(hipHostFree(h_flags), "CCProblem hipHostFree failed", __FILE__, __LINE__);
        if (d_owned_tos) util::GRError(hipFree(d_owned_tos), "CCProblem hipFree failed", __FILE__, __LINE__);
        if (d_owned_first) util::GRError(hipFree(d_owned_first), "CCProblem hipFree failed", __FILE__, __LINE__);
        if (d_owned_low_offsets) util::GRError(hipFree(d_owned_low_offsets), "CCProblem hipFree failed", __FILE__, __LINE__);
    }

    hipError_t AllocData()
    {
        hipError_t retval = hipSuccess;
        data_slices = new DataSlice *[1];
        data_slices[0] = new DataSlice();
        DataSlice *ds = data_slices[0];
        GraphSlice<VertexId, SizeT, Value> *gs = this->graph_slices[0];
        const size_t n = static_cast<size_t>(this->nodes > 0 ? this->nodes : 1);
        const size_t m = static_cast<size_t>(this->edges > 0 ? this->edges : 1);
        GR_CHECK(hipMalloc(&ds->d_component_ids, sizeof(VertexId) * n), "CCProblem hipMalloc d_component_ids failed");
        GR_CHECK(hipMalloc(&ds->d_masks, sizeof(int) * n), "CCProblem hipMalloc d_masks failed");
        GR_CHECK(hipMalloc(&ds->d_marks, m + 16), "CCProblem hipMalloc d_marks failed");  // (+16: the skipping sweep reads 16 flags at a time)
        GR_CHECK(hipMalloc(&ds->d_froms, sizeof(VertexId) * m), "CCProblem hipMalloc d_froms failed");
        GR_CHECK(hipMalloc(&ds->d_vertex_flag, sizeof(int) * 2), "CCProblem hipMalloc flags failed");
        ds->d_edge_flag = ds->d_vertex_flag + 1;
        ds->d_tos = gs->d_column_indices;
        GR_CHECK(hipHostMalloc(&h_flags, sizeof(int) * 2, hipHostMallocDefault), "CCProblem hipHostMalloc failed");
        if (this->nodes > 0) {
            hipLaunchKernelGGL((ExpandRowsKernel<VertexId, SizeT>), dim3(2048), dim3(256), 0, gs->stream, gs->d_row_offsets,
                               this->nodes, ds->d_froms);
            GR_CHECK(hipGetLastError(), "ExpandRowsKernel launch failed");
            GR_CHECK(hipStreamSynchronize(gs->stream), "ExpandRowsKernel failed");
        }
        // Mirrored input (what the reference's CC drivers build, test_cc.cu "undirected")?  Then (f, t) with f < t and its
        // mirror are the same hook, and the hooking sweeps skip the f < t orientation.  Checked once here, outside Enact.
        ds->symmetric = 0;
        if (this->edges > 0) {
            GR_CHECK(hipMemsetAsync(ds->d_vertex_flag, 0, sizeof(int) * 2, gs->stream), "CCProblem memset failed");
            hipLaunchKernelGGL((MirrorCheckKernel<VertexId, SizeT>), dim3(4096), dim3(256), 0, gs->stream, gs->d_row_offsets, ds->d_froms,
                               ds->d_tos, static_cast<long long>(this->edges), ds->d_vertex_flag);
            GR_CHECK(hipGetLastError(), "MirrorCheckKernel launch failed");
            int missing = 0;
            GR_CHECK(hipMemcpyAsync(&missing, ds->d_vertex_flag, sizeof(int), hipMemcpyDeviceToHost, gs->stream), "CCProblem read failed");
            GR_CHECK(hipStreamSynchronize(gs->stream), "MirrorCheckKernel failed");
            ds->symmetric = missing ? 0 : 1;
        }
        sweep_edges = this->edges;
        if (const char *env = std::getenv("GUNROCK_CC_COMPACT")) compact_mirrored = env[0] != '0';  // (tests: "0" keeps both orientations)
        if (const char *env = std::getenv("GUNROCK_CC_ROWFORM")) row_form = env[0] != '0';
        if (const char *env = std::getenv("GUNROCK_CC_NEIGHBOUR_ROUNDS")) neighbour_rounds = std::atoi(env);
        if (ds->symmetric && this->edges > 0 && compact_mirrored) {
            // Mirrored input: the orientation from < to would be parked at first sight by every hooking sweep (its mirror does the very
            // same hook).  Materialise only the from > to pairs -- half the edge stream for HookInit and for every HookMax sweep
            // (the reference sweeps both orientations of its COO, cc_enactor.cuh:407-424, 560-600).
            const long long n1 = static_cast<long long>(this->nodes) + 1;
            unsigned *d_count = nullptr;
            unsigned long long *d_sums = nullptr;
            SizeT *d_low_offsets = nullptr;
            GR_CHECK(hipMalloc(&d_count, sizeof(unsigned) * static_cast<size_t>(n1)), "CCProblem hipMalloc failed");
            GR_CHECK(hipMalloc(&d_low_offsets, sizeof(SizeT) * static_cast<size_t>(n1)), "CCProblem hipMalloc failed");
            GR_CHECK(hipMalloc(&d_sums, sizeof(unsigned long long) * static_cast<size_t>(graphio::ScanScratchWords(n1))), "CCProblem hipMalloc failed");
            hipLaunchKernelGGL((LowerCountKernel<VertexId, SizeT>), dim3(2048), dim3(256), 0, gs->stream, gs->d_row_offsets, ds->d_tos,
                               static_cast<long long>(this->nodes), d_count);
            GR_CHECK(hipGetLastError(), "LowerCountKernel launch failed");
            GR_CHECK(graphio::DeviceExclusiveScan<SizeT>(d_count, d_low_offsets, n1, d_sums, gs->stream), "CCProblem lower-offset scan failed");
            SizeT low_edges = 0;
            GR_CHECK(hipMemcpyAsync(&low_edges, d_low_offsets + this->nodes, sizeof(SizeT), hipMemcpyDeviceToHost, gs->stream), "CCProblem read failed");
            GR_CHECK(hipStreamSynchronize(gs->stream), "CCProblem lower-offset scan failed");
            const size_t lm = static_cast<size_t>(low_edges > 0 ? low_edges : 1);
            VertexId *d_low_froms = nullptr, *d_low_tos = nullptr;
            GR_CHECK(hipMalloc(&d_low_froms, sizeof(VertexId) * lm), "CCProblem hipMalloc failed");
            GR_CHECK(hipMalloc(&d_low_tos, sizeof(VertexId) * lm), "CCProblem hipMalloc failed");
            hipLaunchKernelGGL((LowerFillKernel<VertexId, SizeT>), dim3(2048), dim3(256), 0, gs->stream, gs->d_row_offsets, ds->d_tos, d_low_offsets,
                               static_cast<long long>(this->nodes), d_low_froms, d_low_tos);
            GR_CHECK(hipGetLastError(), "LowerFillKernel launch failed");
            GR_CHECK(hipStreamSynchronize(gs->stream), "LowerFillKernel failed");
            // ... and per vertex its smallest neighbour below it (the first entry of its compact row), which is all HookInit needs
            VertexId *d_first = nullptr;
            GR_CHECK(hipMalloc(&d_first, sizeof(VertexId) * static_cast<size_t>(this->nodes > 0 ? this->nodes : 1)), "CCProblem hipMalloc failed");
            hipLaunchKernelGGL((FirstLowerKernel<VertexId, SizeT>), dim3(2048), dim3(256), 0, gs->stream, d_low_offsets, d_low_tos,
                               static_cast<long long>(this->nodes), d_first);
            GR_CHECK(hipGetLastError(), "FirstLowerKernel launch failed");
            GR_CHECK(hipStreamSynchronize(gs->stream), "FirstLowerKernel failed");
            ds->d_first_lower = d_first;
            d_owned_first = d_first;
            // The row form lets a vertex of the giant component skip its row and relies on the OTHER end of each of its edges walking
            // its own: that needs every edge mirrored, both orientations.  `symmetric` above is weaker (only from < to edges were
            // checked -- enough to drop that orientation): a graph whose unmirrored edges all point from a higher to a lower id
            // passes it, and an edge giant -> outside would then be seen by nobody.  So the exact test decides (graphio/symmetry.hpp).
            bool fully_mirrored = false;
            if constexpr (std::is_same<VertexId, int>::value && std::is_same<SizeT, int>::value) {
                if (row_form)
                    GR_CHECK(graphio::DeviceIsSymmetric(static_cast<int>(this->nodes), static_cast<long long>(this->edges), gs->d_row_offsets,
                                                        gs->d_column_indices, gs->stream, fully_mirrored),
                             "CCProblem symmetry test failed");
            }
            if (fully_mirrored) {
                ds->d_row_offsets = gs->d_row_offsets;
                ds->d_columns = gs->d_column_indices;
            }
            GR_CHECK(hipFree(ds->d_froms), "CCProblem hipFree failed");  // the full expansion is not needed any more
            ds->d_froms = d_low_froms;
            ds->d_tos = d_low_tos;
            d_owned_tos = d_low_tos;
            sweep_edges = low_edges;
            GR_CHECK(hipFree(d_count), "CCProblem hipFree failed");
            ds->d_low_offsets = d_low_offsets;  // (kept: the neighbour rounds of the row form index the compact rows)
            d_owned_low_offsets = d_low_offsets;
            GR_CHECK(hipFree(d_sums), "CCProblem hipFree failed");
        }
        return retval;
    }

    hipError_t Init(bool stream_from_host, const Csr<VertexId, Value, SizeT> &graph, int num_gpus = 1)
    {
        hipError_t retval = hipSuccess;
        if ((retval = Base::Init(stream_from_host, graph, num_gpus))) return retval;
        return AllocData();
    }

    hipError_t InitFromDevice(SizeT nodes, SizeT edges, SizeT *d_row_offsets, VertexId *d_column_indices)
    {
        hipError_t retval = hipSuccess;
        if ((retval = Base::InitFromDevice(nodes, edges, d_row_offsets, d_column_indices))) return retval;
        return AllocData();
    }

    hipError_t Reset(FrontierType /*frontier_type*/, double /*queue_sizing*/ = 1.0)
    {
        hipError_t retval = hipSuccess;
        DataSlice *ds = data_slices[0];
        hipStream_t stream = this->graph_slices[0]->stream;
        util::MemsetIdx(ds->d_component_ids, this->nodes, stream);
        util::Memset(ds->d_masks, 0, this->nodes, stream);
        GR_CHECK(hipMemsetAsync(ds->d_marks, 0, static_cast<size_t>(sweep_edges > 0 ? sweep_edges : 1) + 16, stream),
                 "CCProblem memset d_marks failed");
        GR_CHECK(hipMemsetAsync(ds->d_vertex_flag, 0, sizeof(int) * 2, stream), "CCProblem memset flags failed");
        GR_CHECK(hipStreamSynchronize(stream), "CCProblem Reset sync failed");
        num_components = 0;
        return retval;
    }

    hipError_t Extract(VertexId *h_component_ids)
    {
        hipError_t retval = hipSuccess;
        GR_CHECK(hipStreamSynchronize(this->graph_slices[0]->stream), "CCProblem Extract sync failed");
        if (this->nodes > 0)
            GR_CHECK(hipMemcpy(h_component_ids, data_slices[0]->d_component_ids, sizeof(VertexId) * static_cast<size_t>(this->nodes),
                               hipMemcpyDeviceToHost),
                     "CCProblem hipMemcpy d_component_ids failed");
        num_components = 0;
        for (SizeT i = 0; i < this->nodes; ++i) num_components += (h_component_ids[i] == i);  // cc_problem.cuh:164-170
        return retval;
    }

    // roots in increasing id order and the size of each component, O(n)
    void ComputeCCHistogram(const VertexId *h_component_ids, VertexId *h_roots, unsigned int *h_histograms)
    {
        std::vector<int> slot(static_cast<size_t>(this->nodes > 0 ? this->nodes : 1), -1);
        num_components = 0;
        for (SizeT i = 0; i < this->nodes; ++i) {
            if (h_component_ids[i] == i) {
                slot[i] = static_cast<int>(num_components);
                h_roots[num_components] = i;
                h_histograms[num_components] = 0;
                ++num_components;
            }
        }
        for (SizeT i = 0; i < this->nodes; ++i) {
            const int s = slot[h_component_ids[i]];
            if (s >= 0) ++h_histograms[s];
        }
    }
};

}  // namespace cc
}  // namespace app
}  // namespace gunrock
