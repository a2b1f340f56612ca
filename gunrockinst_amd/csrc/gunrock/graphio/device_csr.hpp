// graphio/device_csr.hpp -- COO -> CSR on the GPU with Csr::FromCoo's graph semantics.
//
// What the reference does on the host (gunrock/csr.cuh:247-340, coo.cuh:71-85): std::stable_sort of the tuples by
// (row, col), drop self loops and duplicates, fill row_offsets including trailing empty rows -- minutes at scale-24.
// Here, all in HBM ("graph ingest on device"):
//
//   1. MakeKeysKernel      one 64-bit key per directed tuple, key = row << cb | col (cb = bits of a vertex id); an
//                          undirected input also emits the mirrored tuple (test_bfs.cu "undirected" doubling,
//                          market.cuh:172-183); self loops -- and, for a vertex-cut partition, tuples whose source
//                          another rank owns -- become the all-ones sentinel, which sorts behind every real key.
//   2. LSD radix sort      8 bits per pass over the 2*cb significant bits: per-tile digit histogram -> device-wide
//                          exclusive scan (digit-major) -> stable scatter.  Ranks inside a tile come from wave ballots
//                          (8 ballots match the lanes holding the same digit) plus ordered per-wave counters in LDS,
//                          so equal digits keep their order: the sort is stable, as LSD needs.
//   3. FlagKernel          keep[i] = key is real and differs from its predecessor (duplicates are adjacent now).
//   4. exclusive scan of the flags = output position of every kept tuple; total = number of edges.
//   5. EmitCsrKernel       col_indices[pos] = col; a kept tuple whose row differs from its predecessor's writes
//                          row_offsets for every row in between (empty rows included).
//
// The device-wide scan is the three-phase kind (tile sums, recursive scan of the sums, add back), hand-written: it
// is also the "device-wide scan" the north star asks for in place of the reference's moderngpu calls.
// Everything is integer and deterministic: the result equals Csr::FromCoo bit for bit (tests/test_device_csr_gpu.py).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#include <gunrock/util/device_intrinsics.hpp>
#include <gunrock/util/error_utils.hpp>

namespace gunrock {
namespace graphio {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;                      // per thread
constexpr int kScanTile = kScanThreads * kScanItems;  // 4096 elements per workgroup

// ---- device-wide exclusive scan of unsigned ints (sums may exceed 32 bits only in the totals: 64-bit block sums) ----
static __global__ __launch_bounds__(kScanThreads) void ScanTileSumsKernel(const unsigned *d_in, long long n, unsigned long long *d_sums)
{
    __shared__ unsigned long long s_wave[kScanThreads / util::kWaveSize];
    const long long base = static_cast<long long>(blockIdx.x) * kScanTile;
    unsigned long long sum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const long long i = base + static_cast<long long>(k) * kScanThreads + threadIdx.x;
        if (i < n) sum += d_in[i];
    }
    sum = util::WaveSum(sum);
    if (util::LaneId() == 0) s_wave[threadIdx.x / util::kWaveSize] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < kScanThreads / util::kWaveSize; ++w) t += s_wave[w];
        d_sums[blockIdx.x] = t;
    }
}

// single workgroup: exclusive scan of up to a few hundred thousand 64-bit sums, in place; writes the total after the end
static __global__ __launch_bounds__(1024) void ScanSumsKernel(unsigned long long *d_sums, long long count)
{
    __shared__ unsigned long long s_wave[1024 / util::kWaveSize];
    __shared__ unsigned long long s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (long long base = 0; base < count; base += 1024) {
        const long long i = base + threadIdx.x;
        const unsigned long long v = (i < count) ? d_sums[i] : 0ull;
        unsigned long long inc = v;  // inclusive scan inside the wave
        for (int o = 1; o < util::kWaveSize; o <<= 1) {
            const unsigned long long up = __shfl_up(inc, o, util::kWaveSize);
            if (static_cast<int>(util::LaneId()) >= o) inc += up;
        }
        if (util::LaneId() == util::kWaveSize - 1) s_wave[threadIdx.x / util::kWaveSize] = inc;
        __syncthreads();
        unsigned long long before = s_carry;
        for (unsigned w = 0; w < threadIdx.x / util::kWaveSize; ++w) before += s_wave[w];
        if (i < count) d_sums[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) d_sums[count] = s_carry;
}

// out[i] = tile offset + exclusive scan inside the tile (64-bit positions; OutT = unsigned or unsigned long long)
template <typename OutT>
static __global__ __launch_bounds__(kScanThreads) void ScanApplyKernel(const unsigned *d_in, long long n, const unsigned long long *d_sums,
                                                                OutT *d_out)
{
    __shared__ unsigned long long s_wave[kScanThreads / util::kWaveSize];
    const long long base = static_cast<long long>(blockIdx.x) * kScanTile + static_cast<long long>(threadIdx.x) * kScanItems;
    unsigned v[kScanItems];
    unsigned long long mine = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {  // blocked layout: thread t owns kScanItems consecutive elements
        v[k] = (base + k < n) ? d_in[base + k] : 0u;
        mine += v[k];
    }
    unsigned long long inc = mine;
    for (int o = 1; o < util::kWaveSize; o <<= 1) {
        const unsigned long long up = __shfl_up(inc, o, util::kWaveSize);
        if (static_cast<int>(util::LaneId()) >= o) inc += up;
    }
    if (util::LaneId() == util::kWaveSize - 1) s_wave[threadIdx.x / util::kWaveSize] = inc;
    __syncthreads();
    unsigned long long run = d_sums[blockIdx.x] + inc - mine;
    for (unsigned w = 0; w < threadIdx.x / util::kWaveSize; ++w) run += s_wave[w];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < n) d_out[base + k] = static_cast<OutT>(run);
        run += v[k];
    }
}

// Exclusive scan of n unsigned values; *d_total (device, 8 bytes) receives the grand total.  d_sums: scratch of
// (tiles + 1) 64-bit words.  Handles up to 1024 * 2^20 tiles in the single-workgroup middle phase comfortably
// (2^31 elements = 524288 tiles = 512 rounds of 1024).
template <typename OutT>
inline hipError_t DeviceExclusiveScan(const unsigned *d_in, OutT *d_out, long long n, unsigned long long *d_sums,
                                      hipStream_t stream)
{
    hipError_t retval = hipSuccess;
    const long long tiles = (n + kScanTile - 1) / kScanTile;
    if (tiles == 0) return util::GRError(hipMemsetAsync(d_sums, 0, sizeof(unsigned long long), stream), "scan memset failed", __FILE__, __LINE__);
    hipLaunchKernelGGL(ScanTileSumsKernel, dim3(static_cast<unsigned>(tiles)), dim3(kScanThreads), 0, stream, d_in, n, d_sums);
    GR_CHECK(hipGetLastError(), "ScanTileSumsKernel launch failed");
    hipLaunchKernelGGL(ScanSumsKernel, dim3(1), dim3(1024), 0, stream, d_sums, tiles);
    GR_CHECK(hipGetLastError(), "ScanSumsKernel launch failed");
    hipLaunchKernelGGL((ScanApplyKernel<OutT>), dim3(static_cast<unsigned>(tiles)), dim3(kScanThreads), 0, stream, d_in, n, d_sums,
                       d_out);
    GR_CHECK(hipGetLastError(), "ScanApplyKernel launch failed");
    return retval;
}
inline long long ScanScratchWords(long long n) { return (n + kScanTile - 1) / kScanTile + 2; }

// ---- keys ----
constexpr unsigned long long kSentinelKey = ~0ull;

// parts > 1: keep only tuples whose source is owned by `rank` (owner = v mod parts) and store the LOCAL row v div parts
// (the reference's ownership rule, problem_base.cuh:185-210); columns stay global.
static __global__ void MakeKeysKernel(const int *d_rows, const int *d_cols, long long pairs, int undirected, int col_bits, int nodes,
                               int parts, int rank, unsigned long long *d_keys)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    const unsigned long long valid_mask = (col_bits >= 32) ? ~0ull : ((1ull << (2 * col_bits)) - 1ull);
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < pairs; i += stride) {
        const unsigned r = static_cast<unsigned>(d_rows[i]), c = static_cast<unsigned>(d_cols[i]);
        auto key = [&](unsigned from, unsigned to) -> unsigned long long {
            if (from == to || from >= static_cast<unsigned>(nodes) || to >= static_cast<unsigned>(nodes))
                return kSentinelKey & valid_mask;  // self loop (dropped like FromCoo does), or an id outside the graph
            if (parts > 1) {
                if (from % static_cast<unsigned>(parts) != static_cast<unsigned>(rank)) return kSentinelKey & valid_mask;
                from /= static_cast<unsigned>(parts);
            }
            return (static_cast<unsigned long long>(from) << col_bits) | to;
        };
        if (undirected) {
            d_keys[2 * i] = key(r, c);
            d_keys[2 * i + 1] = key(c, r);
        } else {
            d_keys[i] = key(r, c);
        }
    }
}

// ---- LSD radix sort, 8 bits per pass ----
constexpr int kSortThreads = 256;
constexpr int kSortRounds = 16;                           // rounds of kSortThreads keys per tile
constexpr int kSortTile = kSortThreads * kSortRounds;     // 4096 keys per workgroup
constexpr int kSortWaves = kSortThreads / util::kWaveSize;

// hist[digit * tiles + tile] = keys of that tile with that digit
static __global__ __launch_bounds__(kSortThreads) void RadixHistogramKernel(const unsigned long long *d_keys, long long n, int shift,
                                                                     long long tiles, unsigned *d_hist)
{
    __shared__ unsigned s_hist[256];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const long long base = static_cast<long long>(blockIdx.x) * kSortTile;
#pragma unroll 4
    for (int r = 0; r < kSortRounds; ++r) {
        const long long i = base + static_cast<long long>(r) * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&s_hist[(d_keys[i] >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    d_hist[static_cast<long long>(threadIdx.x) * tiles + blockIdx.x] = s_hist[threadIdx.x];
}

// stable scatter: out[offset[digit][tile] + rank among the tile's keys of that digit, in input order] = key
static __global__ __launch_bounds__(kSortThreads) void RadixScatterKernel(const unsigned long long *d_in, long long n, int shift,
                                                                   long long tiles, const unsigned long long *d_offsets,
                                                                   unsigned long long *d_out)
{
    __shared__ unsigned long long s_base[256];            // running output position per digit
    __shared__ unsigned s_wave_cnt[kSortWaves][256];      // this round's keys per wave and digit
    s_base[threadIdx.x] = d_offsets[static_cast<long long>(threadIdx.x) * tiles + blockIdx.x];
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) s_wave_cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const unsigned lane = util::LaneId();
    const unsigned wave = threadIdx.x / util::kWaveSize;
    const long long base = static_cast<long long>(blockIdx.x) * kSortTile;
    for (int r = 0; r < kSortRounds; ++r) {
        const long long i = base + static_cast<long long>(r) * kSortThreads + threadIdx.x;
        const bool valid = i < n;
        const unsigned long long key = valid ? d_in[i] : 0ull;
        const unsigned digit = static_cast<unsigned>(key >> shift) & 0xFFu;
        // lanes of this wave holding the same digit: intersect the 8 per-bit ballots
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long set = __ballot((digit >> b) & 1u);
            same &= ((digit >> b) & 1u) ? set : ~set;
        }
        const unsigned rank_in_wave = static_cast<unsigned>(__popcll(same & ((1ull << lane) - 1ull)));
        if (valid && rank_in_wave == 0) s_wave_cnt[wave][digit] = static_cast<unsigned>(__popcll(same));
        __syncthreads();
        if (valid) {
            unsigned long long pos = s_base[digit] + rank_in_wave;
            for (unsigned w = 0; w < wave; ++w) pos += s_wave_cnt[w][digit];
            d_out[pos] = key;
        }
        __syncthreads();
        {   // one thread per digit: fold the round's counts into the running base and clear them
            unsigned add = 0;
#pragma unroll
            for (int w = 0; w < kSortWaves; ++w) {
                add += s_wave_cnt[w][threadIdx.x];
                s_wave_cnt[w][threadIdx.x] = 0;
            }
            s_base[threadIdx.x] += add;
        }
        __syncthreads();
    }
}

// ---- dedup flags, CSR emission ----
static __global__ void FlagKernel(const unsigned long long *d_keys, long long n, unsigned long long sentinel, unsigned *d_keep)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const unsigned long long k = d_keys[i];
        d_keep[i] = (k != sentinel && (i == 0 || d_keys[i - 1] != k)) ? 1u : 0u;
    }
}

static __global__ void EmitCsrKernel(const unsigned long long *d_keys, const unsigned *d_keep, const unsigned long long *d_pos, long long n,
                              int col_bits, int rows, long long edges, int *d_row_offsets, int *d_col_indices)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    const unsigned long long col_mask = (1ull << col_bits) - 1ull;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (!d_keep[i]) continue;
        const unsigned long long k = d_keys[i];
        const long long pos = static_cast<long long>(d_pos[i]);
        const int row = static_cast<int>(k >> col_bits);
        d_col_indices[pos] = static_cast<int>(k & col_mask);
        // rows (prev_row, row] start here; prev_row = -1 for the first kept tuple.  (The predecessor in the sorted array is
        // either the previous kept tuple or a duplicate of it: same row either way.)
        const int prev_row = (pos == 0) ? -1 : static_cast<int>(d_keys[i - 1] >> col_bits);
        for (int r = prev_row + 1; r <= row; ++r) d_row_offsets[r] = static_cast<int>(pos);
    }
}

static __global__ void CloseOffsetsKernel(const unsigned long long *d_keys, const unsigned *d_keep, long long n, int col_bits, int rows,
                                   const unsigned long long *d_total, const unsigned long long *d_last_kept, int *d_row_offsets)
{
    // d_last_kept: index of the last kept tuple + 1 (0 = none), produced by LastKeptKernel
    const long long edges = static_cast<long long>(*d_total);
    const long long last = static_cast<long long>(*d_last_kept);
    const int last_row = (last == 0) ? -1 : static_cast<int>(d_keys[last - 1] >> col_bits);
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    for (long long r = last_row + 1 + static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; r <= rows; r += stride)
        d_row_offsets[r] = static_cast<int>(edges);
}

static __global__ void LastKeptKernel(const unsigned *d_keep, long long n, unsigned long long *d_last_kept)
{
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    unsigned long long best = 0;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
        if (d_keep[i]) best = static_cast<unsigned long long>(i + 1);
    for (int o = 32; o; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o, util::kWaveSize);
        best = other > best ? other : best;
    }
    if (util::LaneId() == 0 && best) atomicMax(d_last_kept, best);
}

// State of one conversion between its two calls (sort + count, then emit into caller-provided arrays).
struct DeviceCooToCsr {
    unsigned long long *d_keys[2] = {nullptr, nullptr};
    unsigned *d_hist = nullptr;             // radix histograms, then the keep flags
    unsigned long long *d_offsets = nullptr;  // scanned histograms, then the output positions
    unsigned long long *d_sums = nullptr;
    unsigned long long *d_scalars = nullptr;  // [0] total kept, [1] last kept index + 1
    long long tuples = 0;
    long long edges = 0;
    int rows = 0, col_bits = 0, sorted = 0;

    void Release()
    {
        for (int i = 0; i < 2; ++i)
            if (d_keys[i]) util::GRError(hipFree(d_keys[i]), "DeviceCooToCsr hipFree failed", __FILE__, __LINE__);
        if (d_hist) util::GRError(hipFree(d_hist), "DeviceCooToCsr hipFree failed", __FILE__, __LINE__);
        if (d_offsets) util::GRError(hipFree(d_offsets), "DeviceCooToCsr hipFree failed", __FILE__, __LINE__);
        if (d_sums) util::GRError(hipFree(d_sums), "DeviceCooToCsr hipFree failed", __FILE__, __LINE__);
        if (d_scalars) util::GRError(hipFree(d_scalars), "DeviceCooToCsr hipFree failed", __FILE__, __LINE__);
        d_keys[0] = d_keys[1] = nullptr;
        d_hist = nullptr;
        d_offsets = nullptr;
        d_sums = nullptr;
        d_scalars = nullptr;
    }

    // Phase 1: keys, sort, flags, positions.  `rows` = number of CSR rows (local rows for a partition), `nodes` = vertex id
    // space of the columns.  Returns the edge count in `edges` (blocking read).
    hipError_t Sort(int rows_, int nodes, long long pairs, const int *d_rows, const int *d_cols, bool undirected, int parts,
                    int rank, hipStream_t stream)
    {
        hipError_t retval = hipSuccess;
        Release();
        rows = rows_;
        col_bits = 1;
        while ((1ll << col_bits) < nodes) ++col_bits;
        tuples = undirected ? 2 * pairs : pairs;
        edges = 0;
        if (tuples == 0) return retval;
        const long long tiles = (tuples + kSortTile - 1) / kSortTile;
        const long long hist_words = 256 * tiles;
        const long long flag_words = tuples > hist_words ? tuples : hist_words;
        for (int i = 0; i < 2; ++i)
            GR_CHECK(hipMalloc(&d_keys[i], sizeof(unsigned long long) * static_cast<size_t>(tuples)), "DeviceCooToCsr hipMalloc keys failed");
        GR_CHECK(hipMalloc(&d_hist, sizeof(unsigned) * static_cast<size_t>(flag_words)), "DeviceCooToCsr hipMalloc failed");
        GR_CHECK(hipMalloc(&d_offsets, sizeof(unsigned long long) * static_cast<size_t>(flag_words)), "DeviceCooToCsr hipMalloc failed");
        GR_CHECK(hipMalloc(&d_sums, sizeof(unsigned long long) * static_cast<size_t>(ScanScratchWords(flag_words))),
                 "DeviceCooToCsr hipMalloc failed");
        GR_CHECK(hipMalloc(&d_scalars, sizeof(unsigned long long) * 2), "DeviceCooToCsr hipMalloc failed");
        GR_CHECK(hipMemsetAsync(d_scalars, 0, sizeof(unsigned long long) * 2, stream), "DeviceCooToCsr memset failed");

        const unsigned grid = 4096;
        hipLaunchKernelGGL(MakeKeysKernel, dim3(grid), dim3(256), 0, stream, d_rows, d_cols, pairs, undirected ? 1 : 0, col_bits, nodes,
                           parts, rank, d_keys[0]);
        GR_CHECK(hipGetLastError(), "MakeKeysKernel launch failed");

        int cur = 0;
        const int key_bits = 2 * col_bits;
        for (int shift = 0; shift < key_bits; shift += 8) {
            hipLaunchKernelGGL(RadixHistogramKernel, dim3(static_cast<unsigned>(tiles)), dim3(kSortThreads), 0, stream, d_keys[cur], tuples,
                               shift, tiles, d_hist);
            GR_CHECK(hipGetLastError(), "RadixHistogramKernel launch failed");
            GR_CHECK(DeviceExclusiveScan<unsigned long long>(d_hist, d_offsets, hist_words, d_sums, stream), "radix scan failed");
            hipLaunchKernelGGL(RadixScatterKernel, dim3(static_cast<unsigned>(tiles)), dim3(kSortThreads), 0, stream, d_keys[cur], tuples,
                               shift, tiles, d_offsets, d_keys[cur ^ 1]);
            GR_CHECK(hipGetLastError(), "RadixScatterKernel launch failed");
            cur ^= 1;
        }
        sorted = cur;

        const unsigned long long sentinel = (key_bits >= 64) ? ~0ull : ((1ull << key_bits) - 1ull);
        hipLaunchKernelGGL(FlagKernel, dim3(grid), dim3(256), 0, stream, d_keys[sorted], tuples, sentinel, d_hist);
        GR_CHECK(hipGetLastError(), "FlagKernel launch failed");
        GR_CHECK(DeviceExclusiveScan<unsigned long long>(d_hist, d_offsets, tuples, d_sums, stream), "flag scan failed");
        // total = last scanned tile sum slot: ScanSumsKernel left it behind the tile offsets
        const long long scan_tiles = (tuples + kScanTile - 1) / kScanTile;
        GR_CHECK(hipMemcpyAsync(d_scalars, d_sums + scan_tiles, sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream),
                 "DeviceCooToCsr copy total failed");
        hipLaunchKernelGGL(LastKeptKernel, dim3(1024), dim3(256), 0, stream, d_hist, tuples, d_scalars + 1);
        GR_CHECK(hipGetLastError(), "LastKeptKernel launch failed");
        unsigned long long total = 0;
        GR_CHECK(hipMemcpyAsync(&total, d_scalars, sizeof(total), hipMemcpyDeviceToHost, stream), "DeviceCooToCsr read total failed");
        GR_CHECK(hipStreamSynchronize(stream), "DeviceCooToCsr sync failed");
        edges = static_cast<long long>(total);
        return retval;
    }

    // Phase 2: write row_offsets[rows + 1] and col_indices[edges] (device arrays of the caller).
    hipError_t Emit(int *d_row_offsets, int *d_col_indices, hipStream_t stream)
    {
        hipError_t retval = hipSuccess;
        if (tuples == 0 || edges == 0) {
            GR_CHECK(hipMemsetAsync(d_row_offsets, 0, sizeof(int) * static_cast<size_t>(rows + 1), stream), "DeviceCooToCsr memset failed");
            return util::GRError(hipStreamSynchronize(stream), "DeviceCooToCsr sync failed", __FILE__, __LINE__);
        }
        hipLaunchKernelGGL(EmitCsrKernel, dim3(4096), dim3(256), 0, stream, d_keys[sorted], d_hist, d_offsets, tuples, col_bits, rows,
                           edges, d_row_offsets, d_col_indices);
        GR_CHECK(hipGetLastError(), "EmitCsrKernel launch failed");
        hipLaunchKernelGGL(CloseOffsetsKernel, dim3(1024), dim3(256), 0, stream, d_keys[sorted], d_hist, tuples, col_bits, rows, d_scalars,
                           d_scalars + 1, d_row_offsets);
        GR_CHECK(hipGetLastError(), "CloseOffsetsKernel launch failed");
        return util::GRError(hipStreamSynchronize(stream), "DeviceCooToCsr sync failed", __FILE__, __LINE__);
    }
};

}  // namespace graphio
}  // namespace gunrock
