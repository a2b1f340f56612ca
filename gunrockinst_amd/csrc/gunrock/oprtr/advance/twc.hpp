// oprtr/advance/twc.hpp -- thread / wave / workgroup tiered expansion for runs of SMALL levels.
//
// Role of the reference's TWC advance (gunrock/oprtr/edge_map_forward/cta.cuh:224-545, chosen for graphs of average degree
// <= 8 by tests/bfs/test_bfs.cu:563-566): every thread takes a frontier vertex; a short neighbour list is expanded by the
// thread itself, a longer one is handed to its warp, a very long one to the whole CTA.  The reference launches it once per
// BSP level.  On a road-like graph (degree ~4, thousands of levels of a few thousand vertices) a level is nothing but a chain
// of dependent memory round trips, so here the tiers live inside ONE 1024-thread workgroup that keeps the frontier in LDS and
// runs level after level without leaving the CU:
//     per level:  (vertex, first edge, end) of the frontier come from LDS
//                 -> columns of up to 4 edges of EACH of a thread's vertices together (one round trip)
//                 -> the claims of all those edges issued together, and next to them -- speculatively, for every
//                    destination -- its row extent, which the NEXT level needs (one round trip)
//                 -> winners appended to the next frontier in LDS as (vertex, first edge, end)
// No degree prefix, no staged tile, no global queue traffic, no grid barrier, no screen in the thread tier (the claim decides;
// a level here is latency, not throughput): 2 round trips per level where the load-balanced tail kernel (kernel.hpp,
// TailLevelsKernel) has ~10 and the persistent kernel a grid barrier.  The kernel hands back -- frontier written out as a
// complete (vertex, row start, degree prefix) queue through the FrontierWriter -- when a level outgrows the LDS queue or
// its edge budget; the enactor then continues with the load-balanced kernels.
// Measured and dropped: claims without global atomics (plain label test + one discoverer per destination elected through an
// LDS hash set) -- 16 -> 20 us per level on a 1024^2 grid.  What a level costs in ONE workgroup is the CU's rate of
// uncoalesced lane accesses (about one per clock: ~10 per frontier vertex here), so the kernel pays off for frontiers of up to
// ~2000 vertices (1024^2 grid, whole search: 18.9 -> 16.3 us per level against the persistent kernel) and is at par beyond
// (2048^2 grid, frontiers up to 4096); larger levels belong to many CUs and a grid barrier (PersistentLevelsKernel).
// Same TailArgs protocol as TailLevelsKernel: ring slot (iteration & 3) holds the packed tail of the frontier in
// queue[selector]; on return *d_levels_done levels ran, the remaining frontier (if any) is in queue[selector ^ (done & 1)]
// with its tail in slot ((first_iteration + done) & 3), and the two slots behind it are zero.
#pragma once

#include <hip/hip_runtime.h>

#include <gunrock/oprtr/advance/functor_hooks.hpp>
#include <gunrock/oprtr/advance/kernel.hpp>
#include <gunrock/oprtr/frontier_writer.hpp>
#include <gunrock/util/device_intrinsics.hpp>

namespace gunrock {
namespace oprtr {
namespace advance {

constexpr int kTwcThreads = 1024;
constexpr int kTwcCapacity = 4096;   // vertices of a frontier kept in LDS: 2 x (vertex, first edge, end) = 96 KiB
constexpr int kTwcBatch = 8;         // edges a lane keeps in flight in the wave / workgroup tiers
constexpr int kTwcQuad = 4;          // thread tier: edges per vertex and round, for all of a thread's vertices at once
constexpr int kTwcThreadMax = 32;    // neighbour lists up to this length are expanded by their thread
constexpr int kTwcWaveMax = 4096;    // ... up to this length by one wave, longer ones by the workgroup
constexpr int kTwcList = 256;        // deferred lists (wave tier / workgroup tier), entries each

// dword-aligned 16- / 8-byte loads (one memory instruction and one cache-line access per lane instead of four / two)
struct __attribute__((packed, aligned(4))) TwcQuad {
    int v[4];
};
struct __attribute__((packed, aligned(4))) TwcPair {
    int v[2];
};

template <typename VertexId, typename SizeT>
struct TwcShared {
    typedef FrontierWriter<kTwcThreads, 2048, VertexId, SizeT> Writer;
    VertexId q[2][kTwcCapacity];
    SizeT q_b[2][kTwcCapacity];     // first edge / end of the vertex's neighbour list (fetched by whoever discovered it)
    SizeT q_e[2][kTwcCapacity];
    int q_count[2];
    VertexId list_v[2][kTwcList];   // [0] wave tier, [1] workgroup tier
    SizeT list_b[2][kTwcList];
    SizeT list_e[2][kTwcList];
    int list_count[2];
    long long wave_sum[kTwcThreads / util::kWaveSize];
    unsigned long long level_tail;
    typename Writer::Storage writer;
};

// Claim N edges (src[i] -> dst[i], live[i]): on return live[i] = this thread discovered dst[i].  rb / re receive the row extent
// of every destination (fetched next to the claims, because winners need it for the next level).  SCREEN: run the functor's
// side-effect-free pre-test first (long lists: most of a hub's edges lead to discovered vertices); the thread tier skips it --
// there a level is latency, and the claim alone decides.
template <int N, bool SCREEN, typename ProblemData, typename Functor, typename VertexId, typename SizeT>
__device__ __forceinline__ void TwcClaim(const TailArgs<VertexId, SizeT> &t, typename ProblemData::DataSlice &slice, const VertexId (&src)[N],
                                         const VertexId (&dst)[N], const SizeT (&edge)[N], bool (&live)[N], SizeT (&rb)[N], SizeT (&re)[N])
{
    typedef typename ProblemData::DataSlice DataSlice;
    if constexpr (SCREEN) {
#pragma unroll
        for (int i = 0; i < N; ++i) live[i] = live[i] & ScreenEdge<Functor>(src[i], dst[i], &slice, edge[i], edge[i]);
    }
    if constexpr (HasIssueEdge<Functor, VertexId, DataSlice>::value) {
        typedef decltype(Functor::IssueEdge(src[0], dst[0], &slice, edge[0], edge[0])) Token;
        Token token[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            token[i] = Token();
            if (live[i]) token[i] = Functor::IssueEdge(src[i], dst[i], &slice, edge[i], edge[i]);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const TwcPair r2 = *reinterpret_cast<const TwcPair *>(t.d_row_offsets + dst[i]);
            rb[i] = r2.v[0];
            re[i] = r2.v[1];
        }
#pragma unroll
        for (int i = 0; i < N; ++i) live[i] = live[i] && Functor::ResolveEdge(token[i], src[i], dst[i], &slice, edge[i], edge[i]);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const TwcPair r2 = *reinterpret_cast<const TwcPair *>(t.d_row_offsets + dst[i]);
            rb[i] = r2.v[0];
            re[i] = r2.v[1];
            live[i] = live[i] && Functor::CondEdge(src[i], dst[i], &slice, edge[i], edge[i]);
        }
    }
}

// ApplyEdge for the winners and their entries (vertex, first edge, end) in the next frontier: LDS, or -- ids only -- the spill
// area in HBM once LDS is full
template <int N, typename ProblemData, typename Functor, typename VertexId, typename SizeT>
__device__ __forceinline__ void TwcEmit(typename ProblemData::DataSlice &slice, const VertexId (&src)[N], const VertexId (&dst)[N],
                                        const SizeT (&edge)[N], const bool (&live)[N], const SizeT (&rb)[N], const SizeT (&re)[N], VertexId *next,
                                        SizeT *next_b, SizeT *next_e, int *next_count, VertexId *d_spill)
{
    int mine = 0;
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (live[i]) {
            Functor::ApplyEdge(src[i], dst[i], &slice, edge[i], edge[i]);
            ++mine;
        }
    if (mine == 0) return;
    int at = atomicAdd(next_count, mine);
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (live[i]) {
            if (at < kTwcCapacity) {
                next[at] = dst[i];
                next_b[at] = rb[i];
                next_e[at] = re[i];
            } else {
                d_spill[at] = dst[i];
            }
            ++at;
        }
}

// one batch of up to kTwcBatch edges of source s (wave / workgroup tiers): columns at `first + j * stride`
template <typename ProblemData, typename Functor, typename VertexId, typename SizeT>
__device__ __forceinline__ void TwcBatch(const TailArgs<VertexId, SizeT> &t, typename ProblemData::DataSlice &slice, VertexId s, SizeT first,
                                         SizeT end, SizeT stride, VertexId *next, SizeT *next_b, SizeT *next_e, int *next_count,
                                         VertexId *d_spill)
{
    VertexId src[kTwcBatch], dst[kTwcBatch];
    SizeT edge[kTwcBatch], rb[kTwcBatch], re[kTwcBatch];
    bool live[kTwcBatch];
#pragma unroll
    for (int j = 0; j < kTwcBatch; ++j) {
        src[j] = s;
        edge[j] = first + static_cast<SizeT>(j) * stride;
        live[j] = edge[j] < end;
        if (!live[j]) edge[j] = first;  // (a dead slot re-reads the first edge: no branch around the load; first < end always)
        dst[j] = t.d_column_indices[edge[j]];
    }
    TwcClaim<kTwcBatch, true, ProblemData, Functor>(t, slice, src, dst, edge, live, rb, re);
    TwcEmit<kTwcBatch, ProblemData, Functor>(slice, src, dst, edge, live, rb, re, next, next_b, next_e, next_count, d_spill);
}

template <typename ProblemData, typename Functor>
__global__ __launch_bounds__(kTwcThreads) void TwcLevelsKernel(TailArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> t,
                                                               typename ProblemData::DataSlice slice)
{
    typedef typename ProblemData::VertexId VertexId;
    typedef typename ProblemData::SizeT SizeT;
    typedef TwcShared<VertexId, SizeT> Shared;
    typedef typename Shared::Writer Writer;
    constexpr int THREADS = kTwcThreads;
    constexpr int WAVES = THREADS / util::kWaveSize;
    constexpr int PER = kTwcCapacity / THREADS;
    __shared__ Shared sh;

    const int tid = threadIdx.x;
    const int lane = static_cast<int>(util::LaneId());
    const int wave = __builtin_amdgcn_readfirstlane(tid / util::kWaveSize);
    long long iteration = t.first_iteration;
    int done = 0;
    unsigned long long sum_len = 0, sum_edges = 0;

    if (tid == 0) {
        sh.level_tail = __hip_atomic_load(t.d_tail + (iteration & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh.q_count[0] = sh.q_count[1] = 0;
        sh.list_count[0] = sh.list_count[1] = 0;
    }
    Writer::Init(sh.writer);
    __syncthreads();
    int len = static_cast<int>(util::TailCount(sh.level_tail));
    if (len > kTwcCapacity) len = -1;  // does not fit: leave everything as it is (done = 0)
    int cur = 0;
    if (len > 0) {
        const VertexId *d_in = t.queue[t.selector].v;
        for (int i = tid; i < len; i += THREADS) {
            const VertexId u = __hip_atomic_load(d_in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const TwcPair r2 = *reinterpret_cast<const TwcPair *>(t.d_row_offsets + u);
            sh.q[0][i] = u;
            sh.q_b[0][i] = r2.v[0];
            sh.q_e[0][i] = r2.v[1];
        }
    }
    __syncthreads();

    bool publish = false;  // the frontier in sh.q[cur] (len entries) has to go back to the global queue
    while (len > 0) {
        // ---- row extents of this thread's vertices, the level's edge count ----
        VertexId v[PER];
        SizeT b[PER], e[PER];
        long long my_edges = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = k * THREADS + tid;
            v[k] = 0;
            b[k] = e[k] = 0;
            if (i < len) {
                v[k] = sh.q[cur][i];
                b[k] = sh.q_b[cur][i];
                e[k] = sh.q_e[cur][i];
            }
            my_edges += e[k] - b[k];
        }
        const long long wsum = util::WaveSum(my_edges);
        if (lane == 0) sh.wave_sum[wave] = wsum;
        __syncthreads();
        long long edges = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) edges += sh.wave_sum[w];
        if (edges > t.edge_limit || done >= t.max_levels) {  // workgroup-uniform: this level is for the load-balanced kernels
            publish = true;
            break;
        }
        slice.iteration = static_cast<VertexId>(iteration);
        VertexId *next = sh.q[cur ^ 1];
        SizeT *next_b = sh.q_b[cur ^ 1], *next_e = sh.q_e[cur ^ 1];
        int *next_count = &sh.q_count[cur ^ 1];
        VertexId *d_spill = t.queue[t.selector ^ (done & 1)].v;  // (free: this level's input is in LDS, its output goes to the other queue)

        // ---- longer lists are deferred to the wave / workgroup tiers ----
        SizeT pos[PER];  // thread tier: next edge of vertex k (== e[k]: nothing left for this thread)
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const SizeT deg = e[k] - b[k];
            pos[k] = b[k];
            const int tier = deg <= kTwcThreadMax ? 0 : (deg <= kTwcWaveMax ? 1 : 2);
            if (tier) {
                const int at = atomicAdd(&sh.list_count[tier - 1], 1);
                if (at < kTwcList) {
                    sh.list_v[tier - 1][at] = v[k];
                    sh.list_b[tier - 1][at] = b[k];
                    sh.list_e[tier - 1][at] = e[k];
                    pos[k] = e[k];
                }  // (list full: the thread walks the list itself, a quad per round -- slow, correct)
            }
        }
        // ---- thread tier: a quad of edges of EVERY vertex of the thread per round, all loads of a round in flight together ----
        for (;;) {
            bool any = false;
#pragma unroll
            for (int k = 0; k < PER; ++k) any |= pos[k] < e[k];
            if (!any) break;
            constexpr int N = PER * kTwcQuad;
            VertexId src[N], dst[N];
            SizeT edge[N], rb[N], re[N];
            bool live[N];
            static_assert(sizeof(VertexId) == 4 && sizeof(SizeT) == 4 && kTwcQuad == 4, "32-bit ids, quads");
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                if (pos[k] + kTwcQuad <= e[k]) {  // a whole quad: one 16-byte load
                    const TwcQuad q4 = *reinterpret_cast<const TwcQuad *>(t.d_column_indices + pos[k]);
#pragma unroll
                    for (int j = 0; j < kTwcQuad; ++j) {
                        live[k * kTwcQuad + j] = true;
                        dst[k * kTwcQuad + j] = q4.v[j];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < kTwcQuad; ++j) {
                        live[k * kTwcQuad + j] = pos[k] + j < e[k];
                        dst[k * kTwcQuad + j] = t.d_column_indices[live[k * kTwcQuad + j] ? pos[k] + j : static_cast<SizeT>(0)];
                    }
                }
#pragma unroll
                for (int j = 0; j < kTwcQuad; ++j) {
                    src[k * kTwcQuad + j] = v[k];
                    edge[k * kTwcQuad + j] = live[k * kTwcQuad + j] ? pos[k] + j : static_cast<SizeT>(0);
                }
            }
            TwcClaim<N, false, ProblemData, Functor>(t, slice, src, dst, edge, live, rb, re);
            TwcEmit<N, ProblemData, Functor>(slice, src, dst, edge, live, rb, re, next, next_b, next_e, next_count, d_spill);
#pragma unroll
            for (int k = 0; k < PER; ++k) pos[k] = pos[k] + kTwcQuad < e[k] ? pos[k] + kTwcQuad : e[k];
        }
        __syncthreads();
        // ---- wave tier: one list per wave at a time, lanes stride it ----
        {
            const int n_wave = sh.list_count[0] < kTwcList ? sh.list_count[0] : kTwcList;
            for (int idx = wave; idx < n_wave; idx += WAVES) {
                const VertexId s = sh.list_v[0][idx];
                const SizeT rb = sh.list_b[0][idx], re = sh.list_e[0][idx];
                for (SizeT off = rb + lane; off < re; off += util::kWaveSize * kTwcBatch)
                    TwcBatch<ProblemData, Functor>(t, slice, s, off, re, static_cast<SizeT>(util::kWaveSize), next, next_b, next_e, next_count, d_spill);
            }
        }
        // ---- workgroup tier: every thread strides every such list ----
        {
            const int n_cta = sh.list_count[1] < kTwcList ? sh.list_count[1] : kTwcList;
            for (int idx = 0; idx < n_cta; ++idx) {
                const VertexId s = sh.list_v[1][idx];
                const SizeT rb = sh.list_b[1][idx], re = sh.list_e[1][idx];
                for (SizeT off = rb + tid; off < re; off += THREADS * kTwcBatch)
                    TwcBatch<ProblemData, Functor>(t, slice, s, off, re, static_cast<SizeT>(THREADS), next, next_b, next_e, next_count, d_spill);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // spilled ids (if any) must be in L2 before they are read back
        __syncthreads();
        sum_len += static_cast<unsigned long long>(len);
        sum_edges += static_cast<unsigned long long>(edges);
        ++iteration;
        ++done;
        const int produced = sh.q_count[cur ^ 1];
        __syncthreads();
        if (tid == 0) {
            sh.q_count[cur] = 0;
            sh.list_count[0] = sh.list_count[1] = 0;
        }
        cur ^= 1;
        len = produced;
        if (produced > kTwcCapacity) {  // the next frontier outgrew LDS: ids [0, capacity) are here, the rest was spilled
            publish = true;
            break;
        }
        __syncthreads();
    }

    // ---- hand over: slot (iteration & 3) = tail of what is left, the two slots behind it zero ----
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(t.d_tail + (iteration & 3), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(t.d_tail + ((iteration + 1) & 3), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(t.d_tail + ((iteration + 2) & 3), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (len < 0) {  // nothing ran: the input frontier and its tail stay where they were
        if (tid == 0) __hip_atomic_store(t.d_tail + (iteration & 3), sh.level_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (publish && len > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // complete frontier (vertex, row start, degree prefix; vertices without out-edges dropped) into the queue the host
        // expects, in chunks of the writer's staging capacity: first the LDS part, then the spilled part
        const util::Frontier<VertexId, SizeT> out = t.queue[t.selector ^ (done & 1)];
        const VertexId *d_spilled = t.queue[t.selector ^ (done & 1) ^ 1].v;
        constexpr int CHUNK = 2048;
        for (int c = 0; c < len; c += CHUNK) {
            const int n = len - c < CHUNK ? len - c : CHUNK;
            for (int i = tid; i < n; i += THREADS) {
                const int at = c + i;
                sh.writer.buf[i] = at < kTwcCapacity ? sh.q[cur][at]
                                                      : __hip_atomic_load(d_spilled + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            Writer::template Flush<true>(sh.writer, n, out, t.d_tail + (iteration & 3), t.d_overflow, t.d_row_offsets);
            __syncthreads();
        }
    }
    if (tid == 0) {
        *t.d_levels_done = done;
        t.d_level_sums[0] = sum_len;
        t.d_level_sums[1] = sum_edges;
    }
}

template <typename ProblemData, typename Functor>
hipError_t LaunchTwcLevels(const TailArgs<typename ProblemData::VertexId, typename ProblemData::SizeT> &args,
                           const typename ProblemData::DataSlice &slice, hipStream_t stream)
{
    hipLaunchKernelGGL((TwcLevelsKernel<ProblemData, Functor>), dim3(1), dim3(kTwcThreads), 0, stream, args, slice);
    return util::GRError("advance::TwcLevelsKernel launch failed", __FILE__, __LINE__);
}

}  // namespace advance
}  // namespace oprtr
}  // namespace gunrock
