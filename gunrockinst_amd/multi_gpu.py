"""Vertex-partitioned multi-GPU BFS: one process per GPU, collectives over torch.distributed (RCCL on xGMI).

The reference is single-GPU (gunrock/app/problem_base.cuh:336-338 is a TODO).  This layer keeps its striped
ownership rule -- owner = v mod P, local id = v div P (problem_base.cuh:185-210) -- and adds the per-level exchange:

  top-down level : local advance -> ids bucketed by owner -> all_to_all_single(counts) + all_to_all_single(ids)
                   -> local filter (claim, label, next frontier)
  bottom-up level: all_gather of the per-rank frontier bitmaps (n/8 bytes in total) -> local bottom-up sweep
  every level    : one all_reduce of (frontier vertices, frontier edges) for termination and direction choice

xGMI is point-to-point: the all-to-all keeps all 7 links of a GPU busy at once, and the dense levels of an R-MAT
search move only bitmaps (2 MiB per level at scale-24), never ids.

Compute never happens here: `engine` performs the local steps.  HipEngine drives the HIP kernels through the C ABI
(grx_pbfs_*).  The level loop is engine-agnostic so CPU tests can run it over gloo with a numpy test double.
"""
import ctypes as C
import os
import time

import numpy as np
import torch
import torch.distributed as dist


# ------------------------------------------------------------------------------------------------------------------
# ownership rule (reference problem_base.cuh:185-210)
# ------------------------------------------------------------------------------------------------------------------
def owner_of(v, parts):
    return v % parts


def local_id(v, parts):
    return v // parts


def local_count(n_global, parts, rank):
    return (n_global - rank + parts - 1) // parts if n_global > rank else 0


def mask_words(n):
    return ((n + 63) // 64) * 2


# ------------------------------------------------------------------------------------------------------------------
# collectives (device tensors on RCCL; staged through the host when the backend is gloo)
# ------------------------------------------------------------------------------------------------------------------
class Comm:
    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.host_staged = self.backend == "gloo"

    def _stage(self, t):
        return t.cpu() if (self.host_staged and t.is_cuda) else t

    def _fence(self):
        """RCCL work is enqueued on torch's stream; the engine's kernels run on their own HIP stream.  Make the collective's
        output visible to them: wait on the host until torch's current stream has drained.  (The engine side always returns
        to Python with its stream synchronised, so the other direction needs nothing.)"""
        if not self.host_staged:
            torch.cuda.current_stream().synchronize()

    def all_reduce_sum(self, values):
        t = torch.tensor(values, dtype=torch.int64)
        if not self.host_staged:
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return [int(x) for x in t.tolist()]

    def all_reduce_max(self, values):
        t = torch.tensor(values, dtype=torch.int64)
        if not self.host_staged:
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return [int(x) for x in t.tolist()]

    def exchange_counts(self, send_counts):
        if self.world == 1:  # (nothing to exchange; RCCL's all-to-all on a one-rank communicator divides by zero)
            return [int(x) for x in send_counts]
        s = torch.tensor(send_counts, dtype=torch.int64)
        if not self.host_staged:
            s = s.cuda()
        r = torch.empty_like(s)
        dist.all_to_all_single(r, s, group=self.group)
        return [int(x) for x in r.tolist()]

    def all_to_all_v(self, send, send_counts, recv_counts):
        """send: int32 tensor, segments in rank order.  Returns the received int32 tensor on send's device."""
        if self.world == 1:
            return send
        dev = send.device
        s = self._stage(send)
        r = torch.empty(int(sum(recv_counts)), dtype=torch.int32, device=s.device)
        dist.all_to_all_single(r, s, list(recv_counts), list(send_counts), group=self.group)
        self._fence()
        return r.to(dev) if r.device != dev else r

    def all_gather(self, t):
        dev = t.device
        s = self._stage(t)
        out = torch.empty(self.world * s.numel(), dtype=s.dtype, device=s.device)
        dist.all_gather_into_tensor(out, s, group=self.group)
        self._fence()
        return out.to(dev) if out.device != dev else out

    def barrier(self):
        dist.barrier(group=self.group)


# ------------------------------------------------------------------------------------------------------------------
# the level loop (engine-agnostic)
# ------------------------------------------------------------------------------------------------------------------
class PartitionedBfs:
    """engine must provide: reset(src)->(len, edges); advance_local()->(send_counts, send_tensor);
    filter_received(recv_tensor)->(len, edges); queue_to_bitmap(); frontier_bitmap()->int32 tensor;
    bottom_up(gathered_tensor, words_per_rank)->(len, edges); bitmap_to_queue()->(len, edges)."""

    def __init__(self, engine, comm, n_global, m_global, alpha=10.0, beta=24.0):
        self.engine, self.comm = engine, comm
        self.n_global, self.m_global = int(n_global), int(m_global)
        self.alpha, self.beta = float(alpha), float(beta)
        self.trace = []
        self.profile = {} if os.environ.get("GUNROCK_PBFS_PROFILE") == "1" else None
        self._send = None

    def run(self, src, direction_optimizing=True, sticky_bottom_up=False):
        """sticky_bottom_up: once the direction rule turns the search bottom-up it stays bottom-up to the end, on the
        one-collective-per-level loop of _gather_levels (no all-reduce, no conversion back, no id exchange for the last
        levels; those sweeps are cheap because almost nothing is unvisited by then)."""
        eng, comm = self.engine, self.comm
        self.trace = []
        prof = self.profile          # None, or a dict phase -> seconds (GUNROCK_PBFS_PROFILE=1)
        clock = time.perf_counter

        def timed(name, fn, *a):
            if prof is None:
                return fn(*a)
            t0 = clock()
            out = fn(*a)
            prof[name] = prof.get(name, 0.0) + clock() - t0
            return out

        local_len, local_edges = timed("reset", eng.reset, src)
        glen, gedges = timed("all_reduce", comm.all_reduce_sum, [local_len, local_edges])
        unexplored = self.m_global
        bottom_up = False
        levels = 0
        while glen > 0:
            if direction_optimizing and not bottom_up and gedges * self.alpha > unexplored:
                timed("queue_to_bitmap", eng.queue_to_bitmap)
                bottom_up = True
                if sticky_bottom_up:
                    return levels + self._gather_levels(local_len, timed)
            elif direction_optimizing and bottom_up and glen * self.beta < self.n_global:
                glen, gedges = timed("all_reduce", comm.all_reduce_sum, list(timed("bitmap_to_queue", eng.bitmap_to_queue)))
                bottom_up = False
                if glen == 0:
                    break
            unexplored -= gedges
            if bottom_up:
                bitmap = timed("frontier_bitmap", eng.frontier_bitmap)
                gathered = timed("all_gather", comm.all_gather, bitmap)
                l, e = timed("bottom_up", eng.bottom_up, gathered, bitmap.numel())
            else:
                send_counts, send = timed("advance_local", eng.advance_local)
                recv_counts = timed("exchange_counts", comm.exchange_counts, send_counts)
                recv = timed("all_to_all_v", comm.all_to_all_v, send, send_counts, recv_counts)
                l, e = timed("filter_received", eng.filter_received, recv)
            self.trace.append(("bottom-up" if bottom_up else "top-down", glen, gedges))
            local_len = l
            glen, gedges = timed("all_reduce", comm.all_reduce_sum, [l, e])
            levels += 1
        return levels


    def _gather_levels(self, local_len, timed):
        """Bottom-up levels until the frontier is empty; each level's ONE collective -- the all-gather of the per-rank
        frontier bitmaps -- also carries each rank's frontier size in a trailing word, so termination needs no all-reduce."""
        eng, comm = self.engine, self.comm
        prof, clock = self.profile, time.perf_counter
        levels = 0
        while True:
            bitmap = timed("frontier_bitmap", eng.frontier_bitmap)
            wpr = int(bitmap.numel())
            if self._send is None or self._send.numel() != wpr + 2 or self._send.device != bitmap.device:
                self._send = torch.zeros(wpr + 2, dtype=torch.int32, device=bitmap.device)
            t0 = clock()
            self._send[:wpr].copy_(bitmap)
            self._send[wpr] = int(local_len)
            gathered = comm.all_gather(self._send)
            total = int(gathered.view(comm.world, wpr + 2)[:, wpr].sum())
            if prof is not None:
                prof["all_gather"] = prof.get("all_gather", 0.0) + clock() - t0
            if total == 0:
                break
            self.trace.append(("bottom-up", total, 0))
            local_len, _ = timed("bottom_up", eng.bottom_up, gathered, wpr + 2)
            levels += 1
        return levels

    def _timer(self):
        prof, clock = self.profile, time.perf_counter

        def timed(name, fn, *a):
            if prof is None:
                return fn(*a)
            t0 = clock()
            out = fn(*a)
            prof[name] = prof.get(name, 0.0) + clock() - t0
            return out
        return timed

    def run_gather(self, src):
        """Bitmap-gather schedule from the first level on: every level is a bottom-up sweep.  Correct, one collective per
        level -- and useless on a large graph: the first levels (a frontier of one vertex) make every unvisited vertex walk
        its whole in-list (measured at scale-24 on one rank: 62 ms per search in the sweeps).  Kept for tests and tiny
        graphs; run() switches to this loop only after the top-down levels (`sticky_bottom_up`)."""
        timed = self._timer()
        self.trace = []
        local_len, _ = timed("reset", self.engine.reset, src)
        timed("queue_to_bitmap", self.engine.queue_to_bitmap)
        return self._gather_levels(local_len, timed)


# ------------------------------------------------------------------------------------------------------------------
# HIP engine (C ABI)
# ------------------------------------------------------------------------------------------------------------------
class HipEngine:
    def __init__(self, n_global, parts, rank, d_row_offsets, d_col_indices, device_index=0):
        from . import capi, devgraph
        self._capi, self._dg = capi, devgraph
        self.lib = capi.lib()
        self.n_global, self.parts, self.rank = n_global, parts, rank
        self.ro, self.ci = d_row_offsets, d_col_indices          # torch int32 tensors (kept alive here)
        self.n_local = int(d_row_offsets.shape[0]) - 1
        self.m_local = int(d_col_indices.shape[0])
        self._h = C.c_void_p()
        self._check(self.lib.grx_pbfs_create(C.byref(self._h), device_index), "grx_pbfs_create")
        self._check(self.lib.grx_pbfs_init_device(self._h, n_global, parts, rank, self.n_local, self.m_local,
                                                  C.c_void_p(d_row_offsets.data_ptr()),
                                                  C.c_void_p(d_col_indices.data_ptr())), "grx_pbfs_init_device")
        self._counts = (C.c_uint32 * 64)()
        self.device = d_row_offsets.device

    @staticmethod
    def _check(rc, what):
        if rc != 0:
            raise RuntimeError("gunrockinst_amd: %s failed (code %d)" % (what, rc))

    def _pair(self, fn, *args):
        a, b = C.c_uint32(), C.c_uint32()
        self._check(fn(self._h, *args, C.byref(a), C.byref(b)), fn.__name__)
        return int(a.value), int(b.value)

    def reset(self, src):
        self._check(self.lib.grx_pbfs_reset(self._h, int(src)), "grx_pbfs_reset")
        return self._pair(self.lib.grx_pbfs_frontier)

    def advance_local(self):
        buf = C.c_void_p()
        self._check(self.lib.grx_pbfs_advance_local(self._h, self._counts, C.byref(buf)), "grx_pbfs_advance_local")
        counts = [int(self._counts[i]) for i in range(self.parts)]
        return counts, self._dg.as_tensor(buf.value, sum(counts), device=self.device)

    def filter_received(self, recv):
        recv = recv.contiguous()
        self._keep = recv
        return self._pair(self.lib.grx_pbfs_filter_received, C.c_void_p(recv.data_ptr() if recv.numel() else None),
                          int(recv.numel()))

    def queue_to_bitmap(self):
        self._check(self.lib.grx_pbfs_queue_to_bitmap(self._h), "grx_pbfs_queue_to_bitmap")

    def frontier_bitmap(self):
        ptr, words = C.c_void_p(), C.c_int()
        self._check(self.lib.grx_pbfs_frontier_bitmap(self._h, C.byref(ptr), C.byref(words)), "grx_pbfs_frontier_bitmap")
        return self._dg.as_tensor(ptr.value, words.value, device=self.device)

    def bottom_up(self, gathered, words_per_rank):
        gathered = gathered.contiguous()
        self._keep = gathered
        return self._pair(self.lib.grx_pbfs_bottom_up, C.c_void_p(gathered.data_ptr()), int(words_per_rank))

    def bitmap_to_queue(self):
        return self._pair(self.lib.grx_pbfs_bitmap_to_queue)

    def labels_tensor(self):
        ptr = C.c_void_p()
        self._check(self.lib.grx_pbfs_labels(self._h, C.byref(ptr)), "grx_pbfs_labels")
        return self._dg.as_tensor(ptr.value, self.n_local, device=self.device)

    def labels(self):
        return self.labels_tensor().cpu().numpy()

    def close(self):
        if self._h:
            self.lib.grx_pbfs_destroy(self._h)
            self._h = None


# ------------------------------------------------------------------------------------------------------------------
# the level loop INSIDE the library (grx_pbfs_search): one call per search, RCCL issued from C++ on the engine's stream
# ------------------------------------------------------------------------------------------------------------------
_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
_A2A_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p,
                      C.POINTER(C.c_size_t), C.POINTER(C.c_size_t))


class LibraryBfs:
    """Vertex-partitioned BFS with the level loop in C++ (csrc/lib/partitioned_bfs.hip, Pbfs::Search).

    transport = "rccl": the library creates its own RCCL communicator (the 128-byte id travels over torch.distributed
    once, at construction) and issues every collective itself -- nothing of a search runs in Python.
    transport = "callbacks": the same C++ loop, the three exchanges handed back to torch.distributed (gloo, host-staged);
    for tests that run several ranks on one GPU, where RCCL cannot be used."""

    def __init__(self, engine, comm, transport="rccl", mark_pred=False, alpha=0.0):
        self.engine, self.comm = engine, comm
        self.lib = engine.lib
        self._h = engine._h
        if transport in ("rccl", "rccl-or-callbacks"):
            # every rank must end up on the same transport: the outcome of each step is agreed on before the next one
            # step 1 is local (dlopen + dlsym): if ANY rank cannot load RCCL, no rank may enter ncclCommInitRank -- that call is
            # collective, and the ranks that did load would wait in it forever for the ones that already gave up
            (rc,) = comm.all_reduce_max([abs(int(self.lib.grx_rccl_load()))])
            box = [b"", int(rc)]
            if rc == 0:
                buf = C.create_string_buffer(128)
                rc = self.lib.grx_rccl_unique_id(buf) if comm.rank == 0 else 0
                box = [bytes(buf.raw), int(rc)]
                dist.broadcast_object_list(box, src=0, group=comm.group)
                rc = box[1]
            if rc == 0:
                rc = self.lib.grx_pbfs_comm_init_rccl(self._h, box[0])
                (rc,) = comm.all_reduce_max([abs(int(rc))])
            if rc != 0:
                if transport == "rccl":
                    raise RuntimeError("gunrockinst_amd: the library could not create its RCCL communicator (code %d)" % rc)
                transport = "callbacks"  # (the same C++ loop; the exchanges go through torch.distributed)
            else:
                transport = "rccl"
        if transport == "callbacks":
            self._gather_cb = _GATHER_FN(self._all_gather)
            self._a2a_cb = _A2A_FN(self._all_to_all_v)
            HipEngine._check(self.lib.grx_pbfs_set_transport(self._h, None, C.cast(self._gather_cb, C.c_void_p),
                                                             C.cast(self._a2a_cb, C.c_void_p)), "grx_pbfs_set_transport")
        HipEngine._check(self.lib.grx_pbfs_set_options(self._h, int(bool(mark_pred)), float(alpha)), "grx_pbfs_set_options")
        self.transport = transport

    # ---- callback transport: device pointers in, torch.distributed (host-staged when the backend is gloo) in between ----
    def _tensor(self, ptr, words):
        return self.engine._dg.as_tensor(ptr, int(words), device=self.engine.device)

    def _all_gather(self, ctx, d_send, d_recv, words):
        try:
            out = self.comm.all_gather(self._tensor(d_send, words))
            self._tensor(d_recv, words * self.comm.world).copy_(out)
            torch.cuda.synchronize()
            return 0
        except Exception as exc:  # a Python exception must not unwind through the C frame
            print("gunrockinst_amd: all_gather callback failed:", exc, flush=True)
            return 1

    def _all_to_all_v(self, ctx, d_send, sc, so, d_recv, rc, ro):
        try:
            world = self.comm.world
            scl = [int(sc[p]) for p in range(world)]
            rcl = [int(rc[p]) for p in range(world)]
            # segments are contiguous in rank order on both sides (offsets = prefix sums of the counts)
            send = self._tensor(d_send, sum(scl)) if sum(scl) else torch.empty(0, dtype=torch.int32, device=self.engine.device)
            recv = self.comm.all_to_all_v(send, scl, rcl)
            if sum(rcl):
                self._tensor(d_recv, sum(rcl)).copy_(recv)
            torch.cuda.synchronize()
            return 0
        except Exception as exc:
            print("gunrockinst_amd: all_to_all_v callback failed:", exc, flush=True)
            return 1

    def search(self, src, direction_optimizing=True):
        """Returns (levels, elapsed_ms on this rank)."""
        levels, ms = C.c_int(), C.c_float()
        HipEngine._check(self.lib.grx_pbfs_search(self._h, int(src), int(bool(direction_optimizing)), C.byref(levels), C.byref(ms)),
                         "grx_pbfs_search")
        return int(levels.value), float(ms.value)

    def set_option(self, name, value):
        """Named tuning knob of the level loop (grx_pbfs_set_option): "lite_factor", "alpha", "sparse_sweep_div"."""
        HipEngine._check(self.lib.grx_pbfs_set_option(self._h, name.encode(), float(value)), "grx_pbfs_set_option(%s)" % name)
        return self

    def stat(self, name):
        return int(self.lib.grx_pbfs_stat(self._h, name.encode()))

    def preds(self):
        ptr = C.c_void_p()
        HipEngine._check(self.lib.grx_pbfs_preds(self._h, C.byref(ptr)), "grx_pbfs_preds")
        return self.engine._dg.as_tensor(ptr.value, self.engine.n_local, device=self.engine.device).cpu().numpy()


# ------------------------------------------------------------------------------------------------------------------
# partitioned graph construction on the device
# ------------------------------------------------------------------------------------------------------------------
def partition_rmat_device(scale, edge_factor, seed, rank, parts, device="cuda"):
    """Each rank generates the whole seeded tuple stream, and the library's device COO -> CSR step keeps the directed tuples
    whose SOURCE the rank owns (grx_coo_to_csr_sort with parts / rank): local row ids, global column ids, Csr::FromCoo's
    graph semantics."""
    from . import devgraph
    rows, cols = devgraph.rmat_tuples_device(scale, edge_factor << scale, seed, device=device)
    return devgraph.csr_from_tuples_device(1 << scale, rows, cols, undirected=True, parts=parts, rank=rank)


def partition_rmat_exchange(scale, edge_factor, seed, comm, device="cuda"):
    """The same slice as partition_rmat_device without the P-fold redundant generation and sort: the seeded generator is a pure
    function of the pair index (grx_rmat_seeded_device(first, count)), so rank r generates pairs [r * ceil(pairs / P), ...) only,
    turns every pair (a, b) into the directed tuples a -> b and b -> a (the graph is mirrored), buckets them by the OWNER of the
    source (v mod P) and hands them over in one all-to-all; the library's COO -> CSR step then sees only tuples this rank owns
    (self loops and duplicates are dropped there, as Csr::FromCoo does).  Returns (row_offsets, col_indices) of the local rows."""
    from . import devgraph
    parts, rank = comm.world, comm.rank
    pairs = edge_factor << scale
    per = (pairs + parts - 1) // parts
    first = min(rank * per, pairs)
    count = min(per, pairs - first)
    if count > 0:
        rows, cols = devgraph.rmat_tuples_device(scale, count, seed, first=first, device=device)
    else:
        rows = cols = torch.empty(0, dtype=torch.int32, device=device)
    src = torch.cat([rows, cols])
    dst = torch.cat([cols, rows])
    del rows, cols
    if parts == 1:
        return devgraph.csr_from_tuples_device(1 << scale, src, dst, undirected=False, parts=1, rank=0)
    owner = src % parts
    # (per-owner counts by comparison, not torch.bincount: on this ROCm build bincount raises SIGFPE for inputs of 2^28 elements)
    send_counts = [int((owner == q).sum()) for q in range(parts)]
    order = torch.argsort(owner, stable=True)
    del owner
    src, dst = src[order].contiguous(), dst[order].contiguous()
    del order
    recv_counts = comm.exchange_counts(send_counts)
    got_src = comm.all_to_all_v(src, send_counts, recv_counts)
    got_dst = comm.all_to_all_v(dst, send_counts, recv_counts)
    del src, dst
    return devgraph.csr_from_tuples_device(1 << scale, got_src, got_dst, undirected=False, parts=parts, rank=rank)


def partition_csr_host(row_offsets, col_indices, rank, parts):
    """Split a host CSR (numpy) by the striped rule; used by tests on small graphs."""
    n = row_offsets.shape[0] - 1
    mine = np.arange(rank, n, parts)
    deg = (row_offsets[1:] - row_offsets[:-1])[mine]
    ro = np.concatenate(([0], np.cumsum(deg))).astype(np.int32)
    idx = np.concatenate([np.arange(row_offsets[v], row_offsets[v + 1]) for v in mine]) if mine.size and deg.sum() else \
        np.empty(0, np.int64)
    return ro, col_indices[idx.astype(np.int64)].astype(np.int32)


def assemble_labels(comm, local_labels, n_global):
    """Gather owner-striped labels to every rank (Extract for the partitioned problem)."""
    parts = comm.world
    n_max = (n_global + parts - 1) // parts
    pad = torch.full((n_max,), -2, dtype=torch.int32)
    pad[:local_labels.shape[0]] = torch.as_tensor(local_labels, dtype=torch.int32)
    if not comm.host_staged:
        pad = pad.cuda()
    out = torch.empty(parts * n_max, dtype=torch.int32, device=pad.device)
    dist.all_gather_into_tensor(out, pad, group=comm.group)
    out = out.cpu().numpy().reshape(parts, n_max)
    full = np.empty(n_global, dtype=np.int32)
    for r in range(parts):
        full[r::parts] = out[r, :local_count(n_global, parts, r)]
    return full


# ------------------------------------------------------------------------------------------------------------------
# bench leg for N > 1 (called by bench.py under torch.distributed.run)
# ------------------------------------------------------------------------------------------------------------------
def bench(args, rank, world, local_rank, checker=None):
    """checker(full_labels, source, n, m_global) -> (parity, cpu_baseline): supplied by bench.py and called on rank 0 only, AFTER the
    timed region (it is where the CPU oracle lives: this package never imports it)."""
    from . import devgraph
    comm = Comm()
    n = 1 << args.scale
    t0 = time.time()
    if os.environ.get("GUNROCK_PBFS_INGEST") == "redundant":   # every rank generates and sorts the whole tuple stream
        ro, ci = partition_rmat_device(args.scale, args.edge_factor, args.seed, rank, world)
    else:                                                       # each rank its share of the stream, one all-to-all by owner
        ro, ci = partition_rmat_exchange(args.scale, args.edge_factor, args.seed, comm)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    deg = (ro[1:] - ro[:-1]).long()
    m_local = int(ci.shape[0])
    (m_global,) = comm.all_reduce_sum([m_local])

    # sources: first vertex of maximal degree + 64 seeded vertices with degree > 0 (same rule as the 1-GPU leg)
    local_max = int(deg.max()) if deg.numel() else 0
    (gmax,) = comm.all_reduce_max([local_max])
    cand = torch.nonzero(deg == gmax)
    first_local = int(cand[0]) * world + rank if cand.numel() else n
    (neg_src0,) = comm.all_reduce_max([-first_local])
    src0 = -neg_src0
    picks, x = [], args.seed
    while len(picks) < 4096:
        x = devgraph._splitmix64(x)
        picks.append(x % n)
    mine = [int(deg[v // world]) if v % world == rank else 0 for v in picks]
    degs = comm.all_reduce_sum(mine)
    sources = [src0] + [v for v, d in zip(picks, degs) if d > 0][:64]

    eng = HipEngine(n, world, rank, ro, ci, local_rank)
    # The level loop runs inside the library (Pbfs::Search): RCCL issued from C++ on the engine's stream.  With
    # GUNROCK_DIST_BACKEND=gloo (several ranks rehearsing on one GPU) the same loop hands its exchanges back to gloo.
    # GUNROCK_PBFS_LOOP=python: the step-wise model of the protocol (PartitionedBfs), collectives through torch.distributed.
    python_loop = os.environ.get("GUNROCK_PBFS_LOOP") == "python"
    if python_loop:
        bfs = PartitionedBfs(eng, comm, n, m_global)

        def search(s):
            return bfs.run(s, True, sticky_bottom_up=True)
    else:
        bfs = LibraryBfs(eng, comm, transport="callbacks" if comm.host_staged else "rccl-or-callbacks", mark_pred=False)
        for kv in os.environ.get("GUNROCK_PBFS_OPTIONS", "").split(","):  # tuning experiments: "lite_factor=0,alpha=30"
            if "=" in kv:
                bfs.set_option(kv.split("=")[0], float(kv.split("=")[1]))

        def search(s):
            return bfs.search(s, True)[0]
    for k in range(args.warmup):
        search(sources[k % len(sources)])
    torch.cuda.synchronize()
    comm.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        search(sources[k % len(sources)])
    torch.cuda.synchronize()
    comm.barrier()
    wall = time.perf_counter() - t0
    (wall_us,) = comm.all_reduce_max([int(wall * 1e6)])
    wall = wall_us / 1e6

    used = [sources[k % len(sources)] for k in range(args.steps)]
    per_src = {}
    labels_t = eng.labels_tensor()
    for s in sorted(set(used)):
        depth = search(s)
        vis = labels_t > -1
        nv, ev = comm.all_reduce_sum([int(vis.sum()), int(deg[vis].sum())])
        per_src[s] = (nv, ev, depth)
    edges_total = sum(per_src[s][1] for s in used)
    nodes_total = sum(per_src[s][0] for s in used)

    # parity + CPU baseline: the caller's checker (bench.py: the CPU oracle's serial BFS over the whole graph) on rank 0
    search(sources[0])
    full = assemble_labels(comm, eng.labels(), n)
    parity, cpu = None, None
    if rank == 0 and checker is not None:
        parity, cpu = checker(full, sources, n, m_global)
    bfs_runs = args.warmup + args.steps + len(set(used)) + 1
    eng.close()

    balg = 4.0 * edges_total + 20.0 * nodes_total
    return {
        "metric": "MTEPS (million traversed edges/sec) BFS R-MAT scale-%d" % args.scale,
        "value": round(edges_total / (wall * 1e6), 2), "unit": "MTEPS", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(wall * 1e3 / args.steps, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "BFS direction-optimizing, R-MAT scale-%d (%d pairs/vertex mirrored, seed 0x%x) "
                               "vertex-partitioned over %d GPUs (owner = v mod %d), %s: "
                               "n=%d, m=%d" % (args.scale, args.edge_factor, args.seed, world, world,
                                               "level loop in C++ inside the library; top-down levels: RCCL count all-gather + grouped send/recv of ids; from the first "
                                               "bottom-up level on one all-gather of the frontier bitmaps per level (sizes ride along)" if not python_loop
                                               else "level loop in Python over torch.distributed (protocol model)", n, m_global),
                   "levels_src0": per_src[used[0]][2], "graph_build_s": round(build_s, 2), "backend": comm.backend,
                   "transport": "python loop" if python_loop else bfs.transport,
                   "count_only_levels_per_search": None if python_loop else round(bfs.stat("marked_levels") / max(bfs_runs, 1), 2)},
        "edges_visited_per_step": edges_total // args.steps, "nodes_visited_per_step": nodes_total // args.steps,
        "parity_vs_oracle": parity,
        "level_loop_profile_ms_per_step": ({k: round(v * 1e3 / max(bfs_runs, 1), 4) for k, v in sorted(bfs.profile.items())}
                                           if getattr(bfs, "profile", None) is not None else None),
        "roofline": {"bound": "hbm", "achieved": round(balg / wall / 1e9, 2), "peak": 8000.0 * world, "unit": "GB/s",
                     "frac": round(balg / wall / 1e9 / (8000.0 * world), 5), "traffic": None,
                     "note": "whole-step wall time (kernels + collectives), all ranks"},
        "cpu_baseline": cpu,
    }
