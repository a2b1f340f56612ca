#!/usr/bin/env python3
"""bench.py -- BFS MTEPS on R-MAT scale-24 (BASELINE.json metric), 1..N MI355X.

A "step" is one complete BFS pass (Problem::Reset + Enactor::Enact) from one source over the synthetic
R-MAT graph, which is already resident in HBM when the timed region starts.  Sources cycle through the
largest-degree vertex and 64 seeded non-isolated vertices.

  value        = sum(edges_visited) / wall time of the K timed steps        [MTEPS, reference formula
                 tests/bfs/test_bfs.cu:187-215 applied to the whole timed region, Reset included]
  enact_mteps  = same edges / summed Enact-only device time (the reference's own timer placement)
  roofline     = algorithmic bytes (4*edges_visited + 20*nodes_visited per BFS) divided by the
                 summed HIP-event durations of the operator kernels, measured live in an instrumented pass
  cpu_baseline = the oracle's serial deque BFS (port of the reference's SimpleReferenceBfs) on the host
                 cores of this box, on a bounded sample of the same graph

N > 1 (launched by torch.distributed.run): the same graph vertex-partitioned over the ranks
(owner = v mod N), per-level halo exchange over RCCL (gunrockinst_amd/multi_gpu.py); "scaling": "strong".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=65)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=int, default=24)
    ap.add_argument("--edge-factor", type=int, default=8, help="generated pairs per vertex (mirrored: x2 directed)")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x6772)
    ap.add_argument("--cpu-baseline-runs", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--primitive", choices=["bfs", "cc", "sssp", "bc", "pr"], default="bfs",
                    help="bfs = the headline metric (default); cc / sssp = BASELINE.json configs 4 / 3 on one GPU")
    ap.add_argument("--delta-factor", type=float, default=16)
    ap.add_argument("--skip-topdown-leg", action="store_true",
                    help="omit the secondary top-down-only figure (keeps rocprof summaries to the headline configuration)")
    ap.add_argument("--alpha", type=float, default=0.0, help="direction switch tuning (0 = library default)")
    ap.add_argument("--beta", type=float, default=0.0)
    ap.add_argument("--lite-factor", type=float, default=-1.0)
    ap.add_argument("--head-pass-min", type=int, default=-1, help="heads-then-rest level: min frontier edges (-1 auto, 0 off)")
    ap.add_argument("--head-pass-max", type=int, default=-1, help="heads-then-rest level: max frontier edges (-1 auto, 0 none)")
    ap.add_argument("--tail-edge-limit", type=int, default=-1, help="levels up to this many edges run in the one-workgroup tail kernel (-1 = library default)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="named enactor knob (grx_bfs_set_option), repeatable")
    ap.add_argument("--traversal-mode", type=int, default=2,
                    help="0 = load-balanced top-down only, 2 = direction-optimizing (default)")
    ap.add_argument("--graph", choices=["rmat", "lj"], default=None,
                    help="rmat = mirrored R-MAT of --scale (default; for --primitive sssp the default is lj); lj = the DIRECTED R-MAT stand-in for soc-LiveJournal1 "
                         "(BASELINE.json config 3: the file is not available offline): 2^22 vertex ids, ~69 M directed edges, not mirrored")
    ap.add_argument("--no-secondary", action="store_true",
                    help="headline leg only: skip the compact legs for BASELINE.json configs 2, 3 (stand-in) and 4 that the default "
                         "N=1 run appends under 'secondary'")
    a = ap.parse_args()
    if a.graph is None:
        a.graph = "lj" if a.primitive == "sssp" else "rmat"
    return a


# soc-LiveJournal1 (SNAP): 4 847 571 vertices, 68 993 773 directed edges.  Stand-in: R-MAT over 2^22 ids (the seeded generator
# works on powers of two), the reference's a/b/c/d, NOT mirrored; the pair count is chosen so that the deduplicated graph has
# about the same number of directed edges.
LJ_SCALE = 22
LJ_PAIRS = 73_400_000


def make_graph(args, devgraph):
    """-> (n, m, row_offsets, col_indices, description, symmetric)"""
    if args.graph == "lj":
        rows, cols = devgraph.rmat_tuples_device(LJ_SCALE, LJ_PAIRS, args.seed)
        ro, ci = devgraph.csr_from_tuples_device(1 << LJ_SCALE, rows, cols, undirected=False)
        n, m = 1 << LJ_SCALE, int(ci.shape[0])
        return n, m, ro, ci, ("DIRECTED R-MAT stand-in for soc-LiveJournal1 (4.85 M vertices / 69.0 M directed edges; file unavailable "
                              "offline): 2^%d ids, %d generated pairs, not mirrored, a=.55 b=.2 c=.2 d=.05, seed 0x%x: n=%d, m=%d directed edges"
                              % (LJ_SCALE, LJ_PAIRS, args.seed, n, m)), False
    ro, ci = devgraph.rmat_csr_device(args.scale, args.edge_factor, args.seed)
    n, m = 1 << args.scale, int(ci.shape[0])
    return n, m, ro, ci, ("R-MAT scale-%d (a=.55 b=.2 c=.2 d=.05, %d pairs/vertex mirrored, seed 0x%x): n=%d, m=%d directed edges"
                          % (args.scale, args.edge_factor, args.seed, n, m)), True


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    # Libraries write banners to stdout (RCCL prints its version block when the first communicator is created): everything
    # but the result line goes to stderr.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    # GUNROCK_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal on a 1-GPU box); the default is RCCL.
    backend = os.environ.get("GUNROCK_DIST_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    # GUNROCK_FORCE_PARTITIONED=1: run the vertex-partitioned level loop even at world 1 (measures its per-level overhead)
    partitioned = world > 1 or os.environ.get("GUNROCK_FORCE_PARTITIONED") == "1"
    if partitioned:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import gunrockinst_amd as ga
    from gunrockinst_amd import devgraph
    ga.lib()  # fail loudly if the HIP library is missing

    if partitioned:
        from gunrockinst_amd import multi_gpu

        def checker(full_labels, sources, n, m_global):
            """rank 0, after the timed region: the whole graph rebuilt on this rank's GPU, the oracle's serial BFS (port of the
            reference's SimpleReferenceBfs) from the first sources -- label-for-label parity of the first one and the CPU baseline"""
            if args.no_cpu_baseline:
                return None, None
            from oracle import gr_oracle as o
            gro, gci = devgraph.rmat_csr_device(args.scale, args.edge_factor, args.seed)
            h_ro, h_ci = devgraph.to_host_csr(gro, gci)
            del gro, gci
            g = o.Csr(n, h_ro, h_ci)
            cpu_edges, cpu_s, parity = 0, 0.0, None
            for k in range(max(args.cpu_baseline_runs, 1)):
                t0 = time.perf_counter()
                ref, _, _ = o.bfs(g, sources[k % len(sources)])
                cpu_s += time.perf_counter() - t0
                cpu_edges += o.bfs_stats(g, ref)[1]
                if k == 0:
                    parity = bool((ref == full_labels).all()) and int(h_ci.shape[0]) == m_global
            cpu = {"value": round(cpu_edges / (cpu_s * 1e6), 2), "unit": "MTEPS", "cores": 1, "kind": "port",
                   "sample": "%d serial deque BFS runs (oracle port of SimpleReferenceBfs) over the whole scale-%d graph on rank 0's host, "
                             "%.1f s CPU" % (max(args.cpu_baseline_runs, 1), args.scale, cpu_s)}
            return parity, cpu

        result = multi_gpu.bench(args, rank, world, local_rank, checker)
    elif args.primitive == "cc":
        result = bench_cc(args, torch, ga, devgraph, local_rank)
    elif args.primitive == "sssp":
        result = bench_sssp(args, torch, ga, devgraph, local_rank)
    elif args.primitive == "pr":
        result = bench_pr(args, torch, ga, devgraph, local_rank)
    elif args.primitive == "bc":
        result = bench_bc(args, torch, ga, devgraph, local_rank)
    else:
        result = bench_single(args, torch, ga, devgraph, local_rank)
    if rank == 0:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)  # the real stdout carries exactly one line
        print(json.dumps(result), flush=True)
    if partitioned:
        dist.barrier()
        dist.destroy_process_group()


_BFS_KERNELS = ("BfsResetKernel", "ArmKernel", "BitmapDiffKernel", "BitmapCopyKernel", "BitmapToQueueKernel", "BottomUpKernel", "BottomUpSparseKernel",
                "BottomUpAutoKernel", "BottomUpHeadsKernel", "ChainedPersistentLevelsKernel", "EmitLabelsKernel",
                "FreshToBitmapKernel", "LoadBalancedKernel", "BinnedExpandKernel", "BinnedApplyKernel", "PersistentLevelsKernel",
                "TailLevelsKernel", "QueueToBitmapKernel", "PublishKernel")
PROFILE_TAG = "r03"


def source_fingerprint():
    """sha1 over the kernel sources: profiles taken from other sources than the ones benchmarked are not quoted (the GPU box
    has no .git, so the commit id itself is not available there)."""
    import hashlib
    h = hashlib.sha1()
    base = os.path.join(ROOT, "gunrockinst_amd", "csrc")
    for d, _, files in sorted(os.walk(base)):
        if os.sep + "build" in d:
            continue
        for f in sorted(files):
            if f.endswith((".hpp", ".hip", ".h")):
                with open(os.path.join(d, f), "rb") as fh:
                    h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _short_kernel(name):
    import re
    name = name.split("(")[0].replace("void ", "")
    return re.sub(r"<.*", "", name).split("::")[-1].strip()


def profile_tables():
    """Per-kernel figures from the committed rocprofv3 passes of THIS command (profiles/README.md): kernel-trace stats
    (calls, average duration) and FETCH_SIZE / WRITE_SIZE from separate --pmc passes.  Returns None when the profiles are
    missing or were taken from different kernel sources.
    FETCH_SIZE is quoted RAW: on gfx950 it counts a wide coalesced streaming read at half its bytes (MI355X_MICROARCH.md,
    HBM) and is uncalibrated for the 4-byte gathers that dominate a traversal, so the true read traffic lies between
    fetch_raw and 2 * fetch_raw."""
    import csv
    pdir = os.path.join(ROOT, "profiles")
    try:
        with open(os.path.join(pdir, PROFILE_TAG + "_profile_meta.json")) as fh:
            meta = json.load(fh)
        if meta.get("source_sha") != source_fingerprint():
            return {"stale": True, "profiled_source_sha": meta.get("source_sha"), "source_sha": source_fingerprint()}
        with open(os.path.join(pdir, PROFILE_TAG + "_bench_pmc_fetch_size.json")) as fh:
            fetch = json.load(fh)
        with open(os.path.join(pdir, PROFILE_TAG + "_bench_pmc_write_size.json")) as fh:
            write = json.load(fh)
        stats = {}
        with open(os.path.join(pdir, PROFILE_TAG + "_bench_kernel_stats.csv")) as fh:
            for r in csv.DictReader(fh):
                k = _short_kernel(r["Name"])
                a = stats.setdefault(k, [0, 0.0])
                a[0] += int(r["Calls"])
                a[1] += float(r["TotalDurationNs"])
    except (OSError, KeyError, ValueError):
        return None
    searches_f = fetch.get("BfsResetKernel", {}).get("dispatches", 0)
    searches_w = write.get("BfsResetKernel", {}).get("dispatches", 0)
    if not searches_f or not searches_w:
        return None
    table = {}
    fb = wb = 0.0
    for k in _BFS_KERNELS:
        f, w, st = fetch.get(k), write.get(k), stats.get(k)
        if not f or not w:
            continue
        fpl = f["sum"] * 1024.0 / f["dispatches"]          # bytes per launch (the tool reports KiB)
        wpl = w["sum"] * 1024.0 / w["dispatches"]
        fb += f["sum"] * 1024.0 / searches_f
        wb += w["sum"] * 1024.0 / searches_w
        row = {"launches_per_search": round(f["dispatches"] / searches_f, 2), "fetch_raw_bytes_per_launch": round(fpl),
               "write_bytes_per_launch": round(wpl)}
        if st and st[0]:
            avg_ns = st[1] / st[0]
            row["avg_launch_us"] = round(avg_ns / 1e3, 2)
            row["hbm_gbps_raw"] = round((fpl + wpl) / avg_ns, 1)
            row["hbm_gbps_upper"] = round((2 * fpl + wpl) / avg_ns, 1)
        table[k] = row
    return {"stale": False, "source_sha": meta.get("source_sha"), "per_search": {"fetch_raw": round(fb), "write": round(wb),
            "bytes": round(fb + wb), "upper": round(2 * fb + wb)}, "per_kernel": table,
            "source": "profiles/%s_bench_{kernel_stats.csv,pmc_fetch_size.json,pmc_write_size.json}" % PROFILE_TAG}


def bench_single(args, torch, ga, devgraph, device_index):
    t0 = time.time()
    n, m, ro, ci, graph_desc, symmetric = make_graph(args, devgraph)
    torch.cuda.synchronize()
    build_s = time.time() - t0
    src0, maxdeg = devgraph.largest_degree_source(ro)
    sources = [src0] + devgraph.seeded_sources(ro, 64, args.seed)
    deg = (ro[1:] - ro[:-1]).long()

    mode = args.traversal_mode
    prob = ga.BfsProblem(mark_pred=False, idempotence=True, instrument=False, device=device_index)
    prob.init_device(n, m, ro.data_ptr(), ci.data_ptr())
    inverse_build_ms = None
    if symmetric:
        prob.set_inverse_graph()      # the R-MAT graph is mirrored: its CSR is its own inverse
    else:                             # what gunrock_bfs_func does for a directed input: transpose built on the device (outside Enact)
        enabled, built, inverse_build_ms = prob.auto_inverse()
        assert enabled and built
    prob.set_tuning(args.alpha, args.beta, args.lite_factor, args.tail_edge_limit)
    prob.set_head_pass(args.head_pass_min, args.head_pass_max)
    for kv in args.opt:
        prob.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    d_labels, _ = prob.device_results()
    labels_t = devgraph.as_tensor(d_labels, n)

    def step(p, k, md=mode):
        s = sources[k % len(sources)]
        p.reset(s)
        return p.enact(s, traversal_mode=md)

    for k in range(args.warmup):
        step(prob, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enact_ms, per_step_ms = 0.0, []
    for k in range(args.steps):
        per_step_ms.append(step(prob, k))
        enact_ms += per_step_ms[-1]
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0

    # per-source visited counts (untimed): rerun each distinct source used, read labels on the device
    used = [sources[k % len(sources)] for k in range(args.steps)]
    per_src = {}
    for s in sorted(set(used)):
        prob.reset(s)
        prob.enact(s, traversal_mode=mode)
        vis = labels_t > -1
        per_src[s] = (int(vis.sum()), int(deg[vis].sum()), prob.stats()["search_depth"])
    edges_total = sum(per_src[s][1] for s in used)
    nodes_total = sum(per_src[s][0] for s in used)
    value = edges_total / (wall * 1e6)
    enact_mteps = edges_total / (enact_ms * 1e3)
    # harmonic mean over the sources of the per-search rate (the reference's own timer placement: Enact only)
    rates = [per_src[s][1] / (t * 1e3) for s, t in zip(used, per_step_ms) if per_src[s][1] > 0 and t > 0]
    harmonic = len(rates) / sum(1.0 / r for r in rates) if rates else None

    # second leg: the same sources with the load-balanced top-down advance only (reference traversal_mode 0 -- the mode SURVEY
    # 8(d) names for config 2 and the engine under SSSP / BC / directed graphs / the partitioned loop's top-down levels)
    td_ms, td_edges, td_nodes, td_steps = 0.0, 0, 0, (0 if args.skip_topdown_leg or mode == 0 else min(args.steps, 8))
    for k in range(td_steps):
        td_ms += step(prob, k, 0)
        s_k = sources[k % len(sources)]
        td_edges += per_src[s_k][1] if s_k in per_src else 0
        td_nodes += per_src[s_k][0] if s_k in per_src else 0
    topdown_mteps = td_edges / (td_ms * 1e3) if td_ms > 0 and td_edges else None

    # instrumented pass (separate enactor instantiation): HIP events around every operator launch, on the launch stream
    iprob = ga.BfsProblem(False, True, instrument=True, device=device_index)
    iprob.init_device(n, m, ro.data_ptr(), ci.data_ptr())
    if symmetric:
        iprob.set_inverse_graph()
    else:
        iprob.auto_inverse()
    iprob.set_tuning(args.alpha, args.beta, args.lite_factor, args.tail_edge_limit)
    iprob.set_head_pass(args.head_pass_min, args.head_pass_max)
    for kv in args.opt:
        iprob.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    names = {6: "BottomUpKernel heads-only + count-only advance + FreshToBitmapKernel", 0: "advance::LoadBalancedKernel (top-down)", 1: "advance::BottomUpAutoKernel (chain of bottom-up sweeps, dense / compacting chosen on the device)",
             2: "BitmapToQueueKernel + PersistentLevelsKernel (+ EmitLabelsKernel queued behind it)", 3: "advance::TailLevelsKernel",
             4: "LoadBalancedKernel count-only + FreshToBitmapKernel", 5: "advance::PersistentLevelsKernel",
             7: "BinnedExpandKernel + BinnedApplyKernel + FreshToBitmapKernel + BitmapToQueueKernel (binned top-down)",
             8: "advance::TwcLevelsKernel", 9: "EmitLabelsKernel (when not already queued behind the closing launch)"}
    by_kind = {}
    kernel_ms, launches, balg = 0.0, 0, 0.0
    for k in range(min(args.steps, len(sources))):
        s = sources[k % len(sources)]
        iprob.reset(s)
        iprob.enact(s, traversal_mode=mode)
        for rec in iprob.level_trace():
            agg = by_kind.setdefault(rec["kind"], [0, 0.0])
            agg[0] += 1
            agg[1] += rec["ms"]
            kernel_ms += rec["ms"]
            launches += 1
        if s in per_src:
            nv, ev, _ = per_src[s]
        else:
            il, _ = iprob.device_results()
            vis = devgraph.as_tensor(il, n) > -1
            nv, ev = int(vis.sum()), int(deg[vis].sum())
        balg += 4.0 * ev + 20.0 * nv
    dom = max(by_kind, key=lambda kd: by_kind[kd][1]) if by_kind else 0
    n_inst = max(min(args.steps, len(sources)), 1)
    # THE roofline figure (BASELINE.md section 4 / SURVEY 8(d)): algorithmic bytes of the searches / their Enact time
    balg_timed = 4.0 * edges_total + 20.0 * nodes_total
    achieved = balg_timed / (enact_ms * 1e-3) / 1e9 if enact_ms > 0 else 0.0
    kernel_only = balg / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0   # same bytes / summed operator-kernel event time
    whole_step = balg_timed / wall / 1e9
    prof = profile_tables() if (mode == 2 and args.graph == "rmat" and args.scale == 24 and args.edge_factor == 8 and args.seed == 0x6772) else None
    fresh = bool(prof) and not prof.get("stale")
    kernel_s_per_search = (kernel_ms / n_inst) * 1e-3
    phys = None
    if fresh and kernel_s_per_search > 0:
        phys = {"raw": round(prof["per_search"]["bytes"] / kernel_s_per_search / 8e12, 4),
                "upper": round(prof["per_search"]["upper"] / kernel_s_per_search / 8e12, 4)}
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 5),
                "definition": "algorithmic bytes (4*edges_visited + 20*nodes_visited) of the timed searches / their summed Enact "
                              "time / 8 TB/s; a direction-optimizing search does not read most of those edges, so this is NOT "
                              "the physical HBM rate (see hbm_frac_physical)",
                "frac_kernel_only": round(kernel_only / 8000.0, 5),
                "frac_whole_step": round(whole_step / 8000.0, 5),
                "traffic": prof["per_search"]["bytes"] if fresh else None,
                "traffic_detail": prof["per_search"] if fresh else (prof if prof else None),
                "hbm_frac_physical": phys,
                "kernel": names.get(dom, str(dom)),
                "kernel_share_of_device_time": round(by_kind[dom][1] / kernel_ms, 4) if kernel_ms else None,
                "kernel_launches": by_kind[dom][0] if by_kind else 0,
                "kernel_avg_launch_ms": round(by_kind[dom][1] / max(by_kind[dom][0], 1), 5) if by_kind else None,
                "all_launches": launches, "all_kernel_ms_per_bfs": round(kernel_ms / n_inst, 5),
                "alg_bytes_per_bfs": round(balg / n_inst, 1),
                "by_kernel_ms": {names.get(kd, str(kd)): round(v[1], 4) for kd, v in sorted(by_kind.items())},
                "per_kernel": prof["per_kernel"] if fresh else None,
                "profile_source": prof.get("source") if fresh else None}
    # the top-down leg's own roofline object: same B_alg definition over its searches / their Enact time; dominant kernel and its
    # average launch from an instrumented traversal_mode-0 pass over the same sources
    topdown = None
    if td_steps and td_ms > 0:
        td_kind, td_kernel_ms = {}, 0.0
        for k in range(td_steps):
            s_k = sources[k % len(sources)]
            iprob.reset(s_k)
            iprob.enact(s_k, traversal_mode=0)
            for rec in iprob.level_trace():
                agg = td_kind.setdefault(rec["kind"], [0, 0.0])
                agg[0] += 1
                agg[1] += rec["ms"]
                td_kernel_ms += rec["ms"]
        td_dom = max(td_kind, key=lambda kd: td_kind[kd][1]) if td_kind else 0
        td_balg = 4.0 * td_edges + 20.0 * td_nodes
        td_achieved = td_balg / (td_ms * 1e-3) / 1e9
        topdown = {"workload": "the first %d sources, traversal_mode 0 (load-balanced top-down advance only; reference default mode)" % td_steps,
                   "enact_ms_per_step": round(td_ms / td_steps, 4), "enact_mteps": round(topdown_mteps, 2),
                   "bound": "hbm", "achieved": round(td_achieved, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(td_achieved / 8000.0, 5),
                   "kernel": names.get(td_dom, str(td_dom)),
                   "kernel_share_of_device_time": round(td_kind[td_dom][1] / td_kernel_ms, 4) if td_kernel_ms else None,
                   "kernel_launches": td_kind[td_dom][0] if td_kind else 0,
                   "kernel_avg_launch_ms": round(td_kind[td_dom][1] / max(td_kind[td_dom][0], 1), 5) if td_kind else None,
                   "all_kernel_ms_per_bfs": round(td_kernel_ms / td_steps, 5),
                   "by_kernel_ms": {names.get(kd, str(kd)): round(v[1], 4) for kd, v in sorted(td_kind.items())}}
    iprob.close()

    cpu = None
    cpu_parallel = None
    parity = None
    if not args.no_cpu_baseline:
        from oracle import gr_oracle as o
        h_ro, h_ci = devgraph.to_host_csr(ro, ci)
        g = o.Csr(n, h_ro, h_ci)
        cpu_edges, cpu_s = 0, 0.0
        for k in range(args.cpu_baseline_runs):
            s = sources[k % len(sources)]
            t0 = time.perf_counter()
            ref_labels, _, _ = o.bfs(g, s)
            cpu_s += time.perf_counter() - t0
            cpu_edges += o.bfs_stats(g, ref_labels)[1]
            if k == 0:
                prob.reset(s)
                prob.enact(s, traversal_mode=mode)
                got, _ = prob.extract()
                parity = bool((got == ref_labels).all())
        cpu = {"value": round(cpu_edges / (cpu_s * 1e6), 2), "unit": "MTEPS", "cores": 1, "kind": "port",
               "sample": "%d serial deque BFS runs (oracle port of SimpleReferenceBfs) on the same scale-%d graph, "
                         "%.1f s CPU" % (args.cpu_baseline_runs, args.scale, cpu_s)}
        # the stronger CPU baseline of SURVEY 8(d): level-synchronous OpenMP BFS on all host cores (not in the reference)
        # threads: the box's CPU share for one GPU is 16 cores even where the OS shows more
        want = min(len(os.sched_getaffinity(0)), int(os.environ.get("GUNROCK_CPU_THREADS", "16")))
        o.bfs_parallel(g, sources[0], want)  # thread start-up
        par_edges, par_s, threads = 0, 0.0, 1
        for k in range(max(args.cpu_baseline_runs, 4)):
            s = sources[k % len(sources)]
            t0 = time.perf_counter()
            par_labels, threads = o.bfs_parallel(g, s, want)
            par_s += time.perf_counter() - t0
            par_edges += o.bfs_stats(g, par_labels)[1]
        cpu_parallel = {"value": round(par_edges / (par_s * 1e6), 2), "unit": "MTEPS", "cores": threads, "kind": "port",
                        "sample": "%d OpenMP level-synchronous BFS runs (oracle/gr_oracle.c gro_bfs_parallel) on the same "
                                  "graph, %.2f s wall" % (max(args.cpu_baseline_runs, 4), par_s)}
    prob.close()

    depth = per_src[used[0]][2]
    result = {
        "metric": "MTEPS (million traversed edges/sec) BFS R-MAT scale-%d" % args.scale if args.graph == "rmat" else
                  "MTEPS (million traversed edges/sec) BFS, directed soc-LiveJournal1 stand-in",
        "value": round(value, 2), "unit": "MTEPS", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "BFS idempotent %s, %s; sources: largest-degree + 64 seeded"
                               % ("direction-optimizing (traversal_mode 2)" if mode == 2 else "top-down (traversal_mode 0)", graph_desc),
                   "search_depth_src0": depth, "graph_build_s": round(build_s, 2), "max_degree": maxdeg,
                   "inverse_graph_build_ms": None if inverse_build_ms is None else round(inverse_build_ms, 3)},
        "enact_mteps": round(enact_mteps, 2), "harmonic_mean_enact_mteps": None if harmonic is None else round(harmonic, 2),
        "topdown_only_enact_mteps": None if topdown_mteps is None else round(topdown_mteps, 2), "enact_ms_per_step": round(enact_ms / args.steps, 4),
        "edges_visited_per_step": edges_total // args.steps, "nodes_visited_per_step": nodes_total // args.steps,
        "parity_vs_oracle": parity,
        "roofline": roofline, "topdown": topdown, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_parallel,
    }
    # BASELINE.json's other single-GPU configs, compact, in the same line: config 2 (BFS R-MAT scale-22), config 4 (CC R-MAT scale-24),
    # config 3 through its directed stand-in (SSSP, and BFS on the device-built inverse graph)
    if args.scale == 24 and args.graph == "rmat" and mode == 2 and not args.no_secondary:
        del ro, ci, deg, labels_t
        torch.cuda.empty_cache()
        result["secondary"] = secondary_legs(args, torch, ga, devgraph, device_index)
    return result


def _compact(d, extra=()):
    keep = ("metric", "value", "unit", "steps", "ms_per_step", "enact_ms_per_step", "parity_vs_oracle", "cpu_baseline") + tuple(extra)
    out = {k: d.get(k) for k in keep if k in d}
    out["workload"] = d["config"]["workload"]
    r = d.get("roofline") or {}
    out["roofline"] = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "frac_whole_step", "kernel", "kernel_avg_launch_ms",
                                             "alg_bytes", "note", "I_h", "I_j", "parked_edge_fraction", "frac_own_sweeps", "this_run") if k in r}
    return out


def secondary_legs(args, torch, ga, devgraph, device_index):
    import copy
    legs = {}

    def run(name, fn, **over):
        a = copy.copy(args)
        a.no_secondary, a.skip_topdown_leg = True, True
        for k, v in over.items():
            setattr(a, k, v)
        try:
            legs[name] = fn(a)
        except Exception as e:  # a failed leg must not take the headline line with it; it is reported as failed
            legs[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()

    run("config2_bfs_rmat22", lambda a: _compact(bench_single(a, torch, ga, devgraph, device_index), ("enact_mteps", "harmonic_mean_enact_mteps")),
        scale=22, steps=65, warmup=3)
    run("config4_cc_rmat24", lambda a: _compact(bench_cc(a, torch, ga, devgraph, device_index)), scale=24, steps=10, warmup=2)
    run("config3_sssp_directed_standin", lambda a: _compact(bench_sssp(a, torch, ga, devgraph, device_index)), graph="lj", steps=9, warmup=2)
    run("config3_graph_bfs_directed", lambda a: _compact(bench_single(a, torch, ga, devgraph, device_index), ("enact_mteps",)),
        graph="lj", steps=33, warmup=3, cpu_baseline_runs=1)
    return legs


def bench_bc(args, torch, ga, devgraph, device_index):
    """Betweenness centrality: one Brandes pass per step (forward BFS with path counts + backward
    dependency accumulation) from the largest-degree source and seeded sources, R-MAT as for the other primitives."""
    import numpy as np
    n = 1 << args.scale
    ro, ci = devgraph.rmat_csr_device(args.scale, args.edge_factor, args.seed)
    m = int(ci.shape[0])
    deg = (ro[1:] - ro[:-1]).long()
    src0, _ = devgraph.largest_degree_source(ro)
    sources = [src0] + devgraph.seeded_sources(ro, 8, args.seed)
    p = ga.BcProblem(device_index).init_device(n, m, ro.data_ptr(), ci.data_ptr())
    steps = max(1, min(args.steps, len(sources)))
    for k in range(min(args.warmup, 2)):
        p.run(sources[k % len(sources)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_ms = 0.0
    for k in range(steps):
        run_ms += p.run(sources[k % len(sources)])
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    # edges of the reached component, from a BFS of the same source (every reached vertex's edges are walked twice)
    bp = ga.BfsProblem(False, True, False, device_index).init_device(n, m, ro.data_ptr(), ci.data_ptr())
    edges_total, nodes_total = 0, 0
    for k in range(steps):
        s = sources[k % len(sources)]
        bp.reset(s); bp.enact(s, traversal_mode=0)
        vis = devgraph.as_tensor(bp.device_results()[0], n) > -1
        nodes_total += int(vis.sum()); edges_total += int(deg[vis].sum())
    bp.close()
    cpu, parity = None, None
    if not args.no_cpu_baseline:
        from oracle import gr_oracle as o
        h_ro, h_ci = devgraph.to_host_csr(ro, ci)
        g = o.Csr(n, h_ro, h_ci)
        t0 = time.perf_counter()
        ref, _ = o.bc(g, src0)
        cpu_s = time.perf_counter() - t0
        p.run(src0)
        _, got = p.extract()
        parity = bool(np.all(np.abs(got.astype(np.float64) - ref) <= 1e-3 * np.abs(ref) + 1e-3))
        cpu = {"value": round(edges_total / steps * 2 / (cpu_s * 1e6), 2), "unit": "MTEPS", "cores": 1, "kind": "port",
               "sample": "1 Brandes pass (oracle, doubles) from the max-degree source, %.1f s" % cpu_s}
    p.close()
    balg = 8.0 * edges_total + 40.0 * nodes_total   # forward + backward: 4 B per edge each way, ~40 B of per-vertex state
    achieved = balg / (run_ms * 1e-3) / 1e9
    return {"metric": "BC R-MAT scale-%d: MTEPS = 2 x edges of reached vertices / time of one source (forward + backward)" % args.scale,
            "value": round(2.0 * edges_total / (run_ms * 1e3), 2), "unit": "MTEPS", "n_gpus": 1, "steps": steps,
            "warmup": min(args.warmup, 2), "ms_per_step": round(wall * 1e3 / steps, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BC (Brandes, one source per step), R-MAT scale-%d: n=%d, m=%d" % (args.scale, n, m)},
            "run_ms_per_step": round(run_ms / steps, 4), "parity_vs_oracle": parity,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 5), "traffic": None, "kernel": "advance::LoadBalancedKernel<Forward/BackwardFunctor>",
                         "alg_bytes": balg, "note": "whole Reset + Enact device time of a source (grx_bc_run)"},
            "cpu_baseline": cpu}


def bench_pr(args, torch, ga, devgraph, device_index):
    """PageRank (the reducing advance's first user): a step = Reset + Enact of a fixed number of iterations
    (threshold 0, so none stops early), R-MAT as for the other primitives; the graph is symmetric, so its CSR is its own
    in-neighbour table.  Parity: against the oracle's restatement of the reference schedule, which is UNPINNED."""
    import numpy as np
    n = 1 << args.scale
    ro, ci = devgraph.rmat_csr_device(args.scale, args.edge_factor, args.seed)
    m = int(ci.shape[0])
    iters = 20
    p = ga.PrProblem(device=device_index).init_device(n, m, ro.data_ptr(), ci.data_ptr())
    p.set_inverse_graph()
    steps = max(1, min(args.steps, 10))
    for _ in range(min(args.warmup, 2)):
        p.reset(-1, 0.85, 0.0); p.enact(iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enact_ms = 0.0
    for _ in range(steps):
        p.reset(-1, 0.85, 0.0)
        enact_ms += p.enact(iters)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    st = p.stats()
    done = max(st["iterations"], 1)
    cpu, parity = None, None
    if not args.no_cpu_baseline:
        from oracle import gr_oracle as o
        h_ro, h_ci = devgraph.to_host_csr(ro, ci)
        g = o.Csr(n, h_ro, h_ci)
        t0 = time.perf_counter()
        ref, _, ref_iters = o.pagerank(g, -1, 0.85, 0.0, iters)
        cpu_s = time.perf_counter() - t0
        ids, ranks = p.extract()
        by_vertex = np.zeros(n, dtype=np.float64)
        by_vertex[ids] = ranks
        parity = bool(np.allclose(by_vertex, ref, rtol=1e-4, atol=1e-6)) and ref_iters == done
        cpu = {"value": round(m * ref_iters / (cpu_s * 1e6), 2), "unit": "MTEPS", "cores": 1, "kind": "port",
               "sample": "%d iterations of the oracle's PageRank restatement (doubles, parity unpinned) on the same graph, %.1f s" % (ref_iters, cpu_s)}
    p.close()
    # per iteration: 4 B per in-edge (column index) + 4 B gathered contribution per in-edge + ~16 B per vertex (rank in/out, contribution, degree)
    balg = steps * done * (8.0 * m + 16.0 * n)
    achieved = balg / (enact_ms * 1e-3) / 1e9
    return {"metric": "PageRank R-MAT scale-%d: MTEPS = edges x iterations / Enact time" % args.scale,
            "value": round(m * done * steps / (enact_ms * 1e3), 2), "unit": "MTEPS", "n_gpus": 1, "steps": steps,
            "warmup": min(args.warmup, 2), "ms_per_step": round(wall * 1e3 / steps, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "PageRank delta 0.85, %d iterations (threshold 0), R-MAT scale-%d: n=%d, m=%d" % (done, args.scale, n, m)},
            "enact_ms_per_step": round(enact_ms / steps, 4), "ms_per_iteration": round(enact_ms / steps / done, 4),
            "parity_vs_oracle": parity, "parity_note": "oracle = restatement of the reference schedule, PARITY UNPINNED",
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 5),
                         "traffic": None, "kernel": "advance::ReduceKernel<PRFunctor, PLUS>",
                         "alg_bytes_per_iteration": 8.0 * m + 16.0 * n,
                         "note": "the contribution gather costs a 64-byte sector per edge once the table outgrows L2 (DESIGN.md 3.6): "
                                 "physical traffic is ~8x the algorithmic 4 B"},
            "cpu_baseline": cpu}


def bench_cc(args, torch, ga, devgraph, device_index):
    """BASELINE.json config 4: connected components on R-MAT (hook / pointer-jump filter loop), one GPU.
    Roofline numerator per SURVEY 8(d): B_alg = I_h * 9m + I_j * 8n with I_h / I_j the hook / jump sweeps of the REFERENCE schedule
    on this input, taken from the oracle's sequential simulation of cc_enactor.cuh:165-873 (cc_reference_schedule) -- not this
    run's own sweep counts, and not discounted for the edges this implementation parks."""
    n, m, ro, ci, graph_desc, _ = make_graph(args, devgraph)
    p = ga.CcProblem(instrument=False, device=device_index).init_device(n, m, ro.data_ptr(), ci.data_ptr())
    steps = max(1, min(args.steps, 10))
    for _ in range(min(args.warmup, 2)):
        p.reset(); p.enact()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enact_ms = 0.0
    for _ in range(steps):
        p.reset()
        enact_ms += p.enact()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    st = p.stats()
    ids = devgraph.as_tensor(p.device_results(), n)
    components = int((ids == torch.arange(n, device=ids.device, dtype=torch.int32)).sum())
    ip = ga.CcProblem(instrument=True, device=device_index).init_device(n, m, ro.data_ptr(), ci.data_ptr())
    ip.reset(); ip.enact()
    ist = ip.stats()
    ip.close()
    cpu, parity, ref_ih, ref_ij = None, None, None, None
    if not args.no_cpu_baseline:
        from oracle import gr_oracle as o
        h_ro, h_ci = devgraph.to_host_csr(ro, ci)
        g = o.Csr(n, h_ro, h_ci)
        t0 = time.perf_counter()
        ref, ref_count = o.cc(g)
        cpu_s = time.perf_counter() - t0
        got, _ = p.extract()
        parity = bool((got == ref).all()) and ref_count == components
        cpu = {"value": round(m / (cpu_s * 1e6), 2), "unit": "M edges/s", "cores": 1, "kind": "port",
               "sample": "1 union-find pass (oracle restatement of Boost connected_components) over the same graph, %.1f s" % cpu_s}
        t0 = time.perf_counter()
        sim, sim_count, ref_ih, ref_ij = o.cc_reference_schedule(g)
        sim_s = time.perf_counter() - t0
        parity = parity and sim_count == ref_count
        cpu["reference_schedule_simulation"] = "I_h=%d hook sweeps, I_j=%d jump sweeps (sequential simulation of cc_enactor.cuh:165-873, %.1f s)" % (ref_ih, ref_ij, sim_s)
    p.close()
    own_balg = st["edge_sweeps"] * 9.0 * st.get("sweep_edges", m) + st["vertex_sweeps"] * 8.0 * n
    balg = (ref_ih * 9.0 * m + ref_ij * 8.0 * n) if ref_ih is not None else own_balg
    t_enact = enact_ms / steps * 1e-3
    achieved = balg / t_enact / 1e9
    return {"metric": "CC R-MAT scale-%d: million edges per second of Enact (m / t)" % args.scale,
            "value": round(m / (enact_ms / steps * 1e3), 2), "unit": "M edges/s", "n_gpus": 1, "steps": steps,
            "warmup": min(args.warmup, 2), "ms_per_step": round(wall * 1e3 / steps, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "CC hook/pointer-jump, %s; %d components; this run: %d edge sweeps, %d vertex sweeps"
                                   % (graph_desc, components, st["edge_sweeps"], st["vertex_sweeps"])},
            "enact_ms_per_step": round(enact_ms / steps, 4), "parity_vs_oracle": parity,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 5), "traffic": None, "kernel": "filter::ApplyKernel (hook / jump sweeps)",
                         "launches": ist["kernel_launches"], "kernel_ms": round(ist["kernel_ms"], 4),
                         "frac_kernel_only": round(balg / (ist["kernel_ms"] * 1e-3) / 8e12, 5) if ist["kernel_ms"] > 0 else None,
                         "alg_bytes": balg, "I_h": ref_ih, "I_j": ref_ij,
                         "numerator": "reference schedule (oracle simulation)" if ref_ih is not None else "this run's sweeps (no oracle)",
                         "this_run": {"edge_sweeps": st["edge_sweeps"], "vertex_sweeps": st["vertex_sweeps"], "sweep_edges": st.get("sweep_edges"),
                                      "alg_bytes_own_sweeps": own_balg},
                         "mirrored": st.get("mirrored"),
                         "parked_edge_fraction": round(1.0 - st.get("sweep_edges", m) / float(m), 4) if m else 0.0,
                         "frac_own_sweeps": round(own_balg / t_enact / 8e12, 5),
                         "note": "frac = B_alg / Enact time / 8 TB/s with B_alg = I_h*9m + I_j*8n of the REFERENCE schedule (I_h hooking and I_j jumping "
                                 "sweeps over both orientations of every edge, from the oracle's simulation of cc_enactor.cuh:165-873), as SURVEY 8(d) "
                                 "defines it.  It can exceed 1: this implementation does LESS than that schedule -- on a mirrored graph it materialises only "
                                 "the from > to orientation (the other one performs the identical hooks: sweep_edges = m * (1 - parked_edge_fraction)) and "
                                 "its opening move hooks every vertex under its SMALLEST lower neighbour, and after one neighbour round the sampled giant "
                                 "component sits out of the hooking sweeps, which run one lane per vertex over CSR rows (this_run counts them, the opening "
                                 "and the neighbour round with the vertex sweeps; DESIGN 3.4).  frac_own_sweeps prices this run's own sweeps (9 B per edge "
                                 "of an edge-form sweep, 8 B per vertex of a vertex sweep): that is the bandwidth figure"},
            "cpu_baseline": cpu}


def bench_sssp(args, torch, ga, devgraph, device_index):
    """BASELINE.json config 3 stand-in: soc-LiveJournal1 is not available offline, so R-MAT with seeded integer weights
    in [1, 64]; delta-stepping advance with near/far pile."""
    n, m, ro, ci, graph_desc, _ = make_graph(args, devgraph)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(args.seed)
    w = torch.randint(1, 65, (m,), generator=gen, device="cuda", dtype=torch.int32)
    deg = (ro[1:] - ro[:-1]).long()
    avg_w, avg_d = float(w.double().mean()), float(int(deg.double().mean()))
    delta = (int(avg_w) * 32.0 / max(avg_d, 1.0)) * args.delta_factor     # SSSPProblem::EstimatedDelta x delta_factor
    src0, _ = devgraph.largest_degree_source(ro)
    sources = [src0] + devgraph.seeded_sources(ro, 8, args.seed)
    p = ga.SsspProblem(mark_pred=False, instrument=False, device=device_index)
    p.init_device(n, m, ro.data_ptr(), ci.data_ptr(), w.data_ptr(), delta)
    if os.environ.get("GUNROCK_SSSP_PULL") == "1":   # (off: on this workload no level's frontier holds 3/4 of the edges)
        p.set_inverse_graph()         # weighted in-neighbour lists (transpose on the device): dense levels relax by pulling
    steps = max(1, min(args.steps, len(sources)))
    for k in range(min(args.warmup, 2)):
        p.reset(sources[k % len(sources)]); p.enact(sources[k % len(sources)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enact_ms, relaxed = 0.0, 0
    for k in range(steps):
        s = sources[k % len(sources)]
        p.reset(s)
        enact_ms += p.enact(s)
        relaxed += p.stats()["relaxed_edges"]
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ip = ga.SsspProblem(False, True, device_index)
    ip.init_device(n, m, ro.data_ptr(), ci.data_ptr(), w.data_ptr(), delta)
    ip.reset(src0); ip.enact(src0)
    ist = ip.stats()
    ip.close()
    # reached vertices / their edges per timed source (on a directed graph they differ from source to source): from a BFS of the
    # same source -- the reachable set is the same
    bp = ga.BfsProblem(False, True, False, device_index).init_device(n, m, ro.data_ptr(), ci.data_ptr())
    reach = {}
    for s in sorted(set(sources[k % len(sources)] for k in range(steps))):
        bp.reset(s); bp.enact(s, traversal_mode=0)
        vis = devgraph.as_tensor(bp.device_results()[0], n) > -1
        reach[s] = (int(vis.sum()), int(deg[vis].sum()))
    bp.close()
    n_r = sum(reach[sources[k % len(sources)]][0] for k in range(steps))
    m_r = sum(reach[sources[k % len(sources)]][1] for k in range(steps))
    balg = 8.0 * m_r + 20.0 * n_r                                             # each needed edge once (4 B col + 4 B weight)
    cpu, parity = None, None
    if not args.no_cpu_baseline:
        from oracle import gr_oracle as o
        h_ro, h_ci = devgraph.to_host_csr(ro, ci)
        h_w = w.cpu().numpy().astype("uint32")
        g = o.Csr(n, h_ro, h_ci)
        t0 = time.perf_counter()
        ref, _ = o.sssp(g, src0, h_w)
        cpu_s = time.perf_counter() - t0
        p.reset(src0); p.enact(src0)
        got, _ = p.extract()
        parity = bool((got == ref).all())
        ok = ref != 0xFFFFFFFF
        parity = parity and int(ok.sum()) == reach[src0][0]
        cpu = {"value": round(reach[src0][1] / (cpu_s * 1e6), 2), "unit": "MTEPS", "cores": 1, "kind": "port",
               "sample": "1 heap-Dijkstra run (oracle restatement of Boost dijkstra_shortest_paths) from the max-degree source, %.1f s" % cpu_s}
    p.close()
    achieved = balg / (enact_ms * 1e-3) / 1e9
    balg0 = 8.0 * reach[src0][1] + 20.0 * reach[src0][0]
    roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 5),
            "traffic": None, "kernel": "advance::LoadBalancedKernel<SSSPFunctor> + priority_queue::BisectKernel",
            "frac_kernel_only": round(balg0 / (ist["kernel_ms"] * 1e-3) / 8e12, 5) if ist["kernel_ms"] > 0 else None,
            "launches_src0": ist["kernel_launches"], "kernel_ms_src0": round(ist["kernel_ms"], 4), "alg_bytes": balg,
            "relaxed_edges_src0": ist["relaxed_edges"], "iterations_src0": ist["iterations"],
            "note": "B_alg = 8*m_r + 20*n_r of the timed sources (reached vertices and their edges) / their summed Enact time"}
    return {"metric": ("SSSP R-MAT scale-%d" % args.scale if args.graph == "rmat" else "SSSP directed soc-LiveJournal1 stand-in") +
                      ", uniform weights [1,64]: MTEPS = edges of reached vertices / Enact time",
            "value": round(m_r / (enact_ms * 1e3), 2), "unit": "MTEPS", "n_gpus": 1,
            "steps": steps, "warmup": min(args.warmup, 2), "ms_per_step": round(wall * 1e3 / steps, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "uint32", "data": "synthetic",
            "config": {"workload": "SSSP delta-stepping (near/far), %s; weights uniform int [1,64] seed 0x%x, delta_factor %g (delta %.1f); "
                                   "sources: largest-degree + 8 seeded" % (graph_desc, args.seed, args.delta_factor, delta)},
            "enact_ms_per_step": round(enact_ms / steps, 4), "relaxed_edges_per_step": relaxed // steps,
            "parity_vs_oracle": parity, "roofline": roof, "cpu_baseline": cpu}


if __name__ == "__main__":
    main()
