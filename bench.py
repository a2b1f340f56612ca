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
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import gunrockinst_amd as ga
    from gunrockinst_amd import devgraph
    ga.lib()  # fail loudly if the HIP library is missing

    if world > 1:
        from gunrockinst_amd import multi_gpu
        result = multi_gpu.bench(args, rank, world, local_rank)
    else:
        result = bench_single(args, torch, ga, devgraph, local_rank)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_single(args, torch, ga, devgraph, device_index):
    n = 1 << args.scale
    t0 = time.time()
    ro, ci = devgraph.rmat_csr_device(args.scale, args.edge_factor, args.seed)
    torch.cuda.synchronize()
    m = int(ci.shape[0])
    build_s = time.time() - t0
    src0, maxdeg = devgraph.largest_degree_source(ro)
    sources = [src0] + devgraph.seeded_sources(ro, 64, args.seed)
    deg = (ro[1:] - ro[:-1]).long()

    prob = ga.BfsProblem(mark_pred=False, idempotence=True, instrument=False, device=device_index)
    prob.init_device(n, m, ro.data_ptr(), ci.data_ptr())
    d_labels, _ = prob.device_results()
    labels_t = devgraph.as_tensor(d_labels, n)

    def step(p, k):
        s = sources[k % len(sources)]
        p.reset(s)
        return p.enact(s)

    for k in range(args.warmup):
        step(prob, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enact_ms = 0.0
    for k in range(args.steps):
        enact_ms += step(prob, k)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0

    # per-source visited counts (untimed): rerun each distinct source used, read labels on the device
    used = [sources[k % len(sources)] for k in range(args.steps)]
    per_src = {}
    for s in sorted(set(used)):
        prob.reset(s)
        prob.enact(s)
        vis = labels_t > -1
        per_src[s] = (int(vis.sum()), int(deg[vis].sum()), prob.stats()["search_depth"])
    edges_total = sum(per_src[s][1] for s in used)
    nodes_total = sum(per_src[s][0] for s in used)
    value = edges_total / (wall * 1e6)
    enact_mteps = edges_total / (enact_ms * 1e3)

    # instrumented pass for the roofline of the dominant kernel (advance::LoadBalancedKernel)
    iprob = ga.BfsProblem(False, True, instrument=True, device=device_index)
    iprob.init_device(n, m, ro.data_ptr(), ci.data_ptr())
    kernel_ms, launches, balg = 0.0, 0, 0.0
    for k in range(min(args.steps, len(sources))):
        s = sources[k % len(sources)]
        iprob.reset(s)
        iprob.enact(s)
        st = iprob.stats()
        kernel_ms += st["kernel_ms"]
        launches += st["kernel_launches"]
        nv, ev, _ = per_src.get(s) or (0, 0, 0)
        if s not in per_src:
            il, _ = iprob.device_results()
            vis = devgraph.as_tensor(il, n) > -1
            nv, ev = int(vis.sum()), int(deg[vis].sum())
        balg += 4.0 * ev + 20.0 * nv
    achieved = balg / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0   # GB/s
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 5), "traffic": None,
                "kernel": "advance::LoadBalancedKernel", "launches": launches,
                "avg_launch_ms": round(kernel_ms / max(launches, 1), 5),
                "alg_bytes_per_launch": round(balg / max(launches, 1), 1)}
    iprob.close()

    cpu = None
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
                prob.enact(s)
                got, _ = prob.extract()
                parity = bool((got == ref_labels).all())
        cpu = {"value": round(cpu_edges / (cpu_s * 1e6), 2), "unit": "MTEPS", "cores": 1, "kind": "port",
               "sample": "%d serial deque BFS runs (oracle port of SimpleReferenceBfs) on the same scale-%d graph, "
                         "%.1f s CPU" % (args.cpu_baseline_runs, args.scale, cpu_s)}
    prob.close()

    depth = per_src[used[0]][2]
    return {
        "metric": "MTEPS (million traversed edges/sec) BFS R-MAT scale-%d" % args.scale,
        "value": round(value, 2), "unit": "MTEPS", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "BFS idempotent, R-MAT scale-%d (a=.55 b=.2 c=.2 d=.05, %d pairs/vertex mirrored, "
                               "seed 0x%x): n=%d, m=%d directed edges; sources: largest-degree + 64 seeded"
                               % (args.scale, args.edge_factor, args.seed, n, m),
                   "search_depth_src0": depth, "graph_build_s": round(build_s, 2), "max_degree": maxdeg},
        "enact_mteps": round(enact_mteps, 2), "enact_ms_per_step": round(enact_ms / args.steps, 4),
        "edges_visited_per_step": edges_total // args.steps, "nodes_visited_per_step": nodes_total // args.steps,
        "parity_vs_oracle": parity,
        "roofline": roofline, "cpu_baseline": cpu,
    }


if __name__ == "__main__":
    main()
