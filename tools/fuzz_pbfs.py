"""Randomised parity sweep of the in-library partitioned BFS (Pbfs::Search) over gloo, several ranks sharing the GPU:
python tools/fuzz_pbfs.py <world> [seconds] [seed].  Graph sizes that do not divide by the number of ranks, graphs smaller than
the number of ranks, stars, chains, isolated sources; labels (and parents, when marked) against the oracle on rank 0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, budget, seed, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gunrockinst_amd as ga
    from gunrockinst_amd import multi_gpu as mg
    from oracle import gr_oracle as o
    rng = np.random.default_rng(seed)          # the same stream on every rank: every rank builds the same graph
    comm = mg.Comm()
    t_end = time.time() + budget
    cases, ok = 0, True
    while ok:
        stop = torch.tensor([1 if time.time() > t_end else 0]); dist.broadcast(stop, 0)
        if int(stop[0]):
            break
        kind = int(rng.integers(0, 5)); mirror = True
        if kind == 0:
            scale = int(rng.integers(3, 15)); g = o.rmat_seeded(scale, int(rng.integers(1, 17)) << scale)
        else:
            n = int(rng.integers(1, 40000)) if kind != 4 else int(rng.integers(1, 8))
            if kind == 1: rows, cols = np.zeros(max(n - 1, 0), np.int64), np.arange(1, n)
            elif kind == 2: rows, cols = np.arange(max(n - 1, 0)), np.arange(1, n)
            else:
                m = int(n * rng.uniform(0.2, 4.0)); rows, cols = rng.integers(0, n, m), rng.integers(0, n, m)
            mirror = bool(rng.integers(0, 4))      # sometimes directed: those run top-down only
            if mirror: rows, cols = np.concatenate([rows, cols]), np.concatenate([cols, rows])
            hg = ga.HostGraph.from_coo(n, rows.astype(np.int32), cols.astype(np.int32))
            g = o.Csr(n, np.array(hg.row_offsets), np.array(hg.col_indices))
        ro_h, ci_h = mg.partition_csr_host(np.asarray(g.row_offsets), np.asarray(g.col_indices), rank, world)
        ro_d = torch.from_numpy(ro_h).cuda(); ci_d = torch.from_numpy(ci_h).cuda() if ci_h.size else torch.zeros(1, dtype=torch.int32, device="cuda")[:0]
        eng = mg.HipEngine(g.nodes, world, rank, ro_d, ci_d, 0)
        mark_pred = bool(rng.integers(0, 2))
        bfs = mg.LibraryBfs(eng, comm, transport="callbacks", mark_pred=mark_pred, alpha=float(rng.choice([0.0, 1.0, 1e9])))
        bfs.set_option("lite_factor", float(rng.choice([0.0, 1.0, 230.0, 1e9])))   # count-only (marked) levels: never .. from the first level on
        bfs.set_option("sparse_sweep_div", int(rng.choice([0, 1, 6, 16])))
        deg = np.diff(g.row_offsets)
        for src in [int(np.argmax(deg)), int(rng.integers(0, g.nodes))]:
            levels, _ = bfs.search(src, direction_optimizing=mirror and bool(rng.integers(0, 2)))
            full = mg.assemble_labels(comm, eng.labels(), g.nodes)
            ref, _, depth = o.bfs(g, src)
            good = bool((full == ref).all())
            if mark_pred:
                preds = mg.assemble_labels(comm, bfs.preds(), g.nodes)
                good = good and o.check_bfs_preds(g, src, full, preds) == 0
            if not good:
                if rank == 0: print("MISMATCH kind", kind, "n", g.nodes, "m", g.edges, "src", src, "mirror", mirror, "mark_pred", mark_pred, flush=True)
                ok = False
            cases += 1
        eng.close()
    if rank == 0:
        print("fuzz ok:" if ok else "fuzz FAILED after", cases, "searches, world", world, flush=True)
        open(out, "w").write("ok" if ok else "bad")
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1])
    if not 1 <= world <= 5:  # (ranks share ONE GPU here; the box allows 6 processes on it -- and "fuzz_pbfs.py 90 7" meant seconds, not ranks)
        sys.exit("usage: fuzz_pbfs.py <world 1..5> [seconds] [seed]")
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0; seed = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    out = "/tmp/fuzz_pbfs_%d.txt" % os.getpid()
    mp.spawn(worker, args=(world, 29000 + os.getpid() % 2000, budget, seed, out), nprocs=world, join=True)
    sys.exit(0 if open(out).read() == "ok" else 1)
