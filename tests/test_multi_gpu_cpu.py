"""The N > 1 path on CPU: world_size 2 and 3 over gloo.  The level loop, ownership rule, count/id exchange, bitmap
all-gather, termination all-reduce and label assembly of gunrockinst_amd/multi_gpu.py run for real; the local compute
steps are the numpy test double (tests/_numpy_engine.py).  Results are compared with the oracle's serial BFS."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, scale, direction_optimizing, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gunrockinst_amd import multi_gpu as mg
    from oracle import gr_oracle as o
    from _numpy_engine import NumpyEngine

    g = o.rmat_seeded(scale, 8 << scale)
    ro, ci = mg.partition_csr_host(g.row_offsets, g.col_indices, rank, world)
    assert ro.shape[0] - 1 == mg.local_count(g.nodes, world, rank)
    comm = mg.Comm()
    eng = NumpyEngine(g.nodes, world, rank, ro, ci)
    bfs = mg.PartitionedBfs(eng, comm, g.nodes, g.edges, alpha=14.0 if direction_optimizing != "always" else 1e9,
                            beta=24.0 if direction_optimizing != "always" else 1.0)
    src, _ = o.highest_degree_node(g)
    ok = True
    for s in (src, int(np.nonzero(np.diff(g.row_offsets) > 0)[0][-1])):
        levels = (bfs.run_gather(s) if direction_optimizing == "gather" else bfs.run(s, True, sticky_bottom_up=True) if direction_optimizing == "sticky"
                  else bfs.run(s, direction_optimizing=bool(direction_optimizing)))
        full = mg.assemble_labels(comm, eng.labels(), g.nodes)
        ref, _, depth = o.bfs(g, s)
        ok = ok and bool((full == ref).all()) and levels in (depth - 1, depth)
        kinds = {k for k, _, _ in bfs.trace}
        if direction_optimizing:
            ok = ok and "bottom-up" in kinds
    if rank == 0:
        with open(out, "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,dobfs", [(2, False), (2, True), (3, True), (2, "always"), (2, "gather"), (3, "gather"), (2, "sticky"), (3, "sticky")])
def test_partitioned_bfs_over_gloo(tmp_path, world, dobfs):
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), 9, dobfs, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_ownership_helpers():
    from gunrockinst_amd import multi_gpu as mg
    n, p = 1003, 8
    assert sum(mg.local_count(n, p, r) for r in range(p)) == n
    v = np.arange(n)
    for r in range(p):
        mine = v[mg.owner_of(v, p) == r]
        assert mine.size == mg.local_count(n, p, r)
        assert (mg.local_id(mine, p) == np.arange(mine.size)).all()
    ro = np.array([0, 2, 2, 5, 6], np.int32)
    ci = np.array([1, 3, 0, 1, 3, 2], np.int32)
    r0 = mg.partition_csr_host(ro, ci, 0, 2)
    r1 = mg.partition_csr_host(ro, ci, 1, 2)
    assert r0[0].tolist() == [0, 2, 5] and r0[1].tolist() == [1, 3, 0, 1, 3]
    assert r1[0].tolist() == [0, 0, 1] and r1[1].tolist() == [2]
