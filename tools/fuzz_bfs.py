"""Randomised parity sweep of the BFS schedules against the oracle: python tools/fuzz_bfs.py [seconds] [seed]
Graph families: R-MAT of random scale / edge factor (directed and mirrored), grids with shortcuts, stars, chains, forests of
small components; traversal modes 0 / 1 / 2, both predecessor settings, random tuning knobs, random sources."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gunrockinst_amd as ga
from oracle import gr_oracle as o

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def from_coo(n, rows, cols, mirror):
    if mirror:
        rows, cols = np.concatenate([rows, cols]), np.concatenate([cols, rows])
    hg = ga.HostGraph.from_coo(n, rows.astype(np.int32), cols.astype(np.int32))
    return o.Csr(n, np.array(hg.row_offsets), np.array(hg.col_indices))


def make_graph():
    kind = rng.integers(0, 6)
    if kind == 0:
        scale = int(rng.integers(6, 17)); ef = int(rng.integers(1, 33))
        return "rmat%d/%d" % (scale, ef), o.rmat_seeded(scale, ef << scale, undirected=bool(rng.integers(0, 2))), None
    if kind == 1:
        side = int(rng.integers(3, 300)); n = side * side
        idx = np.arange(n).reshape(side, side)
        rows = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel()]); cols = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel()])
        extra = int(rng.integers(0, max(n // 50, 1)))
        rows = np.concatenate([rows, rng.integers(0, n, extra)]); cols = np.concatenate([cols, rng.integers(0, n, extra)])
        return "grid%d+%d" % (side, extra), from_coo(n, rows, cols, True), True
    if kind == 2:
        n = int(rng.integers(2, 60000))
        return "star%d" % n, from_coo(n, np.zeros(n - 1, np.int64), np.arange(1, n), bool(rng.integers(0, 2))), None
    if kind == 3:
        n = int(rng.integers(2, 5000))
        return "chain%d" % n, from_coo(n, np.arange(n - 1), np.arange(1, n), bool(rng.integers(0, 2))), None
    if kind == 4:
        n = int(rng.integers(10, 200000)); m = int(n * rng.uniform(0.3, 3.0))
        return "sparse%d/%d" % (n, m), from_coo(n, rng.integers(0, n, m), rng.integers(0, n, m), bool(rng.integers(0, 2))), None
    n = int(rng.integers(100, 30000)); hubs = int(rng.integers(1, 6)); m = int(n * rng.uniform(1, 6))
    rows = rng.integers(0, n, m); cols = np.where(rng.random(m) < 0.4, rng.integers(0, hubs, m), rng.integers(0, n, m))
    return "hubs%d/%d" % (n, hubs), from_coo(n, rows, cols, True), True


t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    name, g, symmetric = make_graph()
    if g.nodes < 1:
        continue
    if symmetric is None:
        ro, ci = np.asarray(g.row_offsets), np.asarray(g.col_indices)
        src_of = np.repeat(np.arange(g.nodes), np.diff(ro))
        key = src_of.astype(np.int64) * g.nodes + ci
        symmetric = bool(np.array_equal(np.sort(key), np.sort(ci.astype(np.int64) * g.nodes + src_of)))
    deg = np.diff(g.row_offsets)
    for mark_pred in (False, True):
        p = ga.BfsProblem(mark_pred, bool(rng.integers(0, 2))).init(g.nodes, g.row_offsets, g.col_indices)
        modes = [0, 1, 2]
        if symmetric and rng.integers(0, 2):
            p.set_inverse_graph()
        else:  # what gunrock_bfs_func does: symmetry check on the device, transpose built there for a directed graph
            enabled, built, _ = p.auto_inverse()
            if not enabled or built == bool(symmetric):
                print("AUTO-INVERSE WRONG", name, "n", g.nodes, "m", g.edges, "symmetric", symmetric, "enabled", enabled, "built", built); sys.exit(1)
        if rng.integers(0, 2):
            p.set_tuning(tail_edge_limit=int(rng.choice([0, 64, 1024, 8192])))
            p.set_twc_limit(int(rng.choice([0, 16, 500, 8192, 1 << 20])))
            p.set_persistent_limit(int(rng.choice([0, 1 << 12, 1 << 20])))
            p.set_binned_min_edges(int(rng.choice([0, 1, 1000, 1 << 23])))
            if True:
                p.set_head_pass(int(rng.choice([-1, 1, 1000])), int(rng.choice([-1, 0])))
        # round 3: deferred labels (on / off / a 4-bitmap pool that flushes mid-search), sweeps per round trip, direction rules,
        # speculative emit, queue-emitting sweeps
        p.set_label_deferral(int(rng.integers(0, 2)), int(rng.choice([0, 4, 5, 12])))
        p.set_option("chain_sweeps", int(rng.choice([0, 1, 2, 3, 6])))
        p.set_option("speculative_emit", int(rng.integers(0, 2)))
        p.set_option("chain_closing", int(rng.integers(0, 2)))
        p.set_option("emit_queue_factor", float(rng.choice([0.0, 32.0, 1e9])))
        p.set_option("sparse_sweep_div", int(rng.choice([0, 1, 16])))
        if rng.integers(0, 2):
            p.set_tuning(alpha=float(rng.choice([0.5, 10.0, 1e6])), beta=float(rng.choice([1e-3, 24.0, 4000.0, 1e9])),
                         lite_factor=float(rng.choice([0.0, 12.0, 1e9])))
        srcs = [int(np.argmax(deg)), int(rng.integers(0, g.nodes)), int(rng.integers(0, g.nodes))]
        for src in srcs:
            ref, _, depth = o.bfs(g, src)
            for mode in modes:
                p.reset(src)
                p.enact(src, traversal_mode=mode)
                labels, preds = p.extract()
                if not np.array_equal(labels, ref):
                    print("MISMATCH", name, "n", g.nodes, "m", g.edges, "src", src, "mode", mode, "mark_pred", mark_pred); sys.exit(1)
                if mark_pred and o.check_bfs_preds(g, src, labels, preds) != 0:
                    print("BAD PARENTS", name, "n", g.nodes, "m", g.edges, "src", src, "mode", mode); sys.exit(1)
                cases += 1
        p.close()
print("fuzz ok:", cases, "searches")
