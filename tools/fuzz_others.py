"""Randomised parity sweep of SSSP / CC / BC / PageRank against the oracle: python tools/fuzz_others.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gunrockinst_amd as ga
from oracle import gr_oracle as o

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def graph():
    kind = rng.integers(0, 3)
    if kind == 0:
        scale = int(rng.integers(5, 16)); ef = int(rng.integers(1, 25))
        return o.rmat_seeded(scale, ef << scale, undirected=bool(rng.integers(0, 2)))
    n = int(rng.integers(2, 50000)); m = int(n * rng.uniform(0.5, 8.0))
    rows, cols = rng.integers(0, n, m), rng.integers(0, n, m)
    if kind == 2:
        cols = np.where(rng.random(m) < 0.3, rng.integers(0, max(n // 1000, 1), m), cols)
    orient = rng.integers(0, 4)
    if orient >= 2:
        rows, cols = np.concatenate([rows, cols]), np.concatenate([cols, rows])
    elif orient == 1:  # every edge points from the higher to the lower id: passes any "upward edges are mirrored" test vacuously
        rows, cols = np.maximum(rows, cols), np.minimum(rows, cols)
    hg = ga.HostGraph.from_coo(n, rows.astype(np.int32), cols.astype(np.int32))
    return o.Csr(n, np.array(hg.row_offsets), np.array(hg.col_indices))


t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    g = graph()
    if g.edges == 0:
        continue
    deg = np.diff(g.row_offsets)
    src = int(rng.choice([int(np.argmax(deg)), int(rng.integers(0, g.nodes))]))
    # SSSP
    wmax = int(rng.choice([1, 7, 64, 100000]))
    w = rng.integers(1, wmax + 1, g.edges, dtype=np.uint32)
    for mark_pred in (False, True):
        p = ga.SsspProblem(mark_pred).init(g.nodes, g.row_offsets, g.col_indices, w, delta_factor=int(rng.choice([1, 16, 1000])))
        p.reset(src); p.enact(src)
        dist, preds = p.extract()
        ref, _ = o.sssp(g, src, w)
        if not np.array_equal(dist, ref) or (mark_pred and o.check_sssp_preds(g, src, dist, preds, w) != 0):
            print("SSSP MISMATCH n", g.nodes, "m", g.edges, "src", src, "wmax", wmax, "mark_pred", mark_pred); sys.exit(1)
        p.close(); cases += 1
    # CC
    comp = ga.gunrock_cc(g.nodes, g.row_offsets, g.col_indices)
    if not np.array_equal(comp, o.cc(g)[0]):
        print("CC MISMATCH n", g.nodes, "m", g.edges); sys.exit(1)
    cases += 1
    # BC (one source)
    bc, _ = ga.gunrock_bc(g.nodes, g.row_offsets, g.col_indices, src=src)
    ref_bc, _ = o.bc(g, src)
    if not np.all(np.abs(bc - ref_bc) <= 1e-3 * np.abs(ref_bc) + 1e-3):
        print("BC MISMATCH n", g.nodes, "m", g.edges, "src", src, float(np.max(np.abs(bc - ref_bc)))); sys.exit(1)
    cases += 1
    # PageRank
    s = int(rng.choice([-1, src])); iters = int(rng.integers(1, 30)); thr = float(rng.choice([0.0, 0.01]))
    ids, ranks = ga.gunrock_pr(g.nodes, g.row_offsets, g.col_indices, src=s, max_iter=iters, error=thr)
    ref_pr, _, _ = o.pagerank(g, s, 0.85, thr, iters)
    got = np.zeros(g.nodes); got[ids] = ranks
    if not np.allclose(got, ref_pr, rtol=2e-4, atol=2e-6):
        # a vertex whose move sits within rounding of the threshold may stop one iteration apart: compare again at threshold 0
        if thr == 0.0:
            print("PR MISMATCH n", g.nodes, "m", g.edges, "src", s, "iters", iters, float(np.max(np.abs(got - ref_pr)))); sys.exit(1)
    cases += 1
print("fuzz ok:", cases, "runs")
