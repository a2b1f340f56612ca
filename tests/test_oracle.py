"""Pins the CPU oracle (oracle/gr_oracle.c) against the reference's own fixtures and the golden values
captured from the reference's host code.  CPU only."""
import os

import numpy as np

from oracle import gr_oracle as o


def test_fixture7_loader_matches_shared_lib_arrays(golden, golden_dir):
    f = golden["fixture7"]
    g = o.build_market(os.path.join(golden_dir, f["mtx"]))          # directed, like tests/cc (test_cc.cu:415)
    assert (g.nodes, g.edges) == (f["nodes"], f["edges"])
    assert g.row_offsets.tolist() == f["row_offsets"]               # pins the "col row" read order + sort
    assert g.col_indices.tolist() == f["col_indices"]
    assert g.edge_values.tolist() == [1] * 15                       # pattern entries -> 1 (market.cuh:146-148)


def test_fixture7_known_answers(golden):
    f = golden["fixture7"]
    g = o.Csr(7, f["row_offsets"], f["col_indices"], f["sssp_weights"])
    labels, preds, depth = o.bfs(g, 0, want_preds=True)
    assert labels.tolist() == f["bfs_src0_labels"]
    assert labels[f["ctest_bfs"]["node"]] == f["ctest_bfs"]["label"]      # CMakeLists.txt:215-217
    assert o.check_bfs_preds(g, 0, labels, preds) == 0
    dist, sp = o.sssp(g, 0)
    assert dist[f["ctest_sssp"]["node"]] == f["ctest_sssp"]["label"]      # CMakeLists.txt:227-229
    assert sp[f["ctest_sssp"]["node"]] == f["ctest_sssp"]["pred"]
    assert o.check_sssp_preds(g, 0, dist, sp) == 0
    comp, count = o.cc(g)
    assert comp[f["ctest_cc"]["node"]] == f["ctest_cc"]["component"]      # CMakeLists.txt:223-225
    assert count == 1


def test_test_cc_graph(golden, golden_dir):
    f = golden["test_cc"]
    g = o.build_market(os.path.join(golden_dir, f["mtx"]), undirected=True)
    assert g.nodes == f["nodes"]
    assert o.highest_degree_node(g) == (f["max_degree_node"], f["max_degree"])
    assert o.average_degree(g) == f["avg_degree"]
    assert g.col_indices[:11].tolist() == f["first_cols"]
    comp, count = o.cc(g)
    assert comp.tolist() == f["cc_labels"] and count == 2
    comp2, count2, ih, ij = o.cc_reference_schedule(g)
    assert comp2.tolist() == f["cc_labels"] and count2 == 2 and ih >= 2 and ij >= 2
    # the reference's tests/cc driver loads the file DIRECTED (test_cc.cu:415): hooks still see weak components
    gd = o.build_market(os.path.join(golden_dir, f["mtx"]))
    assert o.cc(gd)[0].tolist() == f["cc_labels"]


def test_bips98_606(golden, golden_dir):
    f = golden["bips98_606"]
    g = o.build_market(os.path.join(golden_dir, f["mtx"]), undirected=True)
    assert (g.nodes, g.edges) == (f["nodes"], f["edges"])
    assert o.highest_degree_node(g) == (f["max_degree_node"], f["max_degree"])
    assert o.average_degree(g) == f["avg_degree"]
    labels, _, depth = o.bfs(g, 0)
    nv, ev = o.bfs_stats(g, labels)
    b = f["bfs_src0"]
    assert (depth, nv, ev) == (b["depth"], b["nodes_visited"], b["edges_visited"])
    assert labels[:10].tolist() == b["labels_head"]
    labels, _, depth = o.bfs(g, 566)
    assert depth == f["bfs_src566"]["depth"] and labels[:10].tolist() == f["bfs_src566"]["labels_head"]
    comp, count = o.cc(g)
    comp2, count2, _, _ = o.cc_reference_schedule(g)
    assert count == count2 and (comp == comp2).all()
    assert (comp <= np.arange(g.nodes)).all()


def test_rmat_libc_stream(golden):
    f = golden["rmat_libc"]
    g = o.rmat_reference(f["nodes"], f["edges_in"], undirected=f["undirected"], srand=1)
    assert g.edges == f["edges"]
    assert g.row_offsets[1:5].tolist() == f["row_offsets_1_4"]


def test_csr_invariants_and_undirected_symmetry():
    g = o.rmat_seeded(10, 8 << 10)
    assert g.row_offsets[0] == 0 and g.row_offsets[-1] == g.edges
    src = np.repeat(np.arange(g.nodes), np.diff(g.row_offsets))
    assert (src != g.col_indices).all()                                   # no self loops (csr.cuh:272-288)
    key = src.astype(np.int64) << 32 | g.col_indices
    assert (np.diff(key) > 0).all()                                       # sorted, no duplicates
    rev = g.col_indices.astype(np.int64) << 32 | src
    assert np.array_equal(np.sort(rev), key)                              # symmetric


def test_seeded_rmat_is_chunk_invariant():
    r0, c0 = o.rmat_seeded_coo(12, 0, 1000)
    r1, c1 = o.rmat_seeded_coo(12, 400, 600)
    assert np.array_equal(r0[400:], r1) and np.array_equal(c0[400:], c1)


def test_sssp_vs_bfs_unit_weights(golden_dir):
    g = o.build_market(os.path.join(golden_dir, "chesapeake.mtx"), undirected=True)
    labels, _, _ = o.bfs(g, 3)
    dist, preds = o.sssp(g, 3)
    assert np.array_equal(np.where(labels < 0, 0xFFFFFFFF, labels).astype(np.uint32), dist)
    assert o.check_sssp_preds(g, 3, dist, preds) == 0


def test_bc_oracle_reproduces_reference_ctest_answer(golden):
    # CMakeLists.txt:219-221: test_bc on the undirected 7-vertex graph prints BC 0.500000 for node 0 (all sources, halved)
    f = golden["bc_undirected7"]
    bc, _ = o.bc(o.Csr(7, f["row_offsets"], f["col_indices"]), -1)
    assert abs(bc[0] - f["bc_node0"]) < 1e-12
    assert ("%f" % bc[0]) == "0.500000"
    # one source on a path 0-1-2: the middle vertex carries the dependency of the far end (1), halved
    g = o.Csr(3, [0, 1, 3, 4], [1, 0, 2, 1])
    assert o.bc(g, 0)[0].tolist() == [0.0, 0.5, 0.0]


def test_parallel_bfs_baseline_gives_the_serial_labels():
    g = o.rmat_seeded(14, 8 << 14)
    for src in (o.highest_degree_node(g)[0], 3, 12345):
        ref, _, _ = o.bfs(g, src)
        for threads in (1, 3, 0):
            labels, used = o.bfs_parallel(g, src, threads)
            assert np.array_equal(labels, ref)
            assert used >= 1


def test_pagerank_and_topk_restatements(golden):
    f = golden["fixture7"]
    g = o.Csr(7, f["row_offsets"], f["col_indices"])
    # TopK: pinned by the reference's ctest answer "Node ID.*2.*: in_degrees.*3.*: out_degrees.*3" (CMakeLists.txt:235-237),
    # inputs of shared_lib_tests/test_topk.c:27-31
    ids, ind, outd = o.topk(g, 3, [0, 1, 2, 5, 7, 9, 12, 15])
    assert (int(ids[0]), int(ind[0]), int(outd[0])) == (2, 3, 3)
    # PageRank on shared_lib_tests/test_pr.c's inputs: vertices 6, 5, 3 are peeled off in three rounds, the rest converges
    rank, deg, iters = o.pagerank(g, 0, 0.85, 0.01, 20)
    assert deg.tolist() == [2, 3, 1, -1, 1, -1, -1] and iters == 10
    assert int(np.argmax(rank)) == 2 and abs(rank[2] - 0.357589) < 1e-5
    # fixed point of the 4-vertex system that is left (solved by hand in DESIGN.md): r2 = 0.3981
    far, _, _ = o.pagerank(g, 0, 0.85, 0.0, 200)
    assert abs(far[2] - 0.39810) < 1e-4


def test_pagerank_ctest_answer_is_stale(golden):
    """The reference's ctest expects "Node ID.*2.*: Page Rank.*0.402378" from shared_lib_tests/test_pr.c (CMakeLists.txt:231-233).
    The code in the tree cannot print it: from every starting point tried, vertex 2's rank moves monotonically towards the fixed
    point 0.3981 and never passes through 0.402378 -- the regex predates the code (like gunrock/app/sssp/sssp_app.cu, which no
    longer compiles).  So PageRank parity is UNPINNED; this test keeps the evidence."""
    f = golden["fixture7"]
    g = o.Csr(7, f["row_offsets"], f["col_indices"])
    seen = []
    for src in (0, -1):
        for iters in range(1, 41):
            rank, _, _ = o.pagerank(g, src, 0.85, 0.0, iters)
            seen.append(float(rank[2]))
    assert all(abs(x - 0.402378) > 1e-4 for x in seen)
