"""SSSP parity on the GPU: unsigned 32-bit distances must equal the oracle's Dijkstra bit for bit (integer path, no
tolerance needed); predecessors are checked as valid parents (dist[pred] + w(pred, v) == dist[v])."""
import os

import numpy as np
import pytest

import gunrockinst_amd as ga
from oracle import gr_oracle as o

pytestmark = pytest.mark.gpu


def _run(g, w, src, mark_pred=True, delta_factor=16, instrument=False):
    p = ga.SsspProblem(mark_pred, instrument).init(g.nodes, g.row_offsets, g.col_indices, w, delta_factor)
    p.reset(src)
    ms = p.enact(src)
    dist, preds = p.extract()
    st = p.stats()
    p.close()
    return dist, preds, st, ms


def _check(g, w, src, **kw):
    dist, preds, st, _ = _run(g, w, src, **kw)
    ref, _ = o.sssp(g, src, w)
    assert np.array_equal(dist, ref)
    if preds is not None:
        assert o.check_sssp_preds(g, src, dist, preds, w) == 0
    return st


def test_fixture7_known_answer(golden, capfd):
    f = golden["fixture7"]
    ro, ci = np.array(f["row_offsets"], np.int32), np.array(f["col_indices"], np.int32)
    w = np.array(f["sssp_weights"], np.uint32)
    dist, preds = ga.gunrock_sssp(7, ro, ci, w, src=0, mark_pred=True, delta_factor=1)
    assert dist.tolist() == [0, 39, 6, 16, 50, 29, 64]
    k = f["ctest_sssp"]
    assert dist[k["node"]] == k["label"] and preds[k["node"]] == k["pred"]       # CMakeLists.txt:227-229
    assert "GPU Single-Source Shortest Path finished" in capfd.readouterr().out
    g = o.Csr(7, ro, ci)
    for src in range(7):
        for df in (1, 16, 32):
            _check(g, w, src, delta_factor=df)
        _check(g, w, src, mark_pred=False)


def test_pattern_files_have_unit_weights(golden_dir):
    # pattern .mtx -> all weights 1 (market.cuh:146-148): SSSP distances = BFS depths
    gc = o.build_market(os.path.join(golden_dir, "chesapeake.mtx"), undirected=True)
    dist, _, _, _ = _run(gc, gc.weights_u32, 3)
    labels, _, _ = o.bfs(gc, 3)
    assert np.array_equal(dist, np.where(labels < 0, 0xFFFFFFFF, labels).astype(np.uint32))


def test_wrapping_weights_terminate_and_match_saturating_oracle(golden_dir):
    # bips98_606 carries real values, some negative: truncated to ints and reinterpreted as unsigned they are ~4e9.
    # The reference's unchecked add would wrap; the engine rejects wrapped candidates like the oracle's saturating add.
    g = o.build_market(os.path.join(golden_dir, "bips98_606.mtx"), undirected=True)
    w = g.weights_u32
    assert (w > 0x7FFFFFFF).any()
    for src in (0, 566):
        _check(g, w, src, delta_factor=16)


def test_bips_real_weights_as_loaded(golden_dir):
    # bips98_606 carries real values; the loader truncates them to integers (market.cuh:136-141) and SSSP reinterprets
    # the ints as unsigned.  Use |value| + 1 to stay in the overflow-free regime where the oracle's saturating add and
    # the engine's wrapping add agree.
    g = o.build_market(os.path.join(golden_dir, "bips98_606.mtx"), undirected=True)
    w = (np.abs(g.edge_values.astype(np.int64)) % 1000 + 1).astype(np.uint32)
    for src in (0, 566, 7134):
        _check(g, w, src, delta_factor=16)


@pytest.mark.parametrize("scale,ef,wmax,df", [(10, 8, 64, 16), (14, 8, 64, 32), (16, 16, 64, 16), (16, 4, 1000, 1),
                                              (18, 8, 64, 16)])
def test_rmat_parity(scale, ef, wmax, df):
    g = o.rmat_seeded(scale, ef << scale)
    rng = np.random.default_rng(scale * 7 + wmax)
    w = rng.integers(1, wmax + 1, g.edges, dtype=np.uint32)
    src, _ = o.highest_degree_node(g)
    deg = np.diff(g.row_offsets)
    others = rng.choice(np.nonzero(deg > 0)[0], 2)
    for s in [src] + others.tolist():
        st = _check(g, w, int(s), delta_factor=df)
        assert st["relaxed_edges"] >= o.bfs_stats(g, o.bfs(g, int(s))[0])[1]      # every reachable edge relaxed at least once


def test_edge_cases():
    g = o.Csr(4, [0, 0, 1, 1, 1], [0])
    for src in range(4):
        _check(g, np.array([5], np.uint32), src)
    _check(o.Csr(1, [0, 0], []), np.empty(0, np.uint32), 0)
    n = 3000                                               # long weighted path: thousands of buckets
    ro = np.minimum(np.arange(n + 1), n - 1).astype(np.int32)
    g = o.Csr(n, ro, np.arange(1, n, dtype=np.int32))
    w = (np.arange(n - 1) % 97 + 1).astype(np.uint32)
    _check(g, w, 0, delta_factor=1)
    _check(g, w, 0, delta_factor=64)
    # zero-weight edges and widely spread weights (far pile with big bucket gaps)
    gr = o.rmat_seeded(12, 8 << 12)
    rng = np.random.default_rng(5)
    w = rng.choice(np.array([0, 1, 7, 100000, 3000000], np.uint32), gr.edges)
    _check(gr, w, o.highest_degree_node(gr)[0], delta_factor=1)
    _check(gr, w, o.highest_degree_node(gr)[0], delta_factor=16)


def test_instrumented_and_rerun():
    g = o.rmat_seeded(14, 8 << 14)
    w = np.random.default_rng(2).integers(1, 65, g.edges, dtype=np.uint32)
    p = ga.SsspProblem(True, True).init(g.nodes, g.row_offsets, g.col_indices, w, 16)
    deg = np.diff(g.row_offsets)
    for src in np.nonzero(deg > 0)[0][:3].tolist():
        p.reset(src)
        ms = p.enact(src)
        dist, preds = p.extract()
        assert np.array_equal(dist, o.sssp(g, src, w)[0])
        assert o.check_sssp_preds(g, src, dist, preds, w) == 0
    st = p.stats()
    assert st["kernel_launches"] > 0 and 0 < st["kernel_ms"] <= ms and st["delta"] > 0
    p.close()


@pytest.mark.parametrize("mark_pred", [False, True])
@pytest.mark.parametrize("pull_min_edges", [1, -1, 0])
def test_pull_relaxation_levels(mark_pred, pull_min_edges):
    # dense levels relaxed by PULLING over the weighted in-neighbour lists (reducing advance, MINIMUM): distances must equal the
    # oracle's Dijkstra bit for bit, predecessors must be tight parents; pull_min_edges 1 = every level with more than one edge
    rng = np.random.default_rng(7)
    for scale, und in [(12, True), (15, True), (14, False)]:
        g = o.rmat_seeded(scale, 8 << scale, undirected=und)
        w = rng.integers(1, 65, g.edges, dtype=np.uint32)
        deg = np.diff(g.row_offsets)
        for src in [int(o.highest_degree_node(g)[0]), int(np.nonzero(deg > 0)[0][-1])]:
            p = ga.SsspProblem(mark_pred).init(g.nodes, g.row_offsets, g.col_indices, w)
            p.set_inverse_graph(pull_min_edges=pull_min_edges)      # weighted transpose built on the device
            p.reset(src)
            p.enact(src)
            dist, preds = p.extract()
            ref, _ = o.sssp(g, src, w)
            assert np.array_equal(dist, ref)
            if mark_pred:
                assert o.check_sssp_preds(g, src, dist, preds, w) == 0
            if pull_min_edges == 1 and deg[src] > 1:
                assert p.pull_levels() > 0
            if pull_min_edges == 0:
                assert p.pull_levels() == 0
            # (pull_min_edges -1 = default rule: a level pulls only when its frontier holds 3/4 of the edges -- on these small
            #  graphs the hub's second level can; only the distances are asserted)
            p.close()
