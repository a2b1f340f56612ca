"""CC parity on the GPU: component ids must equal the oracle's bit for bit (both converge to the smallest vertex
id of each component); the reference itself only compares the component COUNT (tests/cc/test_cc.cu:315-320)."""
import os

import numpy as np
import pytest

import gunrockinst_amd as ga
from oracle import gr_oracle as o

pytestmark = pytest.mark.gpu


def _run(g, instrument=False):
    p = ga.CcProblem(instrument).init(g.nodes, g.row_offsets, g.col_indices)
    p.reset()
    ms = p.enact()
    ids, count = p.extract()
    st = p.stats()
    p.close()
    return ids, count, st, ms


def _check(g):
    ids, count, st, _ = _run(g)
    ref, ref_count = o.cc(g)
    assert np.array_equal(ids, ref)
    assert count == ref_count
    assert (ids <= np.arange(g.nodes)).all()
    return st


def test_fixture7_known_answer(golden, capfd):
    f = golden["fixture7"]
    ro, ci = np.array(f["row_offsets"], np.int32), np.array(f["col_indices"], np.int32)
    ids = ga.gunrock_cc(7, ro, ci)
    assert ids.tolist() == [0] * 7
    assert ids[f["ctest_cc"]["node"]] == f["ctest_cc"]["component"]          # CMakeLists.txt:223-225
    assert "GPU Connected Component finished in" in capfd.readouterr().out


def test_test_cc_mtx_directed_and_undirected(golden, golden_dir):
    f = golden["test_cc"]
    for und in (True, False):       # tests/cc loads the file directed (test_cc.cu:415); hooks see weak components
        g = o.build_market(os.path.join(golden_dir, f["mtx"]), undirected=und)
        ids, count, _, _ = _run(g)
        assert ids.tolist() == f["cc_labels"] and count == 2


def test_bips98_606_simple_example_path(golden_dir):
    g = o.build_market(os.path.join(golden_dir, "bips98_606.mtx"), undirected=True)
    st = _check(g)
    _, _, ih, ij = o.cc_reference_schedule(g)
    # same schedule family as the reference: sweep counts are of the same order (exact counts depend on race order; on mirrored
    # input the opening, the neighbour round and the row-form hooking sweeps are counted with the vertex sweeps)
    assert st["edge_sweeps"] <= 3 * ih + 2 and st["vertex_sweeps"] >= 3


def test_chesapeake_and_test_pr(golden_dir):
    for name, und in [("chesapeake.mtx", True), ("test_pr.mtx", False), ("test_bc.mtx", False)]:
        _check(o.build_market(os.path.join(golden_dir, name), undirected=und))


def test_edge_cases():
    _check(o.Csr(5, [0, 0, 0, 0, 0, 0], []))                      # no edges: every vertex its own component
    _check(o.Csr(1, [0, 0], []))
    n = 4000                                                       # long path: deep trees, many jump rounds
    ro = np.minimum(np.arange(n + 1), n - 1).astype(np.int32)
    _check(o.Csr(n, ro, np.arange(1, n, dtype=np.int32)))
    # reversed path (edges point to smaller ids) and a star with the hub as the largest id
    ro = np.concatenate(([0], np.arange(0, n))).astype(np.int32)
    _check(o.Csr(n, ro, np.arange(0, n - 1, dtype=np.int32)))
    hub = 70000
    ro = np.zeros(hub + 1, np.int32)
    ro[-1] = hub - 1
    _check(o.Csr(hub, ro, np.arange(0, hub - 1, dtype=np.int32)))


@pytest.mark.parametrize("scale,ef,und", [(10, 1, True), (12, 2, False), (14, 8, True), (16, 4, True), (18, 8, True)])
def test_rmat_parity(scale, ef, und):
    g = o.rmat_seeded(scale, ef << scale, undirected=und)
    _check(g)


def test_many_small_components():
    rng = np.random.default_rng(3)
    n = 50000
    rows = rng.integers(0, n, n // 2, dtype=np.int32)
    cols = (rows + rng.integers(1, 4, n // 2, dtype=np.int32)) % n
    g0 = ga.HostGraph.from_coo(n, rows, cols)
    _check(o.Csr(n, g0.row_offsets.copy(), g0.col_indices.copy()))


def test_instrumented_and_rerun():
    g = o.rmat_seeded(14, 8 << 14)
    p = ga.CcProblem(True).init(g.nodes, g.row_offsets, g.col_indices)
    ref = o.cc(g)[0]
    for _ in range(2):
        p.reset()
        ms = p.enact()
        ids, _ = p.extract()
        assert np.array_equal(ids, ref)
    st = p.stats()
    assert st["kernel_launches"] == st["edge_sweeps"] + st["vertex_sweeps"] and 0 < st["kernel_ms"] <= ms
    p.close()


@pytest.mark.parametrize("compact", ["1", "0"])
def test_mirrored_input_with_and_without_the_compact_edge_list(compact, monkeypatch):
    # mirrored graphs materialise only the from > to orientation of every edge for the hooking sweeps (cc_problem.hpp); with
    # GUNROCK_CC_COMPACT=0 both orientations stay and the from < to one is parked at first sight.  Same labels either way, also for
    # graphs with hubs, isolated vertices, many small components, and for a directed graph (never compacted).
    monkeypatch.setenv("GUNROCK_CC_COMPACT", compact)
    for g in (o.rmat_seeded(10, 4 << 10), o.rmat_seeded(16, 8 << 16), o.rmat_seeded(18, 2 << 18), o.rmat_seeded(15, 8 << 15, undirected=False)):
        st = _check(g)
        assert st["edge_sweeps"] + st["vertex_sweeps"] >= 3 and (compact == "1" or st["edge_sweeps"] >= 1)


@pytest.mark.parametrize("rowform,limit,rounds", [("1", None, None), ("1", "0", None), ("1", "3", "1"), ("0", None, None), ("1", None, "0"),
                                                  ("1", None, "5"), ("1", "0", "0")])
def test_row_form_hooking_sweeps_skip_the_giant_component(rowform, limit, rounds, monkeypatch):
    # mirrored input with a dominant component: vertices rooted at the sampled giant's root skip the hooking sweep, the others walk
    # their rows (cc_functor.hpp HookMaxRowFunctor).  GUNROCK_CC_ROW_LIMIT=0 / 3 makes rows "too long for one lane" so that the edge
    # form takes over mid-run; GUNROCK_CC_ROWFORM=0 switches the row form off; GUNROCK_CC_NEIGHBOUR_ROUNDS sets how many one-edge-per-
    # vertex hooking rounds run in front (0: the first full sweep stays in edge form).  Same labels in every case -- R-MAT (a giant component,
    # thousands of small ones, isolated vertices), a graph of two equal halves (no component holds most samples... one holds half),
    # a forest of small components (no giant: the edge form stays), and a star.
    monkeypatch.setenv("GUNROCK_CC_ROWFORM", rowform)
    if limit is not None:
        monkeypatch.setenv("GUNROCK_CC_ROW_LIMIT", limit)
    if rounds is not None:
        monkeypatch.setenv("GUNROCK_CC_NEIGHBOUR_ROUNDS", rounds)
    graphs = [o.rmat_seeded(10, 4 << 10), o.rmat_seeded(16, 8 << 16), o.rmat_seeded(18, 2 << 18)]
    n = 6000
    half = np.arange(n // 2 - 1, dtype=np.int32)
    rows = np.concatenate([half, half + n // 2, half + 1, half + 1 + n // 2])
    cols = np.concatenate([half + 1, half + 1 + n // 2, half, half + n // 2])
    hg = ga.HostGraph.from_coo(n, rows, cols)                              # two paths of 3000 vertices
    graphs.append(o.Csr(hg.nodes, np.array(hg.row_offsets), np.array(hg.col_indices)))
    rng = np.random.default_rng(5)
    a = rng.integers(0, 20000, 9000).astype(np.int32) // 4 * 4
    b = a + rng.integers(1, 4, 9000).astype(np.int32)
    hg = ga.HostGraph.from_coo(20000, np.concatenate([a, b]), np.concatenate([b, a]))   # components of at most 4 vertices
    graphs.append(o.Csr(hg.nodes, np.array(hg.row_offsets), np.array(hg.col_indices)))
    leaves = np.arange(1, 5000, dtype=np.int32)
    hg = ga.HostGraph.from_coo(5000, np.concatenate([np.zeros(4999, np.int32), leaves]), np.concatenate([leaves, np.zeros(4999, np.int32)]))
    graphs.append(o.Csr(hg.nodes, np.array(hg.row_offsets), np.array(hg.col_indices)))  # star: the hub's row is long
    # ... and a DIRECTED graph whose edges all point from the higher to the lower id: it passes the "every from < to edge has its mirror"
    # test vacuously, so the compact edge list applies, but the row form must not -- vertex 1000 + k of the giant component holds the
    # ONLY edge that ties vertex k (no out-edges, third in its row) to it
    k = np.arange(2, 1000, dtype=np.int32)
    rest = np.arange(1000, 3000, dtype=np.int32)
    rows = np.concatenate([[1], rest, rest, 1000 + k]).astype(np.int32)
    cols = np.concatenate([[0], np.zeros_like(rest), np.ones_like(rest), k]).astype(np.int32)
    hg = ga.HostGraph.from_coo(3000, rows, cols)
    graphs.append(o.Csr(hg.nodes, np.array(hg.row_offsets), np.array(hg.col_indices)))
    rng = np.random.default_rng(11)
    a, b = rng.integers(0, 40000, 150000), rng.integers(0, 40000, 150000)
    hg = ga.HostGraph.from_coo(40000, np.maximum(a, b).astype(np.int32), np.minimum(a, b).astype(np.int32))
    graphs.append(o.Csr(hg.nodes, np.array(hg.row_offsets), np.array(hg.col_indices)))
    for g in graphs:
        _check(g)
