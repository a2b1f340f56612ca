"""Host-side product code (csr.hpp / graphio/*.hpp through the C ABI) against the oracle and the goldens.
CPU only: no HIP compute call is made."""
import os

import numpy as np

import gunrockinst_amd as ga
from oracle import gr_oracle as o


def _same(g, ref, values=True):
    assert (g.nodes, g.edges) == (ref.nodes, ref.edges)
    assert np.array_equal(g.row_offsets, ref.row_offsets)
    assert np.array_equal(g.col_indices, ref.col_indices)
    if values:
        assert np.array_equal(g.edge_values, ref.edge_values)


def test_market_loader_matches_fixture_and_oracle(golden, golden_dir):
    f = golden["fixture7"]
    g = ga.HostGraph.from_market(os.path.join(golden_dir, f["mtx"]))
    assert g.row_offsets.tolist() == f["row_offsets"] and g.col_indices.tolist() == f["col_indices"]
    for name, und, rev in [("test_bc.mtx", False, False), ("test_bc.mtx", False, True), ("test_cc.mtx", True, False),
                           ("chesapeake.mtx", True, False), ("bips98_606.mtx", True, False),
                           ("bips98_606.mtx", False, False), ("test_pr.mtx", False, True)]:
        path = os.path.join(golden_dir, name)
        _same(ga.HostGraph.from_market(path, und, rev), o.build_market(path, und, rev))


def test_market_quirks(tmp_path):
    # real-valued weights truncate at the first non-digit, missing -> 1, blank lines are skipped,
    # duplicate keeps the FIRST value after the stable sort, self loop dropped
    p = tmp_path / "q.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n% c\n4 4 6\n2 1 3.9\n\n   3 1\n2 1 7\n1 1 5\n4 3 -2.5e3\n1 4 8\n")
    g = ga.HostGraph.from_market(str(p))
    ref = o.build_market(str(p))
    _same(g, ref)
    assert g.row_offsets.tolist() == [0, 2, 2, 3, 4]
    assert g.col_indices.tolist() == [1, 2, 3, 0] and g.edge_values.tolist() == [3, 1, -2, 8]
    gu = ga.HostGraph.from_market(str(p), undirected=True)
    _same(gu, o.build_market(str(p), undirected=True))


def test_market_errors(tmp_path):
    bad = tmp_path / "bad.mtx"
    bad.write_text("3 4 1\n1 2\n")
    import pytest
    with pytest.raises(RuntimeError):
        ga.HostGraph.from_market(str(bad))
    with pytest.raises(RuntimeError):
        ga.HostGraph.from_market(str(tmp_path / "missing.mtx"))
    short = tmp_path / "short.mtx"
    short.write_text("3 3 2\n1 2\n")
    with pytest.raises(RuntimeError):
        ga.HostGraph.from_market(str(short))


def test_bips_goldens(golden, golden_dir):
    f = golden["bips98_606"]
    g = ga.HostGraph.from_market(os.path.join(golden_dir, f["mtx"]), undirected=True)
    assert (g.nodes, g.edges) == (f["nodes"], f["edges"])
    assert g.highest_degree_node() == (f["max_degree_node"], f["max_degree"])
    assert g.average_degree() == f["avg_degree"]


def test_rmat_libc_matches_oracle_stream(golden):
    import ctypes
    f = golden["rmat_libc"]
    ctypes.CDLL(None).srand(1)
    g = ga.HostGraph.rmat_libc(f["nodes"], f["edges_in"], undirected=False)
    assert g.edges == f["edges"] and g.row_offsets[1:5].tolist() == f["row_offsets_1_4"]
    ctypes.CDLL(None).srand(1)
    gu = ga.HostGraph.rmat_libc(256, 2048, undirected=True)
    _same(gu, o.rmat_reference(256, 2048, undirected=True, srand=1))


def test_rmat_seeded_matches_oracle():
    for scale, ef, und in [(8, 8, True), (12, 4, False), (14, 8, True)]:
        g = ga.HostGraph.rmat_seeded(scale, ef << scale, undirected=und)
        _same(g, o.rmat_seeded(scale, ef << scale, undirected=und))


def test_from_coo_and_csr_roundtrip():
    rng = np.random.default_rng(7)
    rows = rng.integers(0, 50, 400, dtype=np.int32)
    cols = rng.integers(0, 50, 400, dtype=np.int32)
    vals = rng.integers(1, 99, 400, dtype=np.int32)
    g = ga.HostGraph.from_coo(50, rows, cols, vals)
    # independent numpy restatement of FromCoo: stable sort, drop self loops, keep first of each repeat
    order = np.lexsort((cols, rows))            # stable
    r, c, v = rows[order], cols[order], vals[order]
    keep = (r != c) & np.concatenate(([True], (r[1:] != r[:-1]) | (c[1:] != c[:-1])))
    assert np.array_equal(g.col_indices, c[keep]) and np.array_equal(g.edge_values, v[keep])
    assert np.array_equal(g.row_offsets, np.concatenate(([0], np.cumsum(np.bincount(r[keep], minlength=50)))))
    g2 = ga.HostGraph.from_csr(g.nodes, g.row_offsets, g.col_indices, g.edge_values)
    assert np.array_equal(g2.col_indices, g.col_indices) and g2.highest_degree_node() == g.highest_degree_node()


def test_empty_and_edgeless_graphs():
    g = ga.HostGraph.from_coo(5, np.empty(0, np.int32), np.empty(0, np.int32))
    assert g.edges == 0 and g.row_offsets.tolist() == [0] * 6
    g = ga.HostGraph.from_coo(3, np.array([1, 2], np.int32), np.array([1, 2], np.int32))   # only self loops
    assert g.edges == 0 and g.highest_degree_node() == (0, 0)
