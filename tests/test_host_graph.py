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


def _both(path, **kw):
    """(library graph | None, oracle graph | None): None = that side refused the file."""
    try:
        g = ga.HostGraph.from_market(path, **kw)
    except RuntimeError:
        g = None
    try:
        r = o.build_market(path, kw.get("undirected", False), kw.get("reversed_", False))
    except Exception:
        r = None
    return g, r


def test_market_line_cutting_matches_the_fscanf_loop(tmp_path):
    # The library scans the file from memory by hand; the oracle keeps the reference's fscanf("%1023[^\n]\n") + sscanf pair
    # (market.cuh:81-83,139-148).  Both must cut and read these inputs alike: CRLF, tabs, signs, trailing text after the
    # numbers, comment lines of 1023 / 1024 / 3000 characters (a longer line continues as the NEXT line), white space before
    # the first line, a newline as the very first byte (ends the parse), a value that overflows 64 bits.
    body = "3 1 4\n1 2 7\n2 3\n"
    cases = {
        "crlf": "%%MatrixMarket\r\n3 3 3\r\n" + body.replace("\n", "\r\n"),
        "tabs": "%c\n3\t3\t3\n3\t1\t4\n1 \t 2 \t+7 trailing words\n  2   3   \n\n\n",
        "signs": "3 3 3\n3 1 -4\n1 2 +7\n2 3 0009\n",
        "long1023": "%" * 1023 + "\n3 3 3\n" + body,
        "long1024": "%" * 1024 + "\n3 3 3\n" + body,
        "long3000": "%" * 3000 + "\n3 3 3\n" + body,
        "long_split_becomes_data": "%" + "x" * 1022 + "3 3 3\n" + body,        # the 1024th character starts a new "line"
        "leading_space": "   \n3 3 3\n" + body,
        "newline_first": "\n3 3 3\n" + body,
        "overflow": "3 3 3\n3 1 99999999999999999999999\n1 2 7\n2 3\n",
        "float_first_column": "3 3 3\n3.5 1 4\n1 2 7\n2 3\n",
        "no_trailing_newline": "3 3 3\n3 1 4\n1 2 7\n2 3",
        "too_many": "3 3 2\n3 1 4\n1 2 7\n2 3\n",
    }
    for name, text in cases.items():
        p = tmp_path / (name + ".mtx")
        p.write_bytes(text.encode())
        for kw in ({}, {"undirected": True}, {"reversed_": True}):
            g, r = _both(str(p), **kw)
            assert (g is None) == (r is None), (name, kw)
            if g is not None:
                _same(g, r)


def test_market_binary_cache(tmp_path):
    # reference rule: "<dir>/.<name>_{undirected,reversed,nonreversed}_csr" after the first parse, preferred afterwards
    # (market.cuh:296-339) -- here binary and stamped with the source's size and mtime, so a changed input is never shadowed
    p = tmp_path / "g.mtx"
    p.write_text("4 4 4\n2 1 3\n3 1 5\n4 3 2\n1 4 8\n")
    first = ga.HostGraph.from_market(str(p), cache=True)
    assert not first.cache_hit and (tmp_path / ".g.mtx_nonreversed_csr.bin").exists()
    again = ga.HostGraph.from_market(str(p), cache=True)
    assert again.cache_hit
    _same(again, first)
    _same(again, o.build_market(str(p)))
    und = ga.HostGraph.from_market(str(p), undirected=True, cache=True)      # another mode: its own cache file
    assert not und.cache_hit and (tmp_path / ".g.mtx_undirected_csr.bin").exists()
    rev = ga.HostGraph.from_market(str(p), reversed_=True, cache=True)
    assert not rev.cache_hit and (tmp_path / ".g.mtx_reversed_csr.bin").exists()
    _same(ga.HostGraph.from_market(str(p), undirected=True, cache=True), o.build_market(str(p), True))
    # the input changes (same name): the stale cache must lose
    p.write_text("4 4 3\n2 1 3\n3 1 5\n4 3 2\n")
    os.utime(str(p), ns=(1, 1))
    changed = ga.HostGraph.from_market(str(p), cache=True)
    assert not changed.cache_hit and changed.edges == 3
    assert ga.HostGraph.from_market(str(p), cache=True).cache_hit
    # a truncated or foreign cache file is not a hit either
    c = tmp_path / ".g.mtx_nonreversed_csr.bin"
    c.write_bytes(c.read_bytes()[:-4])
    g = ga.HostGraph.from_market(str(p), cache=True)
    assert not g.cache_hit and g.edges == 3
    c.write_bytes(b"not a cache")
    assert not ga.HostGraph.from_market(str(p), cache=True).cache_hit
    # the plain loader never looks at caches
    p2 = tmp_path / "h.mtx"
    p2.write_text("3 3 1\n2 1\n")
    ga.HostGraph.from_market(str(p2))
    assert not (tmp_path / ".h.mtx_nonreversed_csr.bin").exists()
