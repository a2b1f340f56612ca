"""PageRank and TopK on the GPU against the CPU oracle.

PageRank sums floats (the reference's per-edge atomicAdd, and here the pulled reduction whose pieces meet in an atomic when a
neighbour list straddles waves): results are compared with the oracle's double-precision run of the same schedule within the
reference's own float tolerance (test_utils.cuh:360-405: 5 % relative, 0.05 absolute below 0.01) -- in practice 1e-4 relative
is met and asserted here.  PARITY UNPINNED: the reference's ctest answer for PageRank (CMakeLists.txt:231-233) is not
reproduced by the code in its tree (tests/test_oracle.py::test_pagerank_ctest_answer_is_stale).  TopK is integer work and IS
pinned by its ctest answer (CMakeLists.txt:235-237): bit-exact."""
import numpy as np
import pytest

import gunrockinst_amd as ga
from oracle import gr_oracle as o

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-6


def _transpose(g):
    src_of = np.repeat(np.arange(g.nodes, dtype=np.int32), np.diff(g.row_offsets))
    inv = ga.HostGraph.from_coo(g.nodes, g.col_indices, src_of)      # no self loops / duplicates in these graphs
    return np.array(inv.row_offsets), np.array(inv.col_indices)


def _check_ranks(g, ids, ranks, src, delta, thr, max_iter, top=None):
    ref, deg, iters = o.pagerank(g, src, delta, thr, max_iter)
    by_vertex = np.zeros(g.nodes, dtype=np.float64)
    by_vertex[ids] = ranks
    if top is None:
        assert sorted(ids.tolist()) == list(range(g.nodes))
        assert np.allclose(by_vertex, ref, rtol=RTOL, atol=ATOL)
    else:
        assert np.allclose(by_vertex[ids], ref[ids], rtol=RTOL, atol=ATOL)
        # the reported vertices are the best ones (up to ties inside the tolerance)
        kth = np.sort(ref)[::-1][len(ids) - 1]
        assert np.all(ref[ids] >= kth - ATOL - RTOL * abs(kth))
    assert np.all(np.diff(ranks.astype(np.float64)) <= 0)            # descending
    return ref, deg, iters


def test_fixture7_c_abi(golden, capfd):
    f = golden["fixture7"]
    g = o.Csr(7, f["row_offsets"], f["col_indices"])
    # shared_lib_tests/test_pr.c: delta .85, error .01, 20 iterations, source 0, top 10 of 7 vertices
    ids, ranks = ga.gunrock_pr(7, g.row_offsets, g.col_indices, src=0, delta=0.85, error=0.01, max_iter=20, top_nodes=10)
    assert ids.shape[0] == 7
    ref, deg, iters = _check_ranks(g, ids, ranks, 0, 0.85, 0.01, 20)
    assert ids[0] == 2 and deg.tolist() == [2, 3, 1, -1, 1, -1, -1] and iters == 10      # vertices 3, 5, 6 are peeled off
    assert "[GPU PageRank] finished." in capfd.readouterr().out
    ids3, ranks3 = ga.gunrock_pr(7, g.row_offsets, g.col_indices, src=-1, top_nodes=3)
    _check_ranks(g, ids3, ranks3, -1, 0.85, 0.01, 20, top=3)


def test_topk_known_answer(golden):
    f = golden["fixture7"]
    g = o.Csr(7, f["row_offsets"], f["col_indices"])
    col_offsets = [0, 1, 2, 5, 7, 9, 12, 15]                          # shared_lib_tests/test_topk.c:30-31
    row_indices = [1, 0, 0, 1, 4, 0, 2, 1, 2, 2, 3, 4, 3, 4, 5]
    ids, ind, outd = ga.gunrock_topk(7, g.row_offsets, g.col_indices, col_offsets, row_indices, 3)
    # ctest: "Node ID.*2.*: in_degrees.*3.*: out_degrees.*3" (CMakeLists.txt:235-237)
    assert (int(ids[0]), int(ind[0]), int(outd[0])) == (2, 3, 3)
    rid, rin, rout = o.topk(g, 3, col_offsets)
    assert ids.tolist() == rid.tolist() and ind.tolist() == rin.tolist() and outd.tolist() == rout.tolist()
    for scale in (10, 14):
        g = o.rmat_seeded(scale, 8 << scale, undirected=False)
        co, ri = _transpose(g)
        for k in (1, 17, g.nodes):
            got = ga.gunrock_topk(g.nodes, g.row_offsets, g.col_indices, co, ri, k)
            want = o.topk(g, k, co)
            assert all(a.tolist() == b.tolist() for a, b in zip(got, want))


@pytest.mark.parametrize("scale,undirected", [(10, True), (14, True), (16, True), (12, False), (15, False)])
def test_rmat_parity(scale, undirected):
    g = o.rmat_seeded(scale, 8 << scale, undirected=undirected)
    src, _ = o.highest_degree_node(g)
    p = ga.PrProblem().init(g.nodes, g.row_offsets, g.col_indices)
    p.set_inverse_graph(build=not undirected)                         # symmetric graph: its CSR is its own inverse
    for s, delta, thr, iters in [(-1, 0.85, 0.01, 20), (src, 0.85, 1e-4, 50), (-1, 0.5, 0.0, 5), (src, 0.85, 0.01, 1)]:
        p.reset(s, delta, thr)
        p.enact(iters)
        ids, ranks = p.extract()
        ref, deg, ref_iters = _check_ranks(g, ids, ranks, s, delta, thr, iters)
        st = p.stats()
        assert st["surviving_nodes"] == int((deg > 0).sum())
        assert abs(st["iterations"] - ref_iters) <= 1                 # a vertex whose move is within rounding of the threshold
    p.close()


@pytest.mark.parametrize("shape", ["downward", "mixed"])
def test_c_abi_does_not_take_one_way_edges_for_symmetric(shape):
    # with unmirrored edges that all run from a higher to a lower id the old symmetry test said "symmetric" and
    # gunrock_pr_func pulled ranks over the OUT-lists.  Expected: the transpose is built, ranks = oracle.
    g0 = o.rmat_seeded(13, 8 << 13, undirected=True)
    froms = np.repeat(np.arange(g0.nodes, dtype=np.int64), np.diff(g0.row_offsets))
    tos = g0.col_indices.astype(np.int64)
    keep = (froms > tos) if shape == "downward" else ((froms > tos) | ((froms < 2048) & (tos < 2048)))
    ro = np.zeros(g0.nodes + 1, dtype=np.int32)
    np.cumsum(np.bincount(froms[keep], minlength=g0.nodes), out=ro[1:])
    g = o.Csr(g0.nodes, ro, g0.col_indices[keep])
    ids, ranks = ga.gunrock_pr(g.nodes, g.row_offsets, g.col_indices, src=-1, delta=0.85, error=0.0, max_iter=10)
    _check_ranks(g, ids, ranks, -1, 0.85, 0.0, 10)


def test_directed_graph_needs_its_inverse():
    # a directed chain with a sink: peeling removes everything but a 2-cycle; CSC given explicitly on the device
    import torch
    rows = np.array([0, 1, 2, 3, 3, 4], dtype=np.int32)
    cols = np.array([1, 2, 3, 2, 4, 5], dtype=np.int32)
    hg = ga.HostGraph.from_coo(6, rows, cols)
    g = o.Csr(6, np.array(hg.row_offsets), np.array(hg.col_indices))
    iro, ici = _transpose(g)
    d_iro = torch.tensor(iro, dtype=torch.int32, device="cuda")
    d_ici = torch.tensor(ici, dtype=torch.int32, device="cuda")
    p = ga.PrProblem().init(g.nodes, g.row_offsets, g.col_indices)
    p.set_inverse_graph(d_iro.data_ptr(), d_ici.data_ptr())
    p.reset(-1, 0.85, 0.0)
    p.enact(30)
    ids, ranks = p.extract()
    ref, deg, _ = _check_ranks(g, ids, ranks, -1, 0.85, 0.0, 30)
    assert deg.tolist() == [1, 1, 1, 1, -1, -1]                        # 5 is a sink; 4 pointed only at it
    p.close()


def test_edge_cases():
    g = o.Csr(3, [0, 0, 0, 0], [])                                     # no edges at all: nothing survives; the reference still runs one
    ids, ranks = ga.gunrock_pr(3, g.row_offsets, np.zeros(0, np.int32), src=-1)   # pass over its empty queue, which zeroes every rank
    assert sorted(ids.tolist()) == [0, 1, 2] and np.allclose(ranks, 0.0)            # (pr_enactor.cuh:341,478-482)
    ref, _, iters = o.pagerank(g, -1, 0.85, 0.01, 20)
    assert np.allclose(ref, 0.0) and iters == 1
    # a DAG: peeling removes every vertex round by round
    rows = np.array([0, 0, 1, 2, 2, 3], dtype=np.int32)
    cols = np.array([1, 2, 3, 3, 4, 4], dtype=np.int32)
    hg = ga.HostGraph.from_coo(5, rows, cols)
    g = o.Csr(5, np.array(hg.row_offsets), np.array(hg.col_indices))
    p = ga.PrProblem().init(g.nodes, g.row_offsets, g.col_indices)
    p.set_inverse_graph(build=True)
    p.reset(-1, 0.85, 0.01)
    p.enact(20)
    ids, ranks = p.extract()
    st = p.stats()
    p.close()
    assert np.allclose(ranks, 0.0) and st["iterations"] == 1 and st["surviving_nodes"] == 0
    hub = 5000                                                          # one list far longer than a wave's share, both directions
    rows = np.concatenate([np.zeros(hub - 1, np.int32), np.arange(1, hub, dtype=np.int32)])
    cols = np.concatenate([np.arange(1, hub, dtype=np.int32), np.zeros(hub - 1, np.int32)])
    hg = ga.HostGraph.from_coo(hub, rows, cols)
    g = o.Csr(hub, np.array(hg.row_offsets), np.array(hg.col_indices))
    ids, ranks = ga.gunrock_pr(hub, g.row_offsets, g.col_indices, src=-1, max_iter=10, error=0.0)
    _check_ranks(g, ids, ranks, -1, 0.85, 0.0, 10)
