"""The compacting filter operator by itself (reference filter::Kernel, gunrock/oprtr/filter/kernel.cuh:211-383, with the BFS
functor's CondFilter "valid vertex id", bfs_functor.cuh:100-105), through the C ABI (grx_filter_queue).

Expected values are computed here with numpy: the kept ids are the input without its -1 entries (as a multiset: the operator
does not promise an order), and with row offsets every output entry carries its row start and the exclusive prefix of the
degrees in OUTPUT order -- what the load-balanced advance consumes (reference: GetEdgeCounts + scan, advance/kernel.cuh:300-368).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ga = pytest.importorskip("gunrockinst_amd")
from oracle import gr_oracle as o  # noqa: E402


def _check_frontier(ids, ro, got):
    v, rs, sc, edges = got
    deg = np.diff(ro)
    want = ids[ids >= 0]
    want = want[deg[want] > 0]                                  # vertices without out-edges never enter an advance frontier
    assert np.array_equal(np.sort(v), np.sort(want))
    assert np.array_equal(rs, ro[v])
    prefix = np.concatenate([[0], np.cumsum(deg[v])])
    assert np.array_equal(sc, prefix[:-1]) and edges == int(prefix[-1])


@pytest.mark.parametrize("n,holes", [(0, 0.0), (1, 0.0), (1, 1.0), (63, 0.5), (1024, 0.0), (1025, 1.0), (5000, 0.3), (300_000, 0.9),
                                     (2_000_000, 0.5)])
def test_ids_only(n, holes):
    rng = np.random.default_rng(n + 7)
    ids = rng.integers(0, 1 << 20, n, dtype=np.int32)
    ids[rng.random(n) < holes] = -1
    (v,) = ga.filter_queue(ids)
    assert np.array_equal(np.sort(v), np.sort(ids[ids >= 0]))   # duplicates are kept: the functor culls nothing but -1


@pytest.mark.parametrize("scale,holes,grid", [(10, 0.0, 0), (12, 0.5, 0), (16, 0.2, 0), (16, 0.97, 3), (18, 0.5, 1)])
def test_vertex_frontier_with_degree_prefix(scale, holes, grid):
    g = o.rmat_seeded(scale, 8 << scale)
    ro = np.asarray(g.row_offsets, dtype=np.int32)
    rng = np.random.default_rng(scale)
    ids = rng.permutation(g.nodes).astype(np.int32)             # every vertex once: hubs, leaves and isolated vertices
    ids[rng.random(g.nodes) < holes] = -1
    _check_frontier(ids, ro, ga.filter_queue(ids, ro, max_grid_size=grid))


def test_overflow_is_reported():
    # reference: "Frontier queue overflow.  Please increase queue-sizing factor." (filter/cta.cuh:526-529)
    ids = np.arange(10_000, dtype=np.int32)
    with pytest.raises(RuntimeError):
        ga.filter_queue(ids, capacity=100)
