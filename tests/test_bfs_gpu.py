"""BFS parity on the GPU: the HIP engine (through the C ABI) against the CPU oracle, bit-exact on labels;
predecessors are checked as "valid parent" (reference tests/sssp/test_sssp.cu:488-489: paths are not unique)."""
import os

import numpy as np
import pytest

import gunrockinst_amd as ga
from oracle import gr_oracle as o

pytestmark = pytest.mark.gpu

MODES = [(False, False), (True, False), (False, True), (True, True)]


def _run(g, src, mark_pred, idempotence, instrument=False, queue_sizing=1.0, mode=0, alpha=0.0, beta=0.0):
    p = ga.BfsProblem(mark_pred, idempotence, instrument).init(g.nodes, g.row_offsets, g.col_indices)
    if mode == 2:
        p.set_inverse_graph(alpha=alpha, beta=beta)      # symmetric graphs only
    p.reset(src, queue_sizing)
    ms = p.enact(src, traversal_mode=mode)
    labels, preds = p.extract()
    st = p.stats()
    p.close()
    return labels, preds, st, ms


def _check(g, src, labels, preds, st):
    ref, _, depth = o.bfs(g, src)
    assert np.array_equal(labels, ref)
    if preds is not None:
        assert o.check_bfs_preds(g, src, labels, preds) == 0
    # enactor depth = completed iterations; CPU prints max_label + 1
    assert st["search_depth"] in (depth - 1, depth)


@pytest.mark.parametrize("mark_pred,idempotence", MODES)
def test_fixture7_all_modes(golden, mark_pred, idempotence):
    f = golden["fixture7"]
    g = o.Csr(7, f["row_offsets"], f["col_indices"])
    for src in range(7):
        labels, preds, st, _ = _run(g, src, mark_pred, idempotence)
        _check(g, src, labels, preds, st)
    labels, _, _, _ = _run(g, 0, mark_pred, idempotence)
    assert labels.tolist() == f["bfs_src0_labels"]


def test_c_abi_entry_point_known_answer(golden, capfd):
    f = golden["fixture7"]
    ro = np.array(f["row_offsets"], np.int32)
    ci = np.array(f["col_indices"], np.int32)
    # reference ctest regex: Node_ID [2] : Label [1] (CMakeLists.txt:215-217) holds for src 0 and src 1
    labels = ga.gunrock_bfs(7, ro, ci, src=0)
    assert labels.tolist() == f["bfs_src0_labels"] and labels[2] == 1
    labels = ga.gunrock_bfs(7, ro, ci, src_mode=ga.SRC_LARGEST_DEGREE, idempotence=True)
    assert labels.tolist() == f["bfs_src0_labels"]          # first max-degree vertex is 0
    out = capfd.readouterr().out
    assert "[GPU Breadth-first search] finished." in out and "MiEdges/s" in out


@pytest.mark.parametrize("mark_pred,idempotence", MODES)
def test_bips98_606_simple_example_path(golden, golden_dir, mark_pred, idempotence):
    f = golden["bips98_606"]
    g = o.build_market(os.path.join(golden_dir, f["mtx"]), undirected=True)
    for src, key in [(0, "bfs_src0"), (566, "bfs_src566")]:
        labels, preds, st, _ = _run(g, src, mark_pred, idempotence)
        _check(g, src, labels, preds, st)
        assert labels[:10].tolist() == f[key]["labels_head"]
    nv, ev = o.bfs_stats(g, _run(g, 0, mark_pred, idempotence)[0])
    assert (nv, ev) == (f["bfs_src0"]["nodes_visited"], f["bfs_src0"]["edges_visited"])


def test_edge_cases():
    # isolated source, single vertex, source with only a self-loop-free empty row, chain, star
    g = o.Csr(4, [0, 0, 1, 1, 1], [0])
    for src in range(4):
        labels, preds, st, _ = _run(g, src, True, False)
        _check(g, src, labels, preds, st)
    g = o.Csr(1, [0, 0], [])
    labels, _, st, _ = _run(g, 0, False, True)
    assert labels.tolist() == [0] and st["search_depth"] == 0
    n = 5000                                             # long path: many tiny levels
    ro = np.minimum(np.arange(n + 1), n - 1).astype(np.int32)
    ci = np.arange(1, n, dtype=np.int32)
    g = o.Csr(n, ro, ci)
    labels, preds, st, _ = _run(g, 0, True, True)
    _check(g, 0, labels, preds, st)
    assert labels[-1] == n - 1
    hub = 70000                                          # one row far larger than a tile, split over workgroups
    ro = np.concatenate(([0], np.full(hub, hub - 1))).astype(np.int32)
    ro[-1] = hub - 1
    g = o.Csr(hub, np.concatenate(([0], np.full(hub, hub - 1, dtype=np.int32))), np.arange(1, hub, dtype=np.int32))
    labels, preds, st, _ = _run(g, 0, True, False)
    _check(g, 0, labels, preds, st)


@pytest.mark.parametrize("scale,ef", [(10, 8), (14, 8), (16, 16), (18, 8)])
def test_rmat_seeded_parity(scale, ef):
    g = o.rmat_seeded(scale, ef << scale)
    src, _ = o.highest_degree_node(g)
    rng = np.random.default_rng(scale)
    deg = np.diff(g.row_offsets)
    others = rng.choice(np.nonzero(deg > 0)[0], 3)
    for s in [src] + others.tolist():
        for mark_pred, idempotence in [(False, True), (True, False)]:
            labels, preds, st, _ = _run(g, int(s), mark_pred, idempotence)
            _check(g, int(s), labels, preds, st)


def test_degree_one_rows_fill_a_tile_exactly():
    # perfect matching chained: every frontier vertex has degree 1 -> exercises the cursor hand-over rule
    n = 3 * 2048 + 5
    ro = np.arange(n + 1, dtype=np.int32)
    ro[-1] = n - 1
    ro = np.minimum(ro, n - 1)
    # layered graph: layer 0 = {0}, vertex 0 -> 1..4100 (wide), each of those -> one private vertex
    width = 4100
    rows = [np.zeros(width, np.int32), np.arange(1, width + 1, dtype=np.int32)]
    cols = [np.arange(1, width + 1, dtype=np.int32), np.arange(width + 1, 2 * width + 1, dtype=np.int32)]
    g0 = ga.HostGraph.from_coo(2 * width + 1, np.concatenate(rows), np.concatenate(cols))
    g = o.Csr(g0.nodes, g0.row_offsets.copy(), g0.col_indices.copy())
    labels, preds, st, _ = _run(g, 0, True, True)
    _check(g, 0, labels, preds, st)
    assert (labels[width + 1:] == 2).all()


def test_instrumented_run_reports_kernel_time():
    g = o.rmat_seeded(14, 8 << 14)
    src, _ = o.highest_degree_node(g)
    labels, _, st, ms = _run(g, src, False, True, instrument=True)
    assert np.array_equal(labels, o.bfs(g, src)[0])
    # small consecutive levels share one launch (multi-level tail kernel), so launches <= levels
    assert 0 < st["kernel_launches"] <= st["search_depth"] and 0 < st["kernel_ms"] <= ms


def test_reset_and_rerun_same_problem():
    g = o.rmat_seeded(12, 8 << 12)
    p = ga.BfsProblem(True, False).init(g.nodes, g.row_offsets, g.col_indices)
    deg = np.diff(g.row_offsets)
    for src in np.nonzero(deg > 0)[0][:5].tolist():
        p.reset(src)
        p.enact(src)
        labels, preds = p.extract()
        assert np.array_equal(labels, o.bfs(g, src)[0])
        assert o.check_bfs_preds(g, src, labels, preds) == 0
    p.close()


# ---------------- direction-optimizing traversal (traversal_mode = 2) ----------------
@pytest.mark.parametrize("mark_pred,idempotence", MODES)
def test_dobfs_bips_all_modes(golden, golden_dir, mark_pred, idempotence):
    g = o.build_market(os.path.join(golden_dir, "bips98_606.mtx"), undirected=True)
    for src in (0, 566):
        for alpha, beta in [(0.0, 0.0), (1e9, 1.0), (1e9, 1e9)]:   # default, always bottom-up, flip-flop every level
            labels, preds, st, _ = _run(g, src, mark_pred, idempotence, mode=2, alpha=alpha, beta=beta)
            _check(g, src, labels, preds, st)


@pytest.mark.parametrize("scale,ef", [(10, 8), (14, 8), (16, 16), (18, 8)])
def test_dobfs_rmat_parity(scale, ef):
    g = o.rmat_seeded(scale, ef << scale)
    src, _ = o.highest_degree_node(g)
    rng = np.random.default_rng(scale + 100)
    deg = np.diff(g.row_offsets)
    others = rng.choice(np.nonzero(deg > 0)[0], 3)
    for s in [src] + others.tolist():
        for alpha, beta in [(0.0, 0.0), (1e9, 1.0)]:
            labels, preds, st, _ = _run(g, int(s), True, True, mode=2, alpha=alpha, beta=beta, instrument=True)
            _check(g, int(s), labels, preds, st)


def test_dobfs_long_rows_take_the_wave_sweep():
    # complete bipartite-ish: 300 left vertices each adjacent to 3000 right vertices; source on the right side.
    # Bottom-up from level 1: left vertices find the source only deep in their list (source id is the LAST right id).
    L, R = 300, 3000
    rows = np.repeat(np.arange(L, dtype=np.int32), R)
    cols = np.tile(np.arange(L, L + R, dtype=np.int32), L)
    g0 = ga.HostGraph.from_coo(L + R, np.concatenate([rows, cols]), np.concatenate([cols, rows]))
    g = o.Csr(g0.nodes, g0.row_offsets.copy(), g0.col_indices.copy())
    src = L + R - 1
    labels, preds, st, _ = _run(g, src, True, False, mode=2, alpha=1e9, beta=1.0)
    _check(g, src, labels, preds, st)


def test_dobfs_edge_cases():
    g = o.Csr(1, [0, 0], [])
    labels, _, st, _ = _run(g, 0, False, True, mode=2, alpha=1e9, beta=1.0)
    assert labels.tolist() == [0]
    n = 300                                              # path graph, symmetric, always bottom-up
    rows = np.concatenate([np.arange(n - 1), np.arange(1, n)]).astype(np.int32)
    cols = np.concatenate([np.arange(1, n), np.arange(n - 1)]).astype(np.int32)
    g0 = ga.HostGraph.from_coo(n, rows, cols)
    g = o.Csr(n, g0.row_offsets.copy(), g0.col_indices.copy())
    labels, preds, st, _ = _run(g, 0, True, True, mode=2, alpha=1e9, beta=1.0)
    _check(g, 0, labels, preds, st)
    labels, preds, st, _ = _run(g, n // 2, True, True, mode=2, alpha=1e9, beta=1e9)
    _check(g, n // 2, labels, preds, st)


def test_dobfs_directed_graph_with_explicit_inverse():
    # directed R-MAT: the in-neighbour CSR is a different graph; sources without in-edges and vertices without
    # in-edges (never discoverable, pre-marked in the visited bitmap) must behave
    g = o.rmat_seeded(14, 8 << 14, undirected=False)
    src_of = np.repeat(np.arange(g.nodes, dtype=np.int32), np.diff(g.row_offsets))
    inv0 = ga.HostGraph.from_coo(g.nodes, g.col_indices, src_of)
    import torch
    iro = torch.tensor(inv0.row_offsets, dtype=torch.int32, device="cuda")
    ici = torch.tensor(inv0.col_indices, dtype=torch.int32, device="cuda")
    outdeg, indeg = np.diff(g.row_offsets), np.diff(inv0.row_offsets)
    picks = [int(np.argmax(outdeg)), int(np.nonzero((indeg == 0) & (outdeg > 0))[0][0]), int(np.nonzero(outdeg == 0)[0][0])]
    for src in picks:
        for alpha, beta in [(0.0, 0.0), (1e9, 1.0), (1e9, 1e9)]:
            p = ga.BfsProblem(True, True).init(g.nodes, g.row_offsets, g.col_indices)
            p.set_inverse_graph(iro.data_ptr(), ici.data_ptr(), alpha, beta)
            p.reset(src)
            p.enact(src, traversal_mode=2)
            labels, preds = p.extract()
            st = p.stats()
            p.close()
            _check(g, src, labels, preds, st)


@pytest.mark.parametrize("lite_factor,tail_limit", [(1e9, 0), (1e9, 32768), (0.0, 0), (8.0, 100)])
def test_enactor_schedules_do_not_change_results(lite_factor, tail_limit):
    # force / forbid the count-only top-down level and the multi-level tail kernel: labels and parents must not move
    for scale, ef in [(12, 8), (16, 8), (18, 16)]:
        g = o.rmat_seeded(scale, ef << scale)
        deg = np.diff(g.row_offsets)
        srcs = [o.highest_degree_node(g)[0]] + np.nonzero((deg > 0) & (deg < 4))[0][:2].tolist()
        for mark_pred in (False, True):
            p = ga.BfsProblem(mark_pred, True).init(g.nodes, g.row_offsets, g.col_indices)
            p.set_inverse_graph()
            p.set_tuning(lite_factor=lite_factor, tail_edge_limit=tail_limit)
            for src in srcs:
                for mode in (0, 2):
                    p.reset(int(src))
                    p.enact(int(src), traversal_mode=mode)
                    labels, preds = p.extract()
                    _check(g, int(src), labels, preds, p.stats())
            p.close()


@pytest.mark.parametrize("twc_limit", [64, 5000, 65536, 1 << 22])
def test_twc_levels_kernel_parity(twc_limit):
    # thread / wave / workgroup tiers in one resident workgroup with the frontier in LDS (oprtr/advance/twc.hpp; the reference's
    # TWC advance, edge_map_forward/cta.cuh:224-545, picked by traversal_mode 1 / average degree <= 8, test_bfs.cu:563-566).
    # Grids: whole searches inside the kernel.  R-MAT: hub rows go through the wave and workgroup tiers, a level with hundreds
    # of medium rows overflows the deferred lists, the hub's level outgrows the LDS queue (spill + hand-over through the
    # frontier writer), small limits make it hand back early; labels / parents must equal the oracle's in every case.
    from gunrockinst_amd import devgraph
    graphs = []
    for side, frac in [(200, 0.0), (256, 0.02)]:
        ro, ci = devgraph.grid_csr_device(side, frac)
        h_ro, h_ci = devgraph.to_host_csr(ro, ci)
        graphs.append((o.Csr(side * side, h_ro, h_ci), [0, side * side // 2 + side // 2]))
    for scale in (12, 16, 18):
        g = o.rmat_seeded(scale, 8 << scale)
        deg = np.diff(g.row_offsets)
        graphs.append((g, [o.highest_degree_node(g)[0], int(np.nonzero(deg == 1)[0][0]), int(np.nonzero(deg == 0)[0][0])]))
    star = 20000                                              # one list far beyond the wave tier, leaves of degree 1
    hg = ga.HostGraph.from_coo(star, np.zeros(star - 1, np.int32), np.arange(1, star, dtype=np.int32))
    ro_s, ci_s = np.array(hg.row_offsets), np.array(hg.col_indices)
    graphs.append((o.Csr(star, ro_s, ci_s), [0]))
    for g, srcs in graphs:
        for mark_pred in (False, True):
            p = ga.BfsProblem(mark_pred, True).init(g.nodes, g.row_offsets, g.col_indices)
            p.set_twc_limit(twc_limit)
            for src in srcs:
                p.reset(int(src))
                p.enact(int(src), traversal_mode=1)
                labels, preds = p.extract()
                _check(g, int(src), labels, preds, p.stats())
            p.close()


@pytest.mark.parametrize("persistent_limit,tail_limit", [(1 << 20, 8192), (1 << 20, 0), (1 << 14, 256), (0, 8192)])
def test_persistent_levels_kernel_parity(persistent_limit, tail_limit):
    # mid-size levels inside the persistent multi-workgroup kernel (grid barrier between levels) versus launch-per-level:
    # road-like grids (long diameter, ~4 neighbours: the reference's traversal_mode 1 case, test_bfs.cu:563-566) and R-MAT
    import torch
    from gunrockinst_amd import devgraph
    graphs = []
    for side, frac in [(256, 0.0), (300, 0.01)]:
        ro, ci = devgraph.grid_csr_device(side, frac)
        h_ro, h_ci = devgraph.to_host_csr(ro, ci)
        graphs.append((o.Csr(side * side, h_ro, h_ci), [0, side * side // 2 + side // 2]))
    g = o.rmat_seeded(16, 8 << 16)
    graphs.append((g, [o.highest_degree_node(g)[0], int(np.nonzero(np.diff(g.row_offsets) == 1)[0][0])]))
    for g, srcs in graphs:
        for mark_pred in (False, True):
            p = ga.BfsProblem(mark_pred, True).init(g.nodes, g.row_offsets, g.col_indices)
            p.set_inverse_graph()
            p.set_tuning(tail_edge_limit=tail_limit)
            p.set_persistent_limit(persistent_limit)
            p.set_twc_limit(0)                              # (mode 1 would otherwise run its small levels in the TWC workgroup)
            p.set_cooperative_launch(mark_pred)             # half of the runs through hipLaunchCooperativeKernel
            for src in srcs:
                for mode in (0, 1, 2):
                    p.reset(int(src))
                    p.enact(int(src), traversal_mode=mode)
                    labels, preds = p.extract()
                    _check(g, int(src), labels, preds, p.stats())
            p.close()


@pytest.mark.parametrize("min_edges,lite_factor,tail_limit", [(1, 1e9, 0), (1, 1e9, 100), (1000, 8.0, 8192), (1, 0.5, 64)])
def test_head_pass_then_count_only_level_parity(min_edges, lite_factor, tail_limit):
    # "heads, then the rest" levels (bottom-up pass over the adjacency heads + count-only top-down advance + merge) forced at
    # small scale: labels and parents must equal the plain schedules'
    for scale, ef in [(12, 8), (16, 8), (18, 16)]:
        g = o.rmat_seeded(scale, ef << scale)
        deg = np.diff(g.row_offsets)
        srcs = [o.highest_degree_node(g)[0]] + np.nonzero((deg > 0) & (deg < 4))[0][:2].tolist()
        for mark_pred in (False, True):
            p = ga.BfsProblem(mark_pred, True).init(g.nodes, g.row_offsets, g.col_indices)
            p.set_inverse_graph()
            p.set_tuning(lite_factor=lite_factor, tail_edge_limit=tail_limit)
            p.set_head_pass(min_edges, 0)
            for src in srcs:
                p.reset(int(src))
                p.enact(int(src), traversal_mode=2)
                labels, preds = p.extract()
                _check(g, int(src), labels, preds, p.stats())
            p.close()


@pytest.mark.parametrize("deferral,mask_limit", [(1, 0), (1, 4), (0, 0)])
@pytest.mark.parametrize("beta,lite_factor,min_edges", [(0.0, -1.0, -1), (1e12, 1e9, 1), (1e12, 0.0, 0)])
def test_deferred_labels_equal_labels_at_discovery(deferral, mask_limit, beta, lite_factor, min_edges):
    # Direction-optimizing searches keep the bitmaps of their vertex-ordered levels and write every label in ONE pass at the end
    # of Enact (Reset fills nothing).  Same labels with the deferral on, off, and with a pool of only 4 bitmaps, which forces the
    # kept levels to be flushed in the middle of a search (beta huge: the search never returns to top-down, so every level after the
    # switch is a bitmap level).  The problem is reused across sources and modes, so labels of the PREVIOUS search lie in the
    # array when the next one starts -- the emit pass must overwrite every one of them, also after a top-down-only Enact.
    for scale, ef in [(10, 8), (17, 16)]:
        g = o.rmat_seeded(scale, ef << scale)
        deg = np.diff(g.row_offsets)
        srcs = [o.highest_degree_node(g)[0]] + np.nonzero((deg > 0) & (deg < 4))[0][:2].tolist() + [int(np.nonzero(deg == 0)[0][0])]
        for mark_pred in (False, True):
            p = ga.BfsProblem(mark_pred, True).init(g.nodes, g.row_offsets, g.col_indices)
            p.set_inverse_graph(beta=beta)
            p.set_tuning(lite_factor=lite_factor)
            p.set_head_pass(min_edges, 0 if min_edges >= 0 else -1)
            p.set_label_deferral(deferral, mask_limit)
            for src in srcs:
                for mode in (2, 0, 2):
                    p.reset(int(src))
                    p.enact(int(src), traversal_mode=mode)
                    labels, preds = p.extract()
                    _check(g, int(src), labels, preds, p.stats())
            # Reset without Enact: Extract must still hand back a fully defined array (source 0, everything else -1)
            p.reset(int(srcs[0]))
            labels, _ = p.extract()
            want = np.full(g.nodes, -1, dtype=np.int32)
            want[int(srcs[0])] = 0
            assert np.array_equal(labels, want)
            p.close()


@pytest.mark.parametrize("chain", [0, 2, 6])
@pytest.mark.parametrize("beta,lite_factor,min_edges,sparse_div,emit_factor", [(0.0, -1.0, -1, 16, 32.0), (1e12, 1e9, 1, 1, 1e12),
                                                                                (24.0, 0.0, 0, 0, 0.0), (1e12, 1e9, 0, 16, 1e12)])
def test_chained_sweeps_decide_on_the_device_like_the_host(chain, beta, lite_factor, min_edges, sparse_div, emit_factor):
    # bottom-up levels are queued several at a time and every sweep applies the direction rules itself (stop / back to top-down /
    # dense / compacting / compacting + emitted queue); chain = 0 is the host-driven schedule.  Same labels and valid parents for
    # every chain length, rule setting and both label modes; the enactor itself fails the search if its replay of the rules
    # disagrees with what the device logged.
    for scale, ef in [(10, 8), (17, 16)]:
        g = o.rmat_seeded(scale, ef << scale)
        deg = np.diff(g.row_offsets)
        srcs = [o.highest_degree_node(g)[0]] + np.nonzero((deg > 0) & (deg < 4))[0][:2].tolist()
        for mark_pred, deferral, mask_limit in ((False, 1, 0), (True, 1, 4), (False, 0, 0)):
            p = ga.BfsProblem(mark_pred, True).init(g.nodes, g.row_offsets, g.col_indices)
            p.set_inverse_graph(beta=beta)
            p.set_tuning(lite_factor=lite_factor)
            p.set_head_pass(min_edges, 0 if min_edges >= 0 else -1)
            p.set_label_deferral(deferral, mask_limit)
            p.set_option("chain_sweeps", chain).set_option("sparse_sweep_div", sparse_div).set_option("emit_queue_factor", emit_factor)
            p.set_option("chain_closing", 1 if mark_pred else 0)   # (the closing levels queued behind the chain, or launched by the host)
            for src in srcs:
                p.reset(int(src))
                p.enact(int(src), traversal_mode=2)
                labels, preds = p.extract()
                _check(g, int(src), labels, preds, p.stats())
            p.close()


def test_deferred_labels_on_a_long_bottom_up_run():
    # a path-like graph searched bottom-up only (alpha and beta huge): hundreds of bitmap levels through a pool of 4..12 bitmaps
    n = 3000
    rows = np.arange(n - 1, dtype=np.int32)
    g = ga.HostGraph.from_coo(n, np.concatenate([rows, rows + 1]), np.concatenate([rows + 1, rows]))
    og = o.Csr(g.nodes, g.row_offsets, g.col_indices)
    for mask_limit, chain in ((4, 3), (12, 6), (4, 0), (5, 6)):
        p = ga.BfsProblem(True, True).init(g.nodes, g.row_offsets, g.col_indices)
        p.set_inverse_graph(alpha=1e12, beta=1e12)
        p.set_tuning(tail_edge_limit=0)
        p.set_label_deferral(1, mask_limit)
        p.set_option("chain_sweeps", chain)
        for src in (0, n // 2):
            p.reset(src)
            p.enact(src, traversal_mode=2)
            labels, preds = p.extract()
            assert np.array_equal(labels, o.bfs(og, src)[0])
            assert o.check_bfs_preds(og, src, labels, preds) == 0
        p.close()


@pytest.mark.parametrize("min_edges,lite_factor,beta", [(1, 1e9, 0.0), (1, 8.0, 2.0), (0, 1e9, 50.0), (-1, -1.0, 0.0)])
def test_directed_graph_new_level_kinds(min_edges, lite_factor, beta):
    # directed R-MAT with an explicit in-neighbour CSR: heads ranked by IN-degree, compacted head indices, the heads-then-rest
    # level, the compacting sweep and its emitted queue (which needs FORWARD row extents) must all use the right graph
    import torch
    for scale in (14, 17):
        g = o.rmat_seeded(scale, 8 << scale, undirected=False)
        src_of = np.repeat(np.arange(g.nodes, dtype=np.int32), np.diff(g.row_offsets))
        inv0 = ga.HostGraph.from_coo(g.nodes, g.col_indices, src_of)
        iro = torch.tensor(inv0.row_offsets, dtype=torch.int32, device="cuda")
        ici = torch.tensor(inv0.col_indices, dtype=torch.int32, device="cuda")
        outdeg = np.diff(g.row_offsets)
        picks = [int(np.argmax(outdeg)), int(np.nonzero((outdeg > 0) & (outdeg < 3))[0][0])]
        for mark_pred in (False, True):
            p = ga.BfsProblem(mark_pred, True).init(g.nodes, g.row_offsets, g.col_indices)
            p.set_inverse_graph(iro.data_ptr(), ici.data_ptr(), 0.0, beta)
            p.set_tuning(lite_factor=lite_factor)
            p.set_head_pass(min_edges, 0 if min_edges >= 0 else -1)
            for src in picks:
                p.reset(src)
                p.enact(src, traversal_mode=2)
                labels, preds = p.extract()
                _check(g, src, labels, preds, p.stats())
            p.close()


@pytest.mark.parametrize("min_edges,tail_limit", [(1, 0), (1, 8192), (5000, 8192), (1 << 40, 8192)])
def test_binned_advance_parity(min_edges, tail_limit):
    # destination-binned top-down levels (expand + screen -> per-XCD bins -> claims without atomics -> closing sweep) forced at
    # small scale in all four modes and both traversal schedules; the last parameter set switches the path off
    graphs = []
    for scale, ef, und in [(12, 8, True), (16, 16, True), (18, 8, True), (15, 8, False)]:
        g = o.rmat_seeded(scale, ef << scale, undirected=und)
        deg = np.diff(g.row_offsets)
        graphs.append((g, und, [o.highest_degree_node(g)[0]] + np.nonzero((deg > 0) & (deg < 4))[0][:2].tolist()))
    hub = 70000                                          # one row far larger than a tile and a chunk
    graphs.append((o.Csr(hub, np.concatenate(([0], np.full(hub, hub - 1, dtype=np.int32))), np.arange(1, hub, dtype=np.int32)),
                   False, [0]))
    for g, und, srcs in graphs:
        for mark_pred, idempotence in MODES:
            p = ga.BfsProblem(mark_pred, idempotence).init(g.nodes, g.row_offsets, g.col_indices)
            if und:
                p.set_inverse_graph()
            p.set_tuning(tail_edge_limit=tail_limit)
            p.set_binned_min_edges(min_edges)
            for src in srcs:
                for mode in ((0, 2) if und else (0,)):
                    p.reset(int(src))
                    p.enact(int(src), traversal_mode=mode)
                    labels, preds = p.extract()
                    _check(g, int(src), labels, preds, p.stats())
            p.close()


def test_c_abi_entry_point_picks_direction_optimizing(capfd):
    # gunrock_bfs_func runs the direction-optimizing schedule (the reference's entry point is top-down only, bfs_app.cu:196-200;
    # its DOBFS primitive takes the inverse graph from the caller, dobfs_enactor.cuh:397,569): on a graph that is its own inverse
    # directly, on a directed graph over the transpose it builds on the device; a small graph stays top-down.  Labels are the
    # oracle's either way.
    for scale, und, expect in [(16, True, "symmetric input: direction-optimizing"), (16, False, "directed input: direction-optimizing"),
                               (10, True, None)]:
        g = o.rmat_seeded(scale, 8 << scale, undirected=und)
        src, _ = o.highest_degree_node(g)
        for mark_pred, idem in [(False, True), (True, False)]:
            capfd.readouterr()
            labels = ga.gunrock_bfs(g.nodes, g.row_offsets, g.col_indices, src=int(src), mark_pred=mark_pred, idempotence=idem)
            ref, _, _ = o.bfs(g, int(src))
            assert np.array_equal(labels, ref)
            out = capfd.readouterr().out
            if expect is None:
                assert "direction-optimizing traversal" not in out, out
            else:
                assert expect in out, out


def _oriented(g, keep):
    """Sub-graph of g with the edges (f, t) for which keep(f, t) holds (rows stay sorted and duplicate-free)."""
    froms = np.repeat(np.arange(g.nodes, dtype=np.int64), np.diff(g.row_offsets))
    tos = g.col_indices.astype(np.int64)
    m = keep(froms, tos)
    ro = np.zeros(g.nodes + 1, dtype=np.int32)
    np.cumsum(np.bincount(froms[m], minlength=g.nodes), out=ro[1:])
    return o.Csr(g.nodes, ro, g.col_indices[m].astype(np.int32))


@pytest.mark.parametrize("shape", ["downward", "upward", "mixed"])
def test_symmetry_check_sees_one_way_edges(shape, capfd):
    # a graph whose unmirrored edges all run from a higher to a lower id ({1 -> 0} is the smallest) must not pass as
    # symmetric -- out-lists would be taken for in-lists, sinks would be preloaded as visited and never labelled.  >= 65536 edges
    # so gunrock_bfs_func takes the direction-optimizing branch; expected: the device-built inverse, labels = oracle.
    g0 = o.rmat_seeded(15, 8 << 15, undirected=True)
    if shape == "downward":
        g = _oriented(g0, lambda f, t: f > t)
    elif shape == "upward":
        g = _oriented(g0, lambda f, t: f < t)
    else:  # mirrored pairs below 4096, one-way (downward) edges elsewhere
        g = _oriented(g0, lambda f, t: (f > t) | ((f < 4096) & (t < 4096)))
    assert g.edges >= 1 << 16
    deg = np.diff(g.row_offsets)
    for src in (int(np.argmax(deg)), g.nodes - 1, 0):
        capfd.readouterr()
        labels = ga.gunrock_bfs(g.nodes, g.row_offsets, g.col_indices, src=src, mark_pred=False, idempotence=True)
        out = capfd.readouterr().out
        ref, _, _ = o.bfs(g, src)
        assert np.array_equal(labels, ref), shape
        assert "directed input: direction-optimizing" in out, out
    # phase-level entry: same decision
    p = ga.BfsProblem(mark_pred=True, idempotence=False).init(g.nodes, g.row_offsets, g.col_indices)
    try:
        enabled, built, _ = p.auto_inverse()
        assert enabled and built
        src = int(np.argmax(deg))
        p.reset(src)
        p.enact(src, traversal_mode=2)
        labels, preds = p.extract()
        ref, _, _ = o.bfs(g, src)
        assert np.array_equal(labels, ref)
        assert o.check_bfs_preds(g, src, labels, preds) == 0
    finally:
        p.close()


def test_instrumented_enactor_reports_cta_duty():
    # KernelRuntimeStats role (reference kernel_runtime_stats.cuh:226-279, bfs_enactor.cuh:173-186): avg duty = sum of
    # workgroup runtimes / (longest runtime x workgroups), over the operator launches of the last Enact; 0 when not instrumented
    g = o.rmat_seeded(16, 8 << 16)
    src, _ = o.highest_degree_node(g)
    for mode in (0, 2):
        labels, _, st, _ = _run(g, int(src), False, True, instrument=True, mode=mode)
        ref, _, _ = o.bfs(g, int(src))
        assert np.array_equal(labels, ref)
        assert 0.0 < st["avg_duty"] <= 1.0, st
    _, _, st, _ = _run(g, int(src), False, True, instrument=False)
    assert st["avg_duty"] == 0.0
