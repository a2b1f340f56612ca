"""ctypes binding for the CPU oracle (liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (gunrockinst_amd) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)


class _Csr(C.Structure):
    _fields_ = [("nodes", C.c_int32), ("edges", C.c_int32),
                ("row_offsets", i32p), ("col_indices", i32p), ("edge_values", i32p)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "gr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.gro_build_market.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(_Csr)]
        L.gro_rmat_reference.argtypes = [C.c_int32, C.c_int32, C.c_int] + [C.c_double] * 4 + [C.POINTER(_Csr)]
        L.gro_rmat_seeded.argtypes = [C.c_int, C.c_int64, C.c_uint64, C.c_int] + [C.c_double] * 4 + [C.POINTER(_Csr)]
        L.gro_rmat_seeded_coo.argtypes = [C.c_int, C.c_int64, C.c_uint64, C.c_int] + [C.c_double] * 4 + \
                                         [C.c_int64, C.c_int64, i32p, i32p]
        L.gro_rmat_seeded_coo.restype = None
        L.gro_highest_degree_node.argtypes = [i32p, C.c_int32, i32p]
        L.gro_highest_degree_node.restype = C.c_int32
        L.gro_average_degree.argtypes = [i32p, C.c_int32]
        L.gro_average_degree.restype = C.c_int32
        L.gro_bfs.argtypes = [i32p, i32p, C.c_int32, C.c_int32, i32p, i32p]
        L.gro_bfs.restype = C.c_int32
        L.gro_sssp.argtypes = [i32p, i32p, u32p, C.c_int32, C.c_int32, u32p, i32p]
        L.gro_sssp.restype = None
        L.gro_cc.argtypes = [i32p, i32p, C.c_int32, i32p]
        L.gro_cc.restype = C.c_int32
        L.gro_cc_reference_schedule.argtypes = [i32p, i32p, C.c_int32, i32p, i32p, i32p]
        L.gro_cc_reference_schedule.restype = C.c_int32
        L.gro_bfs_stats.argtypes = [i32p, C.c_int32, i32p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.gro_bfs_stats.restype = None
        L.gro_check_bfs_preds.argtypes = [i32p, i32p, C.c_int32, C.c_int32, i32p, i32p]
        L.gro_check_bfs_preds.restype = C.c_int64
        L.gro_check_sssp_preds.argtypes = [i32p, i32p, u32p, C.c_int32, C.c_int32, u32p, i32p]
        L.gro_check_sssp_preds.restype = C.c_int64
        L.gro_pagerank.argtypes = [i32p, i32p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32, C.POINTER(C.c_double), i32p, i32p]
        L.gro_pagerank.restype = C.c_int
        L.gro_topk.argtypes = [i32p, i32p, C.c_int32, C.c_int32, i32p, i32p, i32p]
        L.gro_topk.restype = None
        L.gro_csr_free.argtypes = [C.POINTER(_Csr)]
        L.gro_csr_free.restype = None
        L.gro_srand.argtypes = [C.c_uint]
        _LIB = L
    return _LIB


def _p(a, t=i32p):
    return a.ctypes.data_as(t)


class Csr:
    """Host CSR as numpy arrays (int32), same fields as gunrock::Csr (csr.cuh:38-80)."""

    def __init__(self, nodes, row_offsets, col_indices, edge_values=None):
        self.nodes = int(nodes)
        self.row_offsets = np.ascontiguousarray(row_offsets, dtype=np.int32)
        self.col_indices = np.ascontiguousarray(col_indices, dtype=np.int32)
        self.edges = int(self.col_indices.shape[0])
        self.edge_values = None if edge_values is None else np.ascontiguousarray(edge_values, dtype=np.int32)

    @property
    def weights_u32(self):
        return self.edge_values.view(np.uint32)


def _take(c):
    n, m = c.nodes, c.edges
    ro = np.ctypeslib.as_array(c.row_offsets, shape=(n + 1,)).copy()
    ci = np.ctypeslib.as_array(c.col_indices, shape=(max(m, 1),))[:m].copy()
    ev = np.ctypeslib.as_array(c.edge_values, shape=(max(m, 1),))[:m].copy()
    lib().gro_csr_free(C.byref(c))
    return Csr(n, ro, ci, ev)


def build_market(path, undirected=False, reversed_=False):
    c = _Csr()
    rc = lib().gro_build_market(os.fsencode(path), int(undirected), int(reversed_), C.byref(c))
    if rc != 0:
        raise ValueError("oracle: cannot parse MARKET file %s" % path)
    return _take(c)


class _Tuple(C.Structure):
    _fields_ = [("row", C.c_int32), ("col", C.c_int32), ("val", C.c_int64)]


def from_coo(nodes, rows, cols, undirected=False):
    """Csr::FromCoo (csr.cuh:247-340) on explicit tuples; `undirected` appends the mirrored tuple right after each entry
    like the reference's loaders (market.cuh:173-184).  Values are all 1."""
    rows = np.asarray(rows, dtype=np.int32)
    cols = np.asarray(cols, dtype=np.int32)
    k = 2 if undirected else 1
    t = (_Tuple * max(k * rows.shape[0], 1))()
    arr = np.ctypeslib.as_array(t).view([("row", "<i4"), ("col", "<i4"), ("val", "<i8")]) if rows.shape[0] else None
    if rows.shape[0]:
        arr = arr.reshape(-1)
        arr["val"][:] = 1
        if undirected:
            arr["row"][0::2], arr["col"][0::2] = rows, cols
            arr["row"][1::2], arr["col"][1::2] = cols, rows
        else:
            arr["row"][:], arr["col"][:] = rows, cols
    out = _Csr()
    rc = lib().gro_csr_from_coo(t, int(nodes), int(k * rows.shape[0]), C.byref(out))
    assert rc == 0
    return _take(out)


def rmat_reference(nodes, edges, undirected=False, a=0.55, b=0.2, c=0.2, d=0.05, srand=1):
    if srand is not None:
        lib().gro_srand(srand)
    out = _Csr()
    rc = lib().gro_rmat_reference(nodes, edges, int(undirected), a, b, c, d, C.byref(out))
    assert rc == 0
    return _take(out)


def rmat_seeded(scale, pairs, seed=0x6772, undirected=True, a=0.55, b=0.2, c=0.2, d=0.05):
    out = _Csr()
    rc = lib().gro_rmat_seeded(scale, pairs, seed, int(undirected), a, b, c, d, C.byref(out))
    assert rc == 0
    return _take(out)


def rmat_seeded_coo(scale, first, count, seed=0x6772, a=0.55, b=0.2, c=0.2, d=0.05):
    rows = np.empty(count, dtype=np.int32)
    cols = np.empty(count, dtype=np.int32)
    lib().gro_rmat_seeded_coo(scale, 0, seed, 0, a, b, c, d, first, count, _p(rows), _p(cols))
    return rows, cols


def highest_degree_node(g):
    md = C.c_int32()
    return int(lib().gro_highest_degree_node(_p(g.row_offsets), g.nodes, C.byref(md))), int(md.value)


def average_degree(g):
    return int(lib().gro_average_degree(_p(g.row_offsets), g.nodes))


def bfs(g, src, want_preds=False):
    labels = np.empty(g.nodes, dtype=np.int32)
    preds = np.empty(g.nodes, dtype=np.int32) if want_preds else None
    depth = lib().gro_bfs(_p(g.row_offsets), _p(g.col_indices), g.nodes, src, _p(labels),
                          _p(preds) if want_preds else None)
    return labels, preds, int(depth)


def bfs_parallel(g, src, threads=0):
    """OpenMP level-synchronous BFS (labels identical to bfs()); returns (labels, threads used)."""
    labels = np.empty(g.nodes, dtype=np.int32)
    used = lib().gro_bfs_parallel(_p(g.row_offsets), _p(g.col_indices), g.nodes, src, _p(labels), int(threads))
    return labels, int(used)


def sssp(g, src, weights=None):
    w = np.ascontiguousarray(g.weights_u32 if weights is None else weights, dtype=np.uint32)
    dist = np.empty(g.nodes, dtype=np.uint32)
    preds = np.empty(g.nodes, dtype=np.int32)
    lib().gro_sssp(_p(g.row_offsets), _p(g.col_indices), _p(w, u32p), g.nodes, src, _p(dist, u32p), _p(preds))
    return dist, preds


def cc(g):
    comp = np.empty(g.nodes, dtype=np.int32)
    count = lib().gro_cc(_p(g.row_offsets), _p(g.col_indices), g.nodes, _p(comp))
    return comp, int(count)


def cc_reference_schedule(g):
    comp = np.empty(g.nodes, dtype=np.int32)
    ih, ij = C.c_int32(), C.c_int32()
    count = lib().gro_cc_reference_schedule(_p(g.row_offsets), _p(g.col_indices), g.nodes, _p(comp),
                                            C.byref(ih), C.byref(ij))
    return comp, int(count), int(ih.value), int(ij.value)


def bc(g, src=-1):
    """Brandes betweenness centrality (halved, reference convention); returns (bc float64[n], sigma of the last source)."""
    out = np.empty(max(g.nodes, 1), dtype=np.float64)
    sig = np.empty(max(g.nodes, 1), dtype=np.float64)
    f64p = C.POINTER(C.c_double)
    rc = lib().gro_bc(_p(g.row_offsets), _p(g.col_indices), g.nodes, int(src), out.ctypes.data_as(f64p), sig.ctypes.data_as(f64p))
    assert rc == 0
    return out[:g.nodes], sig[:g.nodes]


def pagerank(g, src=-1, delta=0.85, threshold=0.01, max_iter=20):
    """The reference's PageRank schedule in doubles; returns (rank float64[n], degrees after peeling int32[n], iterations).
    PARITY UNPINNED: see gr_oracle.c."""
    rank = np.empty(max(g.nodes, 1), dtype=np.float64)
    deg = np.empty(max(g.nodes, 1), dtype=np.int32)
    it = C.c_int32()
    rc = lib().gro_pagerank(_p(g.row_offsets), _p(g.col_indices), g.nodes, int(src), float(delta), float(threshold), int(max_iter),
                            rank.ctypes.data_as(C.POINTER(C.c_double)), _p(deg), C.byref(it))
    assert rc == 0
    return rank[:g.nodes], deg[:g.nodes], int(it.value)


def topk(g, k, col_offsets=None):
    k = min(int(k), g.nodes)
    ids, ind, outd = (np.empty(max(k, 1), dtype=np.int32) for _ in range(3))
    co = None if col_offsets is None else np.ascontiguousarray(col_offsets, dtype=np.int32)
    lib().gro_topk(_p(g.row_offsets), None if co is None else _p(co), g.nodes, k, _p(ids), _p(ind), _p(outd))
    return ids[:k], ind[:k], outd[:k]


def bfs_stats(g, labels):
    nv, ev = C.c_int64(), C.c_int64()
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    lib().gro_bfs_stats(_p(g.row_offsets), g.nodes, _p(labels), C.byref(nv), C.byref(ev))
    return int(nv.value), int(ev.value)


def check_bfs_preds(g, src, labels, preds):
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    preds = np.ascontiguousarray(preds, dtype=np.int32)
    return int(lib().gro_check_bfs_preds(_p(g.row_offsets), _p(g.col_indices), g.nodes, src, _p(labels), _p(preds)))


def check_sssp_preds(g, src, dist, preds, weights=None):
    w = np.ascontiguousarray(g.weights_u32 if weights is None else weights, dtype=np.uint32)
    dist = np.ascontiguousarray(dist, dtype=np.uint32)
    preds = np.ascontiguousarray(preds, dtype=np.int32)
    return int(lib().gro_check_sssp_preds(_p(g.row_offsets), _p(g.col_indices), _p(w, u32p), g.nodes, src,
                                          _p(dist, u32p), _p(preds)))
