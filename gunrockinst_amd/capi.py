"""ctypes binding of libgunrock.so.

Struct layouts mirror include/gunrock/gunrock.h (reference gunrock/gunrock.h:51-99) field by field.
No compute happens in Python and nothing here falls back to a CPU implementation: if the HIP
library is missing, :func:`lib` raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# GUNROCK_LIB_PATH: another build of the same library (tuning variants from tools/build_variant.sh); still no fallback.
LIB_PATH = os.environ.get("GUNROCK_LIB_PATH") or os.path.join(_PKG, "lib", "libgunrock.so")
_LIB = None

# enum VertexIdType / SizeTType / ValueType / SrcMode (gunrock.h)
VTXID_INT = 0
SIZET_INT = 0
VALUE_INT, VALUE_UINT, VALUE_FLOAT = 0, 1, 2
SRC_MANUALLY, SRC_RANDOMIZE, SRC_LARGEST_DEGREE = 0, 1, 2

i32p = C.POINTER(C.c_int32)


class GunrockDataType(C.Structure):
    _fields_ = [("VTXID_TYPE", C.c_int), ("SIZET_TYPE", C.c_int), ("VALUE_TYPE", C.c_int)]


class GunrockGraph(C.Structure):
    _fields_ = [("num_nodes", C.c_size_t), ("num_edges", C.c_size_t),
                ("row_offsets", C.c_void_p), ("col_indices", C.c_void_p),
                ("col_offsets", C.c_void_p), ("row_indices", C.c_void_p),
                ("node_values", C.c_void_p), ("edge_values", C.c_void_p)]


class GunrockConfig(C.Structure):
    _fields_ = [("mark_pred", C.c_bool), ("idempotence", C.c_bool),
                ("src_node", C.c_int), ("device", C.c_int), ("max_iter", C.c_int),
                ("top_nodes", C.c_int), ("delta_factor", C.c_int),
                ("delta", C.c_float), ("error", C.c_float), ("queue_size", C.c_float),
                ("src_mode", C.c_int)]


def build_library(force=False):
    """Compile libgunrock.so for gfx950 (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_PKG, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src_dir, "clean"])
    subprocess.check_call(["make", "-C", src_dir, "-j8", "-s"])
    return LIB_PATH


# every symbol declared in include/gunrock/*.h (tests/test_capi_symbols.py checks the list against the headers)
_SIGNATURES = {
    "gunrock_bfs_func": (None, [C.POINTER(GunrockGraph), C.POINTER(GunrockGraph), GunrockConfig, GunrockDataType]),
    "gunrock_bc_func": (None, [C.POINTER(GunrockGraph), C.POINTER(GunrockGraph), GunrockConfig, GunrockDataType]),
    "gunrock_cc_func": (None, [C.POINTER(GunrockGraph), C.POINTER(GunrockGraph), GunrockConfig, GunrockDataType]),
    "gunrock_sssp_func": (None, [C.POINTER(GunrockGraph), C.c_void_p, C.POINTER(GunrockGraph), GunrockConfig,
                                 GunrockDataType]),
    "gunrock_pr_func": (None, [C.POINTER(GunrockGraph), C.c_void_p, C.c_void_p, C.POINTER(GunrockGraph),
                               GunrockConfig, GunrockDataType]),
    "gunrock_topk_func": (None, [C.POINTER(GunrockGraph), C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(GunrockGraph), GunrockConfig, GunrockDataType]),
    "grx_graph_from_market": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "grx_graph_from_market_cached": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "grx_graph_rmat_libc": (C.c_int, [C.c_int, C.c_int, C.c_int] + [C.c_double] * 4 + [C.POINTER(C.c_void_p)]),
    "grx_graph_rmat_seeded": (C.c_int, [C.c_int, C.c_longlong, C.c_uint64, C.c_int] + [C.c_double] * 4 +
                              [C.POINTER(C.c_void_p)]),
    "grx_graph_from_coo": (C.c_int, [C.c_int, C.c_longlong, i32p, i32p, i32p, C.POINTER(C.c_void_p)]),
    "grx_graph_from_csr": (C.c_int, [C.c_int, C.c_int, i32p, i32p, i32p, C.POINTER(C.c_void_p)]),
    "grx_graph_nodes": (C.c_int, [C.c_void_p]),
    "grx_graph_edges": (C.c_int, [C.c_void_p]),
    "grx_graph_row_offsets": (i32p, [C.c_void_p]),
    "grx_graph_col_indices": (i32p, [C.c_void_p]),
    "grx_graph_edge_values": (i32p, [C.c_void_p]),
    "grx_graph_highest_degree_node": (C.c_int, [C.c_void_p, i32p]),
    "grx_graph_average_degree": (C.c_int, [C.c_void_p]),
    "grx_random_node": (C.c_int, [C.c_int]),
    "grx_graph_free": (None, [C.c_void_p]),
    "grx_rmat_seeded_device": (C.c_int, [C.c_int, C.c_longlong, C.c_longlong, C.c_uint64] + [C.c_double] * 4 +
                               [C.c_void_p, C.c_void_p, C.c_void_p]),
    "grx_coo_to_csr_sort": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_longlong, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_int, C.c_int, C.POINTER(C.c_longlong), C.c_void_p]),
    "grx_coo_to_csr_emit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "grx_coo_to_csr_free": (None, [C.c_void_p]),
    "grx_bc_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "grx_bc_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "grx_bc_init_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "grx_bc_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_float)]),
    "grx_bc_extract": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "grx_bc_destroy": (None, [C.c_void_p]),
    "grx_bfs_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int]),
    "grx_bfs_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, i32p, i32p]),
    "grx_bfs_init_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "grx_bfs_set_inverse_graph": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float]),
    "grx_bfs_auto_inverse": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "grx_bfs_set_tuning": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int]),
    "grx_bfs_set_persistent_limit": (C.c_int, [C.c_void_p, C.c_int]),
    "grx_bfs_set_twc_limit": (C.c_int, [C.c_void_p, C.c_int]),
    "grx_bfs_set_binned_min_edges": (C.c_int, [C.c_void_p, C.c_longlong]),
    "grx_bfs_set_label_deferral": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "grx_bfs_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "grx_bfs_set_cooperative_launch": (C.c_int, [C.c_void_p, C.c_int]),
    "grx_bfs_set_head_pass": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "grx_bfs_reset": (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    "grx_bfs_enact": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "grx_bfs_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]),
    "grx_bfs_level_trace": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                      C.POINTER(C.c_double), i32p]),
    "grx_bfs_extract": (C.c_int, [C.c_void_p, i32p, i32p]),
    "grx_bfs_device_results": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "grx_bfs_destroy": (None, [C.c_void_p]),
    "grx_cc_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "grx_cc_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, i32p, i32p]),
    "grx_cc_init_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "grx_cc_reset": (C.c_int, [C.c_void_p]),
    "grx_cc_enact": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "grx_cc_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                               C.POINTER(C.c_double)]),
    "grx_cc_mirrored": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "grx_cc_sweep_edges": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong)]),
    "grx_cc_extract": (C.c_int, [C.c_void_p, i32p, C.POINTER(C.c_uint)]),
    "grx_cc_device_results": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "grx_cc_destroy": (None, [C.c_void_p]),
    "grx_sssp_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    "grx_sssp_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, i32p, i32p, C.POINTER(C.c_uint32), C.c_int]),
    "grx_sssp_init_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]),
    "grx_sssp_set_inverse_graph": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]),
    "grx_sssp_pull_levels": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong)]),
    "grx_filter_queue": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int),
                                   C.POINTER(C.c_longlong), C.c_int]),
    "grx_sssp_reset": (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    "grx_sssp_enact": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "grx_sssp_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong),
                                 C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_float)]),
    "grx_sssp_extract": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), i32p]),
    "grx_sssp_destroy": (None, [C.c_void_p]),
    "grx_pr_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "grx_pr_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, i32p, i32p]),
    "grx_pr_init_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "grx_pr_set_inverse_graph": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "grx_pr_reset": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float]),
    "grx_pr_enact": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "grx_pr_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "grx_pr_extract": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), i32p, C.c_int]),
    "grx_pr_device_results": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "grx_pr_destroy": (None, [C.c_void_p]),
    "grx_pbfs_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "grx_pbfs_init_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "grx_pbfs_reset": (C.c_int, [C.c_void_p, C.c_int]),
    "grx_pbfs_frontier": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "grx_pbfs_advance_local": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_void_p)]),
    "grx_pbfs_filter_received": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "grx_pbfs_queue_to_bitmap": (C.c_int, [C.c_void_p]),
    "grx_pbfs_frontier_bitmap": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "grx_pbfs_bottom_up": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "grx_pbfs_bitmap_to_queue": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "grx_pbfs_labels": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "grx_pbfs_preds": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "grx_pbfs_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "grx_pbfs_stat": (C.c_longlong, [C.c_void_p, C.c_char_p]),
    "grx_rccl_load": (C.c_int, []),
    "grx_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "grx_pbfs_comm_init_rccl": (C.c_int, [C.c_void_p, C.c_char_p]),
    "grx_pbfs_set_transport": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "grx_pbfs_set_options": (C.c_int, [C.c_void_p, C.c_int, C.c_float]),
    "grx_pbfs_search": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "grx_pbfs_destroy": (None, [C.c_void_p]),
    "grx_bfs_count_visited": (None, [C.c_int, i32p, i32p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "grx_version": (C.c_char_p, []),
}


def lib():
    """Load libgunrock.so (once).  Raises if the HIP library has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "gunrockinst_amd: %s is missing -- build it with `make -C gunrockinst_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def exported_symbols():
    return sorted(_SIGNATURES)


def version():
    return lib().grx_version().decode()


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("gunrockinst_amd: %s failed (code %d)" % (what, rc))


def _p(a):
    return a.ctypes.data_as(i32p)


class HostGraph:
    """gunrock::Csr<int,int,int> built by the library's own host graph code."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        L = lib()
        self.nodes = L.grx_graph_nodes(self._h)
        self.edges = L.grx_graph_edges(self._h)

    @classmethod
    def from_market(cls, path, undirected=False, reversed_=False, cache=False):
        """cache=True: the reference's CSR cache rule with a binary, stamped cache file next to the input; the instance's
        `cache_hit` tells whether the graph came from it."""
        h = C.c_void_p()
        if cache:
            hit = C.c_int()
            _check(lib().grx_graph_from_market_cached(os.fsencode(path), int(undirected), int(reversed_), C.byref(hit), C.byref(h)),
                   "BuildMarketGraphCached(%s)" % path)
            g = cls(h.value)
            g.cache_hit = bool(hit.value)
            return g
        _check(lib().grx_graph_from_market(os.fsencode(path), int(undirected), int(reversed_), C.byref(h)),
               "BuildMarketGraph(%s)" % path)
        return cls(h.value)

    @classmethod
    def rmat_libc(cls, nodes, edges, undirected=False, a=0.55, b=0.2, c=0.2, d=0.05):
        h = C.c_void_p()
        _check(lib().grx_graph_rmat_libc(nodes, edges, int(undirected), a, b, c, d, C.byref(h)), "BuildRmatGraph")
        return cls(h.value)

    @classmethod
    def rmat_seeded(cls, scale, pairs, seed=0x6772, undirected=True, a=0.55, b=0.2, c=0.2, d=0.05):
        h = C.c_void_p()
        _check(lib().grx_graph_rmat_seeded(scale, pairs, seed, int(undirected), a, b, c, d, C.byref(h)),
               "BuildSeededRmatGraph")
        return cls(h.value)

    @classmethod
    def from_coo(cls, nodes, rows, cols, vals=None):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        v = None if vals is None else np.ascontiguousarray(vals, dtype=np.int32)
        h = C.c_void_p()
        _check(lib().grx_graph_from_coo(nodes, rows.shape[0], _p(rows), _p(cols), None if v is None else _p(v),
                                        C.byref(h)), "Csr::FromCoo")
        return cls(h.value)

    @classmethod
    def from_csr(cls, nodes, row_offsets, col_indices, edge_values=None):
        ro = np.ascontiguousarray(row_offsets, dtype=np.int32)
        ci = np.ascontiguousarray(col_indices, dtype=np.int32)
        ev = None if edge_values is None else np.ascontiguousarray(edge_values, dtype=np.int32)
        h = C.c_void_p()
        _check(lib().grx_graph_from_csr(nodes, ci.shape[0], _p(ro), _p(ci), None if ev is None else _p(ev),
                                        C.byref(h)), "grx_graph_from_csr")
        return cls(h.value)

    @property
    def row_offsets(self):
        return np.ctypeslib.as_array(lib().grx_graph_row_offsets(self._h), shape=(self.nodes + 1,))

    @property
    def col_indices(self):
        if self.edges == 0:
            return np.empty(0, dtype=np.int32)
        return np.ctypeslib.as_array(lib().grx_graph_col_indices(self._h), shape=(self.edges,))

    @property
    def edge_values(self):
        p = lib().grx_graph_edge_values(self._h)
        if not p or self.edges == 0:
            return None
        return np.ctypeslib.as_array(p, shape=(self.edges,))

    def highest_degree_node(self):
        md = C.c_int32()
        v = lib().grx_graph_highest_degree_node(self._h, C.byref(md))
        return int(v), int(md.value)

    def average_degree(self):
        return int(lib().grx_graph_average_degree(self._h))

    def close(self):
        if self._h:
            lib().grx_graph_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BfsProblem:
    """BFSProblem + BFSEnactor behind the handle C ABI (Init once, Reset + Enact per run, Extract)."""

    def __init__(self, mark_pred=False, idempotence=False, instrument=False, device=0):
        self._h = C.c_void_p()
        self.mark_pred = bool(mark_pred)
        _check(lib().grx_bfs_create(C.byref(self._h), int(mark_pred), int(idempotence), int(instrument), device),
               "grx_bfs_create")
        self.nodes = 0

    def init(self, nodes, row_offsets, col_indices):
        ro = np.ascontiguousarray(row_offsets, dtype=np.int32)
        ci = np.ascontiguousarray(col_indices, dtype=np.int32)
        self.nodes = int(nodes)
        _check(lib().grx_bfs_init(self._h, nodes, ci.shape[0], _p(ro), _p(ci)), "BFSProblem::Init")
        return self

    def init_device(self, nodes, edges, d_row_offsets, d_col_indices):
        """d_* are integer device addresses (e.g. torch.Tensor.data_ptr()); borrowed, must outlive the problem."""
        self.nodes = int(nodes)
        _check(lib().grx_bfs_init_device(self._h, nodes, edges, C.c_void_p(d_row_offsets), C.c_void_p(d_col_indices)),
               "BFSProblem::Init(device)")
        return self

    def set_inverse_graph(self, d_inv_row_offsets=None, d_inv_col_indices=None, alpha=0.0, beta=0.0):
        """Enable traversal_mode=2 (direction-optimizing).  No arguments = the graph is symmetric."""
        _check(lib().grx_bfs_set_inverse_graph(self._h, C.c_void_p(d_inv_row_offsets), C.c_void_p(d_inv_col_indices),
                                               alpha, beta), "BFSProblem::SetInverseGraph")
        return self

    def auto_inverse(self, build_if_directed=True):
        """What gunrock_bfs_func does before its search: symmetric graph -> its own inverse; directed -> transpose built on the
        device (owned by the handle).  Returns (enabled, built, build_ms)."""
        on, made, ms = C.c_int(), C.c_int(), C.c_float()
        _check(lib().grx_bfs_auto_inverse(self._h, int(bool(build_if_directed)), C.byref(on), C.byref(made), C.byref(ms)),
               "grx_bfs_auto_inverse")
        return bool(on.value), bool(made.value), float(ms.value)

    def set_tuning(self, alpha=0.0, beta=0.0, lite_factor=-1.0, tail_edge_limit=-1):
        _check(lib().grx_bfs_set_tuning(self._h, alpha, beta, lite_factor, tail_edge_limit), "grx_bfs_set_tuning")
        return self

    def set_head_pass(self, min_edges=-1, max_edges=-1):
        _check(lib().grx_bfs_set_head_pass(self._h, int(min_edges), int(max_edges)), "grx_bfs_set_head_pass")
        return self

    def set_option(self, name, value):
        """Named enactor tuning knob (include/gunrock/gunrock_mi355x.h grx_bfs_set_option); results never depend on them."""
        _check(lib().grx_bfs_set_option(self._h, name.encode(), float(value)), "grx_bfs_set_option(%s)" % name)
        return self

    def set_label_deferral(self, enabled=-1, mask_limit=0):
        """Deferred labels of direction-optimizing searches (one emit pass at the end of Enact); effective at the next reset."""
        _check(lib().grx_bfs_set_label_deferral(self._h, int(enabled), int(mask_limit)), "grx_bfs_set_label_deferral")
        return self

    def set_binned_min_edges(self, min_edges):
        _check(lib().grx_bfs_set_binned_min_edges(self._h, int(min_edges)), "grx_bfs_set_binned_min_edges")
        return self

    def set_cooperative_launch(self, on=True):
        _check(lib().grx_bfs_set_cooperative_launch(self._h, int(bool(on))), "grx_bfs_set_cooperative_launch")
        return self

    def set_persistent_limit(self, edge_limit):
        _check(lib().grx_bfs_set_persistent_limit(self._h, int(edge_limit)), "grx_bfs_set_persistent_limit")
        return self

    def set_twc_limit(self, edge_limit):
        _check(lib().grx_bfs_set_twc_limit(self._h, int(edge_limit)), "grx_bfs_set_twc_limit")
        return self

    def reset(self, src, queue_sizing=1.0):
        _check(lib().grx_bfs_reset(self._h, int(src), float(queue_sizing)), "BFSProblem::Reset")

    def enact(self, src, max_grid_size=0, traversal_mode=0):
        ms = C.c_float()
        _check(lib().grx_bfs_enact(self._h, int(src), max_grid_size, traversal_mode, C.byref(ms)), "BFSEnactor::Enact")
        return float(ms.value)

    def stats(self):
        q, d, l = C.c_longlong(), C.c_longlong(), C.c_longlong()
        duty, kms = C.c_double(), C.c_double()
        _check(lib().grx_bfs_stats(self._h, C.byref(q), C.byref(d), C.byref(duty), C.byref(l), C.byref(kms)),
               "BFSEnactor::GetStatistics")
        return {"total_queued": q.value, "search_depth": d.value, "avg_duty": duty.value,
                "kernel_launches": l.value, "kernel_ms": kms.value}

    def level_trace(self, max_levels=4096):
        fr = (C.c_longlong * max_levels)()
        ed = (C.c_longlong * max_levels)()
        ms = (C.c_double * max_levels)()
        kd = (C.c_int32 * max_levels)()
        n = lib().grx_bfs_level_trace(self._h, max_levels, fr, ed, ms, kd)
        n = min(max(n, 0), max_levels)
        return [{"frontier": fr[i], "edges": ed[i], "ms": ms[i], "kind": kd[i]} for i in range(n)]

    def extract(self):
        labels = np.empty(max(self.nodes, 1), dtype=np.int32)
        preds = np.empty(max(self.nodes, 1), dtype=np.int32) if self.mark_pred else None
        _check(lib().grx_bfs_extract(self._h, _p(labels), None if preds is None else _p(preds)), "BFSProblem::Extract")
        return labels[:self.nodes], (None if preds is None else preds[:self.nodes])

    def device_results(self):
        dl, dp = C.c_void_p(), C.c_void_p()
        _check(lib().grx_bfs_device_results(self._h, C.byref(dl), C.byref(dp)), "grx_bfs_device_results")
        return dl.value, dp.value

    def close(self):
        if self._h:
            lib().grx_bfs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _graph_struct(nodes, row_offsets, col_indices, edge_values=None):
    g = GunrockGraph()
    g.num_nodes = nodes
    g.num_edges = col_indices.shape[0]
    g.row_offsets = row_offsets.ctypes.data
    g.col_indices = col_indices.ctypes.data
    g.edge_values = None if edge_values is None else edge_values.ctypes.data
    return g


def _take_node_values(gout, nodes, dtype):
    """graph_out->node_values is malloc()ed by the library and owned by the caller (bfs_app.cu:211)."""
    ptr = gout.node_values
    if not ptr:
        raise RuntimeError("gunrockinst_amd: the call produced no node_values")
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), shape=(max(nodes, 1),))[:nodes].copy()
    C.CDLL(None).free(C.c_void_p(ptr))
    return arr.view(dtype)


def gunrock_bfs(nodes, row_offsets, col_indices, src=0, mark_pred=False, idempotence=False, queue_size=1.0,
                src_mode=SRC_MANUALLY, device=0):
    """Call gunrock_bfs_func exactly as reference shared_lib_tests/test_bfs.c does; returns the labels."""
    ro = np.ascontiguousarray(row_offsets, dtype=np.int32)
    ci = np.ascontiguousarray(col_indices, dtype=np.int32)
    gin = _graph_struct(nodes, ro, ci)
    gout = GunrockGraph()
    cfg = GunrockConfig()
    cfg.mark_pred, cfg.idempotence = mark_pred, idempotence
    cfg.src_node, cfg.device, cfg.queue_size, cfg.src_mode = src, device, queue_size, src_mode
    dt = GunrockDataType(VTXID_INT, SIZET_INT, VALUE_INT)
    lib().gunrock_bfs_func(C.byref(gout), C.byref(gin), cfg, dt)
    return _take_node_values(gout, nodes, np.int32)


class CcProblem:
    """CCProblem + CCEnactor behind the handle C ABI."""

    def __init__(self, instrument=False, device=0):
        self._h = C.c_void_p()
        _check(lib().grx_cc_create(C.byref(self._h), int(instrument), device), "grx_cc_create")
        self.nodes = 0

    def init(self, nodes, row_offsets, col_indices):
        ro = np.ascontiguousarray(row_offsets, dtype=np.int32)
        ci = np.ascontiguousarray(col_indices, dtype=np.int32)
        self.nodes = int(nodes)
        _check(lib().grx_cc_init(self._h, nodes, ci.shape[0], _p(ro), _p(ci)), "CCProblem::Init")
        return self

    def init_device(self, nodes, edges, d_row_offsets, d_col_indices):
        self.nodes = int(nodes)
        _check(lib().grx_cc_init_device(self._h, nodes, edges, C.c_void_p(d_row_offsets), C.c_void_p(d_col_indices)),
               "CCProblem::Init(device)")
        return self

    def reset(self):
        _check(lib().grx_cc_reset(self._h), "CCProblem::Reset")

    def enact(self, max_grid_size=0):
        ms = C.c_float()
        _check(lib().grx_cc_enact(self._h, max_grid_size, C.byref(ms)), "CCEnactor::Enact")
        return float(ms.value)

    def stats(self):
        es, vs, l = C.c_longlong(), C.c_longlong(), C.c_longlong()
        k = C.c_double()
        _check(lib().grx_cc_stats(self._h, C.byref(es), C.byref(vs), C.byref(l), C.byref(k)), "grx_cc_stats")
        mir = C.c_int()
        _check(lib().grx_cc_mirrored(self._h, C.byref(mir)), "grx_cc_mirrored")
        se = C.c_longlong()
        _check(lib().grx_cc_sweep_edges(self._h, C.byref(se)), "grx_cc_sweep_edges")
        return {"edge_sweeps": es.value, "vertex_sweeps": vs.value, "kernel_launches": l.value, "kernel_ms": k.value,
                "mirrored": bool(mir.value), "sweep_edges": se.value}

    def extract(self):
        ids = np.empty(max(self.nodes, 1), dtype=np.int32)
        nc = C.c_uint()
        _check(lib().grx_cc_extract(self._h, _p(ids), C.byref(nc)), "CCProblem::Extract")
        return ids[:self.nodes], int(nc.value)

    def device_results(self):
        d = C.c_void_p()
        _check(lib().grx_cc_device_results(self._h, C.byref(d)), "grx_cc_device_results")
        return d.value

    def close(self):
        if self._h:
            lib().grx_cc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gunrock_cc(nodes, row_offsets, col_indices, device=0):
    """Call gunrock_cc_func as reference shared_lib_tests/test_cc.c does; returns the component ids."""
    ro = np.ascontiguousarray(row_offsets, dtype=np.int32)
    ci = np.ascontiguousarray(col_indices, dtype=np.int32)
    gin = _graph_struct(nodes, ro, ci)
    gout = GunrockGraph()
    cfg = GunrockConfig()
    cfg.device = device
    dt = GunrockDataType(VTXID_INT, SIZET_INT, VALUE_INT)
    lib().gunrock_cc_func(C.byref(gout), C.byref(gin), cfg, dt)
    return _take_node_values(gout, nodes, np.int32)


def gunrock_bc(nodes, row_offsets, col_indices, src=-1, queue_size=1.0, src_mode=SRC_MANUALLY, device=0):
    """Call gunrock_bc_func as reference shared_lib_tests/test_bc.c does (src -1 = every vertex in turn); returns
    (bc_values, ebc_values) as float32 arrays."""
    ro = np.ascontiguousarray(row_offsets, dtype=np.int32)
    ci = np.ascontiguousarray(col_indices, dtype=np.int32)
    gin = _graph_struct(nodes, ro, ci)
    gout = GunrockGraph()
    cfg = GunrockConfig()
    cfg.src_node, cfg.device, cfg.queue_size, cfg.src_mode = src, device, queue_size, src_mode
    dt = GunrockDataType(VTXID_INT, SIZET_INT, VALUE_FLOAT)
    lib().gunrock_bc_func(C.byref(gout), C.byref(gin), cfg, dt)
    edges = int(ci.shape[0])
    eptr = gout.edge_values
    bc = _take_node_values(gout, nodes, np.float32)
    if not eptr:
        raise RuntimeError("gunrockinst_amd: the call produced no edge_values")
    ebc = np.ctypeslib.as_array(C.cast(eptr, C.POINTER(C.c_float)), shape=(max(edges, 1),))[:edges].copy()
    C.CDLL(None).free(C.c_void_p(eptr))
    return bc, ebc


def gunrock_pr(nodes, row_offsets, col_indices, src=-1, delta=0.85, error=0.01, max_iter=20, top_nodes=0, src_mode=SRC_MANUALLY,
               device=0):
    """Call gunrock_pr_func as reference shared_lib_tests/test_pr.c does; returns (node_ids, page_rank) in descending rank order
    (top_nodes entries, all of them when top_nodes <= 0)."""
    ro = np.ascontiguousarray(row_offsets, dtype=np.int32)
    ci = np.ascontiguousarray(col_indices, dtype=np.int32)
    gin = _graph_struct(nodes, ro, ci)
    gout = GunrockGraph()
    cfg = GunrockConfig()
    cfg.src_node, cfg.device, cfg.src_mode = src, device, src_mode
    cfg.delta, cfg.error, cfg.max_iter, cfg.top_nodes = delta, error, max_iter, top_nodes
    count = nodes if top_nodes <= 0 else min(nodes, top_nodes)
    ids = np.empty(max(count, 1), dtype=np.int32)
    ranks = np.empty(max(count, 1), dtype=np.float32)
    dt = GunrockDataType(VTXID_INT, SIZET_INT, VALUE_FLOAT)
    lib().gunrock_pr_func(C.byref(gout), ids.ctypes.data_as(C.c_void_p), ranks.ctypes.data_as(C.c_void_p), C.byref(gin), cfg, dt)
    return ids[:count], ranks[:count]


def gunrock_topk(nodes, row_offsets, col_indices, col_offsets, row_indices, top_nodes, device=0):
    """Call gunrock_topk_func as reference shared_lib_tests/test_topk.c does; returns (node_ids, in_degrees, out_degrees)."""
    ro = np.ascontiguousarray(row_offsets, dtype=np.int32)
    ci = np.ascontiguousarray(col_indices, dtype=np.int32)
    co = np.ascontiguousarray(col_offsets, dtype=np.int32)
    ri = np.ascontiguousarray(row_indices, dtype=np.int32)
    gin = _graph_struct(nodes, ro, ci)
    gin.col_offsets, gin.row_indices = co.ctypes.data, ri.ctypes.data
    gout = GunrockGraph()
    cfg = GunrockConfig()
    cfg.device, cfg.top_nodes = device, top_nodes
    k = max(min(nodes, top_nodes), 1)
    ids, ind, outd = (np.empty(k, dtype=np.int32) for _ in range(3))
    dt = GunrockDataType(VTXID_INT, SIZET_INT, VALUE_INT)
    lib().gunrock_topk_func(C.byref(gout), ids.ctypes.data_as(C.c_void_p), ind.ctypes.data_as(C.c_void_p), outd.ctypes.data_as(C.c_void_p),
                            C.byref(gin), cfg, dt)
    k = min(nodes, top_nodes)
    return ids[:k], ind[:k], outd[:k]


class PrProblem:
    """PRProblem + PREnactor behind the handle ABI (grx_pr_*)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().grx_pr_create(C.byref(self._h), device), "grx_pr_create")
        self.nodes = self.edges = 0

    def init(self, nodes, row_offsets, col_indices):
        ro = np.ascontiguousarray(row_offsets, dtype=np.int32)
        ci = np.ascontiguousarray(col_indices, dtype=np.int32)
        self.nodes, self.edges = int(nodes), int(ci.shape[0])
        _check(lib().grx_pr_init(self._h, self.nodes, self.edges, _p(ro), _p(ci)), "PRProblem::Init")
        return self

    def init_device(self, nodes, edges, d_row_offsets, d_col_indices):
        self.nodes, self.edges = int(nodes), int(edges)
        _check(lib().grx_pr_init_device(self._h, self.nodes, self.edges, C.c_void_p(d_row_offsets), C.c_void_p(d_col_indices)),
               "PRProblem::Init (device)")
        return self

    def set_inverse_graph(self, d_inv_row_offsets=None, d_inv_col_indices=None, build=False):
        """No arguments: the graph is symmetric (its CSR is its own inverse); build=True: transpose on the device."""
        _check(lib().grx_pr_set_inverse_graph(self._h, C.c_void_p(d_inv_row_offsets), C.c_void_p(d_inv_col_indices), int(bool(build))),
               "PRProblem::SetInverseGraph")
        return self

    def reset(self, src=-1, delta=0.85, threshold=0.01):
        _check(lib().grx_pr_reset(self._h, int(src), float(delta), float(threshold)), "PRProblem::Reset")
        return self

    def enact(self, max_iter=20, max_grid_size=0):
        ms = C.c_float()
        _check(lib().grx_pr_enact(self._h, int(max_iter), int(max_grid_size), C.byref(ms)), "PREnactor::Enact")
        return float(ms.value)

    def stats(self):
        it, pr, sv = C.c_longlong(), C.c_longlong(), C.c_longlong()
        _check(lib().grx_pr_stats(self._h, C.byref(it), C.byref(pr), C.byref(sv)), "grx_pr_stats")
        return {"iterations": int(it.value), "peeling_rounds": int(pr.value), "surviving_nodes": int(sv.value)}

    def extract(self, count=-1):
        k = self.nodes if count < 0 else min(count, self.nodes)
        ranks = np.empty(max(k, 1), dtype=np.float32)
        ids = np.empty(max(k, 1), dtype=np.int32)
        _check(lib().grx_pr_extract(self._h, ranks.ctypes.data_as(C.POINTER(C.c_float)), _p(ids), k), "PRProblem::Extract")
        return ids[:k], ranks[:k]

    def device_results(self):
        r, i = C.c_void_p(), C.c_void_p()
        _check(lib().grx_pr_device_results(self._h, C.byref(r), C.byref(i)), "grx_pr_device_results")
        return r.value, i.value

    def close(self):
        if self._h:
            lib().grx_pr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BcProblem:
    """BCProblem + BCEnactor behind the handle ABI (grx_bc_*)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().grx_bc_create(C.byref(self._h), device), "grx_bc_create")
        self.nodes = self.edges = 0
        self._keep = None

    def init(self, nodes, row_offsets, col_indices):
        ro = np.ascontiguousarray(row_offsets, dtype=np.int32)
        ci = np.ascontiguousarray(col_indices, dtype=np.int32)
        self.nodes, self.edges = int(nodes), int(ci.shape[0])
        _check(lib().grx_bc_init(self._h, self.nodes, self.edges, ro.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p)),
               "BCProblem::Init")
        return self

    def init_device(self, nodes, edges, d_row_offsets, d_col_indices):
        self.nodes, self.edges = int(nodes), int(edges)
        _check(lib().grx_bc_init_device(self._h, self.nodes, self.edges, C.c_void_p(d_row_offsets), C.c_void_p(d_col_indices)),
               "BCProblem::Init (device)")
        return self

    def run(self, src=-1, max_grid_size=0, queue_sizing=1.0):
        ms = C.c_float()
        _check(lib().grx_bc_run(self._h, int(src), max_grid_size, float(queue_sizing), C.byref(ms)), "BCEnactor::Enact")
        return float(ms.value)

    def extract(self):
        sig = np.empty(max(self.nodes, 1), dtype=np.float32)
        bc = np.empty(max(self.nodes, 1), dtype=np.float32)
        _check(lib().grx_bc_extract(self._h, sig.ctypes.data_as(C.c_void_p), bc.ctypes.data_as(C.c_void_p), None), "BCProblem::Extract")
        return sig[:self.nodes], bc[:self.nodes]

    def close(self):
        if self._h:
            lib().grx_bc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def filter_queue(ids, row_offsets=None, capacity=None, max_grid_size=0):
    """oprtr::filter::Kernel with the BFS functor over a queue of vertex ids (-1 = culled entry), on the GPU.
    row_offsets given: returns (v, row_start, scan, edges) -- a complete vertex frontier, zero-degree vertices dropped;
    else (v,).  Output order is unspecified."""
    import numpy as np
    import torch
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    n = int(ids.shape[0])
    cap = int(capacity if capacity is not None else max(n, 1))
    d_in = torch.from_numpy(ids).cuda() if n else torch.zeros(1, dtype=torch.int32, device="cuda")
    d_v = torch.empty(max(cap, 1), dtype=torch.int32, device="cuda")
    out_len, out_edges = C.c_int(), C.c_longlong()
    if row_offsets is None:
        _check(lib().grx_filter_queue(n, C.c_void_p(d_in.data_ptr()), None, cap, C.c_void_p(d_v.data_ptr()), None, None, C.byref(out_len),
                                      C.byref(out_edges), int(max_grid_size)), "filter::Kernel")
        return (d_v[:out_len.value].cpu().numpy(),)
    d_ro = torch.from_numpy(np.ascontiguousarray(row_offsets, dtype=np.int32)).cuda()
    d_rs = torch.empty_like(d_v)
    d_sc = torch.empty_like(d_v)
    _check(lib().grx_filter_queue(n, C.c_void_p(d_in.data_ptr()), C.c_void_p(d_ro.data_ptr()), cap, C.c_void_p(d_v.data_ptr()),
                                  C.c_void_p(d_rs.data_ptr()), C.c_void_p(d_sc.data_ptr()), C.byref(out_len), C.byref(out_edges),
                                  int(max_grid_size)), "filter::Kernel")
    k = out_len.value
    return d_v[:k].cpu().numpy(), d_rs[:k].cpu().numpy(), d_sc[:k].cpu().numpy(), int(out_edges.value)


class SsspProblem:
    """SSSPProblem + SSSPEnactor behind the handle C ABI."""

    def __init__(self, mark_pred=False, instrument=False, device=0):
        self._h = C.c_void_p()
        self.mark_pred = bool(mark_pred)
        _check(lib().grx_sssp_create(C.byref(self._h), int(mark_pred), int(instrument), device), "grx_sssp_create")
        self.nodes = 0

    def init(self, nodes, row_offsets, col_indices, weights, delta_factor=16):
        ro = np.ascontiguousarray(row_offsets, dtype=np.int32)
        ci = np.ascontiguousarray(col_indices, dtype=np.int32)
        w = np.ascontiguousarray(weights, dtype=np.uint32)
        self.nodes = int(nodes)
        _check(lib().grx_sssp_init(self._h, nodes, ci.shape[0], _p(ro), _p(ci), w.ctypes.data_as(C.POINTER(C.c_uint32)),
                                   delta_factor), "SSSPProblem::Init")
        return self

    def init_device(self, nodes, edges, d_row_offsets, d_col_indices, d_weights, delta):
        self.nodes = int(nodes)
        _check(lib().grx_sssp_init_device(self._h, nodes, edges, C.c_void_p(d_row_offsets), C.c_void_p(d_col_indices),
                                          C.c_void_p(d_weights), float(delta)), "SSSPProblem::Init(device)")
        return self

    def set_inverse_graph(self, d_inv_row_offsets=None, d_inv_col_indices=None, d_inv_weights=None, pull_min_edges=-1):
        """Enable pull relaxation of dense levels; no arrays = build the weighted transpose on the device."""
        _check(lib().grx_sssp_set_inverse_graph(self._h, C.c_void_p(d_inv_row_offsets), C.c_void_p(d_inv_col_indices),
                                                C.c_void_p(d_inv_weights), int(pull_min_edges)), "SSSPProblem::SetInverseGraph")
        return self

    def pull_levels(self):
        n = C.c_longlong()
        _check(lib().grx_sssp_pull_levels(self._h, C.byref(n)), "grx_sssp_pull_levels")
        return int(n.value)

    def reset(self, src, queue_sizing=1.0):
        _check(lib().grx_sssp_reset(self._h, int(src), float(queue_sizing)), "SSSPProblem::Reset")

    def enact(self, src, max_grid_size=0):
        ms = C.c_float()
        _check(lib().grx_sssp_enact(self._h, int(src), max_grid_size, C.byref(ms)), "SSSPEnactor::Enact")
        return float(ms.value)

    def stats(self):
        v, e, it, l = C.c_longlong(), C.c_longlong(), C.c_longlong(), C.c_longlong()
        k, d = C.c_double(), C.c_float()
        _check(lib().grx_sssp_stats(self._h, C.byref(v), C.byref(e), C.byref(it), C.byref(l), C.byref(k), C.byref(d)),
               "grx_sssp_stats")
        return {"relaxed_vertices": v.value, "relaxed_edges": e.value, "iterations": it.value,
                "kernel_launches": l.value, "kernel_ms": k.value, "delta": d.value}

    def extract(self):
        dist = np.empty(max(self.nodes, 1), dtype=np.uint32)
        preds = np.empty(max(self.nodes, 1), dtype=np.int32) if self.mark_pred else None
        _check(lib().grx_sssp_extract(self._h, dist.ctypes.data_as(C.POINTER(C.c_uint32)),
                                      None if preds is None else _p(preds)), "SSSPProblem::Extract")
        return dist[:self.nodes], (None if preds is None else preds[:self.nodes])

    def close(self):
        if self._h:
            lib().grx_sssp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gunrock_sssp(nodes, row_offsets, col_indices, weights, src=0, mark_pred=True, delta_factor=16, queue_size=1.0,
                 src_mode=SRC_MANUALLY, device=0):
    """Call gunrock_sssp_func as reference shared_lib_tests/test_sssp.c does; returns (distances, predecessors)."""
    ro = np.ascontiguousarray(row_offsets, dtype=np.int32)
    ci = np.ascontiguousarray(col_indices, dtype=np.int32)
    w = np.ascontiguousarray(weights, dtype=np.uint32)
    gin = _graph_struct(nodes, ro, ci, w)
    gout = GunrockGraph()
    cfg = GunrockConfig()
    cfg.mark_pred, cfg.src_node, cfg.device = mark_pred, src, device
    cfg.delta_factor, cfg.queue_size, cfg.src_mode = delta_factor, queue_size, src_mode
    preds = np.empty(max(nodes, 1), dtype=np.int32)
    dt = GunrockDataType(VTXID_INT, SIZET_INT, VALUE_UINT)
    lib().gunrock_sssp_func(C.byref(gout), preds.ctypes.data_as(C.c_void_p), C.byref(gin), cfg, dt)
    return _take_node_values(gout, nodes, np.uint32), preds[:nodes]
