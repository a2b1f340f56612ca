"""Device-resident synthetic graphs for benchmarks and full-size tests.

The reference builds every graph on the host with a serial rand() generator and std::stable_sort
(reference gunrock/graphio/rmat.cuh:27-91, csr.cuh:247-340) -- minutes at scale-24.  Here the seeded R-MAT
tuples come from the library's HIP generator (grx_rmat_seeded_device) and the COO -> CSR step
(sort by (row, col), drop self loops and duplicates: the same graph Csr::FromCoo would give, all values 1)
is the library's hand-written device radix sort + scan (grx_coo_to_csr_*).
torch is used for device memory only; no sorting or traversal work happens here.
"""
import numpy as np
import torch

from . import capi


class DevArray:
    """Wrap a raw device pointer so torch can view it (no copy)."""

    def __init__(self, ptr, n, typestr="<i4"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


def as_tensor(ptr, n, typestr="<i4", device="cuda"):
    if n == 0 or not ptr:
        return torch.empty(0, dtype=torch.int32, device=device)
    return torch.as_tensor(DevArray(ptr, n, typestr), device=device)


def rmat_tuples_device(scale, pairs, seed=0x6772, a=0.55, b=0.2, c=0.2, d=0.05, first=0, device="cuda"):
    rows = torch.empty(pairs, dtype=torch.int32, device=device)
    cols = torch.empty(pairs, dtype=torch.int32, device=device)
    torch.cuda.synchronize()
    rc = capi.lib().grx_rmat_seeded_device(scale, first, pairs, seed, a, b, c, d, rows.data_ptr(), cols.data_ptr(), None)
    if rc != 0:
        raise RuntimeError("grx_rmat_seeded_device failed (%d)" % rc)
    torch.cuda.synchronize()
    return rows, cols


def csr_from_tuples_device(nodes, rows, cols, undirected=True, parts=1, rank=0):
    """(row, col) int32 tuples on the GPU -> CSR on the GPU with Csr::FromCoo's graph semantics (values all 1), through the
    library's hand-written radix sort / scan (grx_coo_to_csr_*; gunrock/graphio/device_csr.hpp).

    parts > 1 builds rank's slice of the vertex-cut partition (local rows, global columns).
    Returns int32 tensors (row_offsets[rows+1], col_indices[m]); torch only allocates the output arrays.
    """
    import ctypes as C
    rows = rows.int().contiguous()
    cols = cols.int().contiguous()
    pairs = int(rows.shape[0])
    n_rows = nodes if parts == 1 else ((nodes - rank + parts - 1) // parts if nodes > rank else 0)
    torch.cuda.synchronize()
    h, edges = C.c_void_p(), C.c_longlong()
    rc = capi.lib().grx_coo_to_csr_sort(C.byref(h), n_rows, nodes, pairs, C.c_void_p(rows.data_ptr() if pairs else None),
                                        C.c_void_p(cols.data_ptr() if pairs else None), int(bool(undirected)), parts, rank,
                                        C.byref(edges), None)
    if rc == -2:
        raise ValueError("graph exceeds the SIZET_INT contract of the C ABI")
    if rc != 0:
        raise RuntimeError("grx_coo_to_csr_sort failed (%d)" % rc)
    try:
        ro = torch.empty(n_rows + 1, dtype=torch.int32, device=rows.device)
        ci = torch.empty(int(edges.value), dtype=torch.int32, device=rows.device)
        torch.cuda.synchronize()
        rc = capi.lib().grx_coo_to_csr_emit(h, C.c_void_p(ro.data_ptr()), C.c_void_p(ci.data_ptr() if edges.value else None), None)
        if rc != 0:
            raise RuntimeError("grx_coo_to_csr_emit failed (%d)" % rc)
    finally:
        capi.lib().grx_coo_to_csr_free(h)
    return ro, ci


def rmat_csr_device(scale, edge_factor=8, seed=0x6772, undirected=True, device="cuda"):
    """SURVEY 8(d) benchmark graph: 2^scale vertices, edge_factor * 2^scale generated pairs, mirrored."""
    pairs = edge_factor << scale
    rows, cols = rmat_tuples_device(scale, pairs, seed, device=device)
    ro, ci = csr_from_tuples_device(1 << scale, rows, cols, undirected)
    return ro, ci


def largest_degree_source(row_offsets):
    """Csr::GetNodeWithHighestDegree on a device CSR: FIRST vertex of maximal degree (csr.cuh:442-455)."""
    deg = row_offsets[1:] - row_offsets[:-1]
    m = deg.max()
    return int(torch.nonzero(deg == m)[0]), int(m)


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


def seeded_sources(row_offsets, count, seed=0x6772):
    """`count` pseudo-random vertices with degree > 0 from splitmix64(seed)."""
    n = row_offsets.shape[0] - 1
    deg = (row_offsets[1:] - row_offsets[:-1])
    out = []
    x = seed
    guard = 0
    while len(out) < count and guard < 1000 * (count + 1):
        x = _splitmix64(x)
        v = x % n
        guard += 1
        if int(deg[v]) > 0:
            out.append(int(v))
    return out


def to_host_csr(row_offsets, col_indices):
    return (row_offsets.cpu().numpy().astype(np.int32, copy=False),
            col_indices.cpu().numpy().astype(np.int32, copy=False))


def grid_csr_device(side, shortcut_fraction=0.01, seed=0x6772, device="cuda"):
    """Road-like stand-in of SURVEY 8(d): side x side 4-neighbour grid plus `shortcut_fraction` * n random shortcuts,
    undirected.  Average degree ~4, so the reference driver would pick traversal_mode 1 (test_bfs.cu:563-566)."""
    n = side * side
    v = torch.arange(n, device=device, dtype=torch.int64)
    x, y = v % side, v // side
    right = v[x < side - 1]
    down = v[y < side - 1]
    rows = torch.cat([right, down])
    cols = torch.cat([right + 1, down + side])
    k = int(n * shortcut_fraction)
    if k > 0:
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        rows = torch.cat([rows, torch.randint(0, n, (k,), generator=gen, device=device)])
        cols = torch.cat([cols, torch.randint(0, n, (k,), generator=gen, device=device)])
    return csr_from_tuples_device(n, rows.int(), cols.int(), undirected=True)
