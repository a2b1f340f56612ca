"""gunrockinst_amd -- MI355X-native frontier engine (BFS / CC / SSSP / BC / PageRank / TopK) behind Gunrock's C ABI.

The product is the shared library ``gunrockinst_amd/lib/libgunrock.so`` (hand-written HIP for gfx950,
built by ``gunrockinst_amd/csrc/Makefile``).  This package is only the host-side binding: ctypes
mirrors of ``include/gunrock/gunrock.h`` and ``include/gunrock/gunrock_mi355x.h``.  There is no
CPU fallback: importing the binding without the built library raises.
"""
from .capi import (  # noqa: F401
    GunrockConfig, GunrockDataType, GunrockGraph, LIB_PATH, lib, build_library,
    VTXID_INT, SIZET_INT, VALUE_INT, VALUE_UINT, VALUE_FLOAT, SRC_MANUALLY, SRC_RANDOMIZE, SRC_LARGEST_DEGREE,
    HostGraph, BfsProblem, CcProblem, SsspProblem, BcProblem, PrProblem, gunrock_bfs, gunrock_cc, gunrock_sssp, gunrock_bc,
    gunrock_pr, gunrock_topk, version, filter_queue,
)

__all__ = [
    "GunrockConfig", "GunrockDataType", "GunrockGraph", "LIB_PATH", "lib", "build_library",
    "HostGraph", "BfsProblem", "CcProblem", "SsspProblem", "BcProblem", "gunrock_bfs", "gunrock_cc", "gunrock_sssp",
    "gunrock_bc", "PrProblem", "gunrock_pr", "gunrock_topk", "version", "filter_queue",
]
