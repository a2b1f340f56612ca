"""The C-ABI library loads and exports every symbol include/gunrock/*.h declares (no compute calls)."""
import ctypes
import os
import re

import gunrockinst_amd as ga
from gunrockinst_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for h in ("gunrock.h", "gunrock_mi355x.h"):
        text = open(os.path.join(ROOT, "include", "gunrock", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b((?:gunrock|grx)_[a-z0-9_]+)\s*\(", text))
    return names


def test_every_declared_symbol_is_exported_and_bound():
    L = ga.lib()
    declared = _declared()
    assert declared, "header parse found nothing"
    for name in sorted(declared):
        assert hasattr(L, name), "libgunrock.so does not export %s" % name
    assert declared == set(capi.exported_symbols()), declared ^ set(capi.exported_symbols())


def test_struct_layout_matches_reference_header():
    # x86-64 SysV layout of the structs in reference gunrock/gunrock.h:51-99
    assert ctypes.sizeof(capi.GunrockDataType) == 12
    assert ctypes.sizeof(capi.GunrockGraph) == 64
    assert capi.GunrockGraph.node_values.offset == 48 and capi.GunrockGraph.edge_values.offset == 56
    assert ctypes.sizeof(capi.GunrockConfig) == 40
    assert capi.GunrockConfig.src_node.offset == 4 and capi.GunrockConfig.src_mode.offset == 36
    assert capi.GunrockConfig.queue_size.offset == 32


def test_c_program_compiles_against_header(tmp_path):
    # a C99 translation unit written like the reference's shared_lib_tests links against the library
    import subprocess
    src = tmp_path / "t.c"
    src.write_text(
        "#include <stdio.h>\n#include <gunrock/gunrock.h>\n"
        "int main(void){struct GunrockConfig c; c.src_mode = largest_degree; struct GunrockDataType d;"
        " d.VALUE_TYPE = VALUE_UINT; (void)c; (void)d;"
        " printf(\"%p %p %p\\n\", (void*)gunrock_bfs_func, (void*)gunrock_cc_func, (void*)gunrock_sssp_func); return 0;}\n")
    exe = tmp_path / "t"
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lgunrock", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])


def test_version_string():
    assert "gfx950" in ga.version()
