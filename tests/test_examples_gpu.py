"""C programs written against include/gunrock/gunrock.h the way the reference's shared_lib_tests are: compile with gcc,
link libgunrock.so, run on the GPU, check the ctest-style known answers (reference CMakeLists.txt:213-229)."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name,regex", [
    ("test_bfs", r"Node_ID.*2.*: Label.*1"),
    ("test_cc", r"Node_ID.*1.*: Component_ID.*0"),
    ("test_sssp", r"Node ID.*1.*: Label.*39.*: Predecessor.*0"),
    ("test_bc", r"Node_ID.*0.*: BC.*0.500000"),
    ("test_topk", r"Node ID.*2.*: in_degrees.*3.*: out_degrees.*3"),     # reference ctest TestTopK (CMakeLists.txt:235-237)
    ("test_pr", r"Node ID \[2\] : Page Rank \[0\.3575"),                   # the oracle's value; the reference's regex is stale
])
def test_c_program_known_answer(name, regex):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), name, "-s"])
    out = subprocess.run([os.path.join(ROOT, "examples", name)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert re.search(regex, out.stdout), out.stdout


def _run(cmd, timeout=300):
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)


def test_simple_example_prints_test_passed():
    # reference ctest "SimpleExample" (CMakeLists.txt:239-241): CC + BFS self-check on bips98_606.mtx ends with TEST PASSED
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "simple_example", "-s"])
    out = _run([os.path.join(ROOT, "examples", "simple_example"), "market", os.path.join(ROOT, "tests", "golden", "bips98_606.mtx")])
    assert out.returncode == 0, out.stdout + out.stderr
    assert "TEST PASSED" in out.stdout and "Label Validity: \nCORRECT" in out.stdout
    assert "Validity BC Value: \nCORRECT" in out.stdout                # third stage of the reference's example (simple_example.cu:592-672)
    assert "CPU components: 542, GPU components: 542" in out.stdout
    assert "7135 nodes, 30380 edges" in out.stdout


@pytest.mark.parametrize("flags", [
    ["--undirected", "--src=largestdegree", "--quick=0"],
    ["--undirected", "--src=0", "--quick=0", "--mark-pred", "--idempotence=0", "--traversal-mode=0"],
    ["--undirected", "--src=566", "--quick=0", "--mark-pred", "--traversal-mode=2", "--iteration-num=3", "--instrumented"],
    ["--src=randomize", "--quick=0", "--idempotence=0"],
])
def test_test_bfs_cli_flags(flags):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "test_bfs_cli", "-s"])
    out = _run([os.path.join(ROOT, "examples", "test_bfs_cli"), "market", os.path.join(ROOT, "tests", "golden", "bips98_606.mtx")] + flags)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Label Validity: \nCORRECT" in out.stdout and "MiEdges/s" in out.stdout
    if "--mark-pred" in flags:
        assert "Predecessor Validity: CORRECT" in out.stdout
    if "--src=largestdegree" in flags:
        assert "Using highest degree (111) vertex: 566" in out.stdout


def test_test_bfs_cli_rmat_matches_reference_graph_size():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "test_bfs_cli", "-s"])
    out = _run([os.path.join(ROOT, "examples", "test_bfs_cli"), "rmat", "--quick=0", "--src=largestdegree"])
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Graph: 1024 nodes, 997 edges" in out.stdout              # libc R-MAT golden
    assert "Label Validity: \nCORRECT" in out.stdout


@pytest.mark.parametrize("flags", [["--undirected", "--src=566"], ["--src=0"], ["--undirected", "--src=7134"]])
def test_user_functor_with_only_the_reference_methods(flags):
    # a user-written problem + functor with ONLY CondEdge / ApplyEdge / CondFilter / ApplyFilter (the reference's
    # non-idempotent atomicCAS claim, bfs_functor.cuh:49-117, restated) drives advance::LaunchKernel + filter::LaunchKernel level
    # by level like codesnaps/bfs/bfs_enactor.cuh:41-62 -- no engine hook defined -- and matches the in-program CPU BFS.
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "user_functor", "-s"])
    src = open(os.path.join(ROOT, "examples", "user_functor.hip")).read()
    code = src.split("struct DepthFunctor {")[1].split("};")[0]
    for hook in ("ScreenEdge", "IssueEdge", "ResolveEdge", "SourceData", "ApplyEdgeWave", "ReduceValue", "IssueFilter"):
        assert hook not in code
    out = _run([os.path.join(ROOT, "examples", "user_functor"), "market", os.path.join(ROOT, "tests", "golden", "bips98_606.mtx")] + flags)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Label Validity: \nCORRECT" in out.stdout and "TEST PASSED" in out.stdout
    m = re.search(r"search depth: GPU (\d+), CPU (\d+)", out.stdout)
    assert m and m.group(1) == m.group(2), out.stdout
    if flags == ["--undirected", "--src=566"]:
        assert int(m.group(1)) == 18                                  # 19 levels from vertex 566 = deepest label 18
