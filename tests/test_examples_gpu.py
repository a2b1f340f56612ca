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
])
def test_c_program_known_answer(name, regex):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), name, "-s"])
    out = subprocess.run([os.path.join(ROOT, "examples", name)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert re.search(regex, out.stdout), out.stdout
