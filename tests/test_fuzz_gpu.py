"""Randomised parity sweeps (tools/fuzz_*.py) as part of the GPU suite: fixed seeds, a bounded time budget each.

The sweeps draw random graphs (R-MAT of random scale / edge factor, directed and mirrored; grids with shortcuts; stars; chains;
sparse forests; hub graphs), random sources, modes and tuning knobs, and compare every result with the CPU oracle: BFS labels
bit-exact and parents valid in traversal modes 0 / 1 / 2 (directed inputs through the device-built inverse graph), SSSP distances
bit-exact, CC labels bit-exact, BC within 1e-3, PageRank within 2e-4 of the (unpinned) restatement, and the in-library partitioned
BFS over gloo with ranks sharing the GPU.  Round 2's sweep found the BC sink bug (60d4428); running them here keeps that net under
every round's GPU test run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUDGET_S = os.environ.get("GUNROCK_FUZZ_SECONDS", "15")  # (per sweep; the whole GPU suite is kept near six minutes)


def _run(script, *args, timeout=240):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + [str(a) for a in args], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=timeout)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0 and "fuzz ok:" in r.stdout, tail
    return r.stdout


def test_fuzz_bfs_schedules():
    out = _run("fuzz_bfs.py", BUDGET_S, 20261004)
    assert int(out.split("fuzz ok:")[1].split()[0]) >= 50, out[-500:]


def test_fuzz_sssp_cc_bc_pagerank():
    out = _run("fuzz_others.py", BUDGET_S, 20261005)
    assert int(out.split("fuzz ok:")[1].split()[0]) >= 20, out[-500:]


def test_fuzz_partitioned_bfs_over_gloo():
    _run("fuzz_pbfs.py", 3, BUDGET_S, 20261006)
