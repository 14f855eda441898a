"""Runs the C++ facade test binary (tests/cpp/facade_test.cpp: the reference's qp_test.cc cases re-stated against
mini_opt_amd/cpp/mini_opt_hip.hpp) on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_cpp_facade():
    exe = os.path.join(ROOT, "tests", "cpp", "facade_test")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "mini_opt_amd", "lib") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "all tests passed" in res.stdout
