"""The fused kernels hand out problems in two ways: static rounds (small and mid-size launches: no ticket, slot-major wave order) and tickets from
a device counter in guided chunks (large launches).  The suite's own batches are small, i.e. static by default -- so the parity tests of the fused
kernels are run ONCE more here, in one child process, with the plan flag MO_PLAN_TICKETS_ALWAYS on every plan, and a second time with
MO_PLAN_STATIC_ROUNDS_ALWAYS (also the full-size launches).  The flags reach the plans through the Python mirror's MO_PLAN_EXTRA_FLAGS (a knob of
the mirror: the library itself reads no environment variable)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_subset(flags: int, select: str):
    env = dict(os.environ, MO_PLAN_EXTRA_FLAGS=str(flags))
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_gpu_fullsize.py"),
           "-x", "-q", "-m", "gpu", "-k", select, "-p", "no:cacheprovider"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, (res.stdout[-3000:], res.stderr[-2000:])
    assert " passed" in res.stdout and "failed" not in res.stdout, res.stdout[-2000:]


def test_fused_parity_with_tickets_only():
    _run_subset(4, "fused or fullsize or batched or golden")   # MO_PLAN_TICKETS_ALWAYS


def test_fused_parity_with_static_rounds_only():
    _run_subset(8, "fused or fullsize or batched or golden")   # MO_PLAN_STATIC_ROUNDS_ALWAYS
