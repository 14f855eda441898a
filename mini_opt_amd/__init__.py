"""mini_opt_amd -- MI355X-native batched interior-point Newton steps behind mini_opt's QP API.

Only what the hot path needs: csrc/ (HIP kernels + C ABI, built into lib/libminiopt_hip.so), qp.py (host mirror of
mini_opt::QP / QPInteriorPointSolver for batches), synth.py (synthetic workloads), sharding.py (multi-GPU batch shards).
"""
from . import _lib  # noqa: F401
from ._lib import MiniOptError, build  # noqa: F401
