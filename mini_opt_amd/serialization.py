"""JSON form of NLSSolverOutputs in the reference's own schema (source/serialization.cc:32-136, nlohmann::json): one object per problem
of a batched solve, with the reference's keys and enum strings, so that tooling written against mini_opt's serialized logs reads ours.

    {"termination_state": "SATISFIED_ABSOLUTE_TOL",
     "iterations": [{"iteration": 0, "optimizer_state": "NOMINAL", "lambda": ..., "errors_initial": {"f": ..., "equality": ...},
                     "qp_outputs": {"termination_state": ..., "iterations": [{"kkt_initial": {...}, "kkt_final": {...}, "ip_outputs": {...}}],
                                    "lagrange_multipliers": null | {"min": ..., "l_infinity": ...}}   -- or "SUCCESS" / "NOT_POSITIVE_DEFINITE"
                                                                                                         for the null-space solver (:89-101)
                     "qp_eigenvalues": null | {"min": ..., "max": ..., "abs_min": ...}, "directional_derivatives": {"d_f": ..., "d_equality": ...}, "penalty": ...,
                     "step_result": "SUCCESS", "line_search_steps": [{"alpha": ..., "errors": {"f": ..., "equality": ...}}]}]}

The per-iteration QP records exist only when the solve was run with record_qp_iterations=True; otherwise "iterations" of qp_outputs is
empty and the count is given as "num_iterations" (an extra key).  NaN becomes null, as nlohmann::json writes it.  MO_NLS_QP_FAILURE
(the reference throws there) is written as "QP_FAILURE".
"""
from __future__ import annotations

import json
import math
from typing import Any, Dict, List

OPTIMIZER_STATE = ["NOMINAL", "ATTEMPTING_RESTORE_LM"]                                                 # serialization.cc:32-34
STEP_RESULT = ["SUCCESS", "MAX_ITERATIONS", "FIRST_ORDER_SATISFIED", "POSITIVE_DERIVATIVE", "FAILURE_NON_FINITE_COST",
               "FAILURE_INVALID_ALPHA"]                                                                # :36-43
NLS_TERMINATION = ["MAX_ITERATIONS", "SATISFIED_ABSOLUTE_TOL", "SATISFIED_RELATIVE_TOL", "SATISFIED_FIRST_ORDER_TOL", "MAX_LAMBDA",
                   "QP_INDEFINITE", "USER_CALLBACK", "QP_FAILURE"]                                     # :45-53
QP_TERMINATION = ["SATISFIED_KKT_TOL", "MAX_ITERATIONS"]                                               # :55-58
NULLSPACE_TERMINATION = ["SUCCESS", "NOT_POSITIVE_DEFINITE"]                                           # :60-63


def _num(v: float):
    v = float(v)
    return v if math.isfinite(v) else None


def _kkt(r) -> Dict[str, Any]:                                                                          # KKTError, :73
    return {"r_dual": _num(r[0]), "r_comp": _num(r[1]), "r_primal_eq": _num(r[2]), "r_primal_ineq": _num(r[3])}


def _qp_iteration(r) -> Dict[str, Any]:                                                                 # QPInteriorPointIteration, :74
    return {"kkt_initial": _kkt(r[0:4]), "kkt_final": _kkt(r[4:8]),
            "ip_outputs": {"mu": _num(r[8]), "alpha": {"primal": _num(r[9]), "dual": _num(r[10])},
                           "alpha_probe": {"primal": _num(r[11]), "dual": _num(r[12])}, "mu_affine": _num(r[13])}}   # :71-72


def nls_outputs_to_json(outputs, problem: int) -> Dict[str, Any]:
    """NLSSolverOutputs of problem `problem` of a batched solve (mini_opt_amd.nls.NLSSolverOutputs) as the reference's JSON object."""
    from . import _lib as L
    p = int(problem)
    term = int(outputs.termination_state[p])
    recs = outputs.iterations[p].cpu().numpy()
    qp_recs = None if outputs.qp_iterations is None else outputs.qp_iterations[:, p].cpu().numpy()
    qp_lag = None if outputs.qp_lagrange is None else outputs.qp_lagrange[:, p].cpu().numpy()
    qp_eig = None if getattr(outputs, "qp_eigenvalues", None) is None else outputs.qp_eigenvalues[:, p].cpu().numpy()
    its: List[Dict[str, Any]] = []
    for i in range(int(outputs.num_iterations[p])):
        r = recs[i]
        if math.isnan(r[1]):   # QP_INDEFINITE ends a problem without logging the iteration (nonlinear.cc:103-105)
            break
        nsteps = 0 if math.isnan(r[8]) else int(r[8])
        steps = [{"alpha": _num(r[L.MO_NLS_ITER_HEADER + 3 * s]),
                  "errors": {"f": _num(r[L.MO_NLS_ITER_HEADER + 3 * s + 1]), "equality": _num(r[L.MO_NLS_ITER_HEADER + 3 * s + 2])}}
                 for s in range(nsteps)]
        if outputs.null_space_path:
            qp_out: Any = NULLSPACE_TERMINATION[0 if int(r[11]) == 0 else 1]
        else:
            nqp = 0 if math.isnan(r[10]) else int(r[10])
            qp_out = {"termination_state": QP_TERMINATION[0 if math.isnan(r[9]) else int(r[9])],
                      "iterations": [] if qp_recs is None else [_qp_iteration(qp_recs[i][q]) for q in range(nqp)],
                      "lagrange_multipliers": None if qp_lag is None or math.isnan(qp_lag[i][0]) else
                      {"min": _num(qp_lag[i][0]), "l_infinity": _num(qp_lag[i][1])}}
            if qp_recs is None:
                qp_out["num_iterations"] = nqp
        its.append({"iteration": i, "optimizer_state": OPTIMIZER_STATE[int(r[0])], "lambda": _num(r[1]),
                    "errors_initial": {"f": _num(r[2]), "equality": _num(r[3])}, "qp_outputs": qp_out,
                    "qp_eigenvalues": None if qp_eig is None or math.isnan(qp_eig[i][0]) else       # std::optional<QPEigenvalues>, serialization.cc:66
                    {"min": _num(qp_eig[i][0]), "max": _num(qp_eig[i][1]), "abs_min": _num(qp_eig[i][2])},
                    "directional_derivatives": {"d_f": _num(r[4]), "d_equality": _num(r[5])}, "penalty": _num(r[6]),
                    "step_result": STEP_RESULT[0 if math.isnan(r[7]) else int(r[7])], "line_search_steps": steps})
    return {"termination_state": NLS_TERMINATION[term], "iterations": its}


def dumps(outputs, problem: int, **kw) -> str:
    return json.dumps(nls_outputs_to_json(outputs, problem), **kw)
