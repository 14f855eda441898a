#!/usr/bin/env python3
"""Dump the direction of one mo_newton_step launch (named config, small batch) to a .npy file: A/B builds of the library (MO_LIB_PATH) are
compared bit for bit by running this once per build.  usage: dump_step.py cfg4 256 out.npy"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from mini_opt_amd import qp as Q
from mini_opt_amd import synth

cfg, batch, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
d = synth.CONFIGS[cfg]
dt = torch.float64 if d["dtype"] == "f64" else torch.float32
prob, vars_, mu = synth.make_batch_torch(d["n"], d["k"], d["m"], d["m_r"], batch, torch.device("cuda:0"), dt)
s = Q.QPInteriorPointSolver(prob)
s.SetVariables(vars_)
delta, alpha, status = s.NewtonStep(mu, 0.995)
torch.cuda.synchronize()
np.save(out, np.concatenate([delta.cpu().numpy().ravel(), alpha.cpu().numpy().ravel(), status.cpu().numpy().ravel().astype(delta.cpu().numpy().dtype)]))
print(s.step_kernel(), "status ok", int((status == 0).sum()), "of", batch)
