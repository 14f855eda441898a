"""PCIe-inclusive rate of the headline step: inputs start in PINNED HOST memory, the timed region copies them to the
device, launches mo_newton_step and copies delta / alpha / status back.  Not the contract's `value` (bench.py times resident inputs); this
is the figure a caller sees who keeps its QPs on the host.  Two variants: one copy + one launch + one copy back, and the batch cut into
chunks whose copies overlap the previous chunk's kernel on a second stream.
usage: python tools/bench_pcie.py [--batch 16384] [--chunks 8] [--reps 5]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mini_opt_amd import qp as Q
from mini_opt_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--chunks", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    d = synth.CONFIGS["cfg3"]
    n, k, m, m_r = d["n"], d["k"], d["m"], d["m_r"]
    prob, vars_, mu = synth.make_batch_torch(n, k, m, m_r, args.batch, dev, torch.float64)
    names = ["J", "r", "A_eq", "b_eq", "cons_var", "cons_a", "cons_b"]
    host = {nm: getattr(prob, nm).cpu().pin_memory() for nm in names}
    host["vars"] = vars_.cpu().pin_memory(); host["mu"] = mu.cpu().pin_memory()
    bytes_in = sum(t.numel() * t.element_size() for t in host.values())
    V = vars_.shape[1]
    out_host = {"delta": torch.empty(args.batch, V, dtype=torch.float64).pin_memory(), "alpha": torch.empty(args.batch, 2, dtype=torch.float64).pin_memory(),
                "status": torch.empty(args.batch, dtype=torch.int32).pin_memory()}
    bytes_out = sum(t.numel() * t.element_size() for t in out_host.values())

    def run(chunks):
        B = args.batch
        edges = [B * c // chunks for c in range(chunks + 1)]
        streams = [torch.cuda.Stream() for _ in range(2)]
        devbuf = [{nm: torch.empty_like(t[: edges[1] - edges[0] + 1], device=dev) for nm, t in host.items()} for _ in range(2)]
        solvers = [None, None]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in range(chunks):
            lo, hi = edges[c], edges[c + 1]
            sidx = c & 1
            with torch.cuda.stream(streams[sidx]):
                buf = {nm: devbuf[sidx][nm][: hi - lo] for nm in host}
                for nm, t in host.items():
                    buf[nm].copy_(t[lo:hi], non_blocking=True)
                p = Q.BatchedQP(n=n, k=k, m=m, J=buf["J"], r=buf["r"], lam=prob.lam, A_eq=buf["A_eq"], b_eq=buf["b_eq"], cons_var=buf["cons_var"],
                                cons_a=buf["cons_a"], cons_b=buf["cons_b"])
                s = Q.QPInteriorPointSolver(p)
                s.SetVariables(buf["vars"])
                delta, alpha, status = s.NewtonStep(buf["mu"], 0.995)
                out_host["delta"][lo:hi].copy_(delta, non_blocking=True)
                out_host["alpha"][lo:hi].copy_(alpha, non_blocking=True)
                out_host["status"][lo:hi].copy_(status, non_blocking=True)
                solvers[sidx] = s   # keep the plan alive until the stream has drained
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    res = {}
    for label, chunks in (("one_shot", 1), ("pipelined", args.chunks)):
        run(chunks)  # warm-up (plan creation, page mapping)
        ts = sorted(run(chunks) for _ in range(args.reps))
        t = ts[len(ts) // 2]
        res[label] = {"chunks": chunks, "seconds": t, "steps_per_s": args.batch / t, "host_to_device_GBps": bytes_in / t / 1e9}
    assert int((out_host["status"] != 0).sum()) == 0
    print(json.dumps({"workload": f"cfg3 shape, batch {args.batch}, inputs in pinned host memory", "bytes_in_per_step": bytes_in / args.batch,
                      "bytes_out_per_step": bytes_out / args.batch, **res}))


if __name__ == "__main__":
    main()
