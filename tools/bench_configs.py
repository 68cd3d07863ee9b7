"""Throughput of the other BASELINE configs (parity-test cases, not the headline bench line).

python tools/bench_configs.py robertson 262144 | seir 16384 | lv 65536 | network100 4096
Host-memory API (solve_forward_batch / solve_backward_batch); prints kernel times from HIP events.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sunode_amd import SympyProblem  # noqa: E402
from sunode_amd.solver import AdjointSolver  # noqa: E402
from tools.problems import PROBLEMS, lv_batch, network100, network_batch, robertson_batch, seir_batch  # noqa: E402


def main():
    name = sys.argv[1]
    B = int(sys.argv[2])
    s = network100() if name == "network100" else PROBLEMS[name]
    prob = SympyProblem(s["params"], s["states"], s["rhs"], s["derivative_params"])
    if name == "lv":
        d = lv_batch(B)
        ps, pr = d["params"][:, :2], d["params"][:, 2:]
        rt, at, cap = 1e-8, 1e-8, 512
    elif name == "robertson":
        d = robertson_batch(B)
        ps, pr = d["params"], np.zeros(0)
        rt, at, cap = 1e-8, 1e-10, 2048
    elif name == "network100":
        d = network_batch(B)
        ps, pr = d["ps"], d["pr"]
        rt, at, cap = 1e-8, 1e-8, 1024
    else:
        d = seir_batch(B)
        ps, pr = d["ps"], d["pr"]
        rt, at, cap = 1e-8, 1e-8, 1024
    tv = d["tvals"]
    n = prob.n_states
    k = np.arange(len(tv))[:, None]; i = np.arange(n)[None, :]
    grads = np.ones((len(tv), n)) if name == "lv" else 1.0 + 0.5 * np.cos(1.7 * k + 0.9 * i)
    sol = AdjointSolver(prob, abstol=at, reltol=rt, backward_abstol=at, backward_reltol=rt, quad_abstol=at,
                        quad_reltol=rt, max_steps=cap)
    for rep in range(2):
        t0 = time.perf_counter()
        y, st, stats = sol.solve_forward_batch(0.0, tv, d["y0"], ps, pr)
        g, lam, stb, statsb = sol.solve_backward_batch(tv[-1], 0.0, tv, grads)
        wall = time.perf_counter() - t0
        f, b = sol._engine().last_kernel_ms()
    if "--cpu" in sys.argv:
        # the CPU oracle on the box's host cores, same workload, bounded sample (the checker timed as a baseline)
        from oracle.harness import Oracle
        from bench import usable_cores
        nc = usable_cores()
        ns = {"lv": 65536, "robertson": 16384, "seir": 2048, "network100": 64}[name]
        ds = {"lv": lv_batch, "robertson": robertson_batch, "seir": seir_batch, "network100": network_batch}[name](ns)
        ps_s, pr_s = (ds["params"][:, :2], ds["params"][:, 2:]) if name == "lv" else \
            ((ds["params"], np.zeros(0)) if name == "robertson" else (ds["ps"], ds["pr"]))
        orc = Oracle(prob, tag=name, opt="-O1" if name == "network100" else "-O2")
        cfg = orc.config(rtol=rt, atol=at, rtolB=rt, atolB=at, rtolQB=rt, atolQB=at)
        t1 = time.perf_counter()
        orc.solve_forward(cfg, ds["y0"], ps_s, pr_s, 0.0, tv, nthreads=nc)
        orc.solve_backward(cfg, tv[-1], 0.0, tv, grads, nthreads=nc)
        cpu = time.perf_counter() - t1
        print("%s CPU oracle: %d instances in %.2f s on %d cores -> %.3g solves/s" % (name, ns, cpu, nc, ns / cpu))
    print("%s B=%d: fwd kernel %.2f ms, bwd kernel %.2f ms, wall (host arrays) %.1f ms -> %.3g solves/s (kernels), "
          "failed %d/%d, fwd steps %.0f, bwd steps %.0f, bwd wave-iters %.0f"
          % (name, B, f, b, 1e3 * wall, B / ((f + b) * 1e-3), int((st != 0).sum()), int((stb != 0).sum()),
             stats[:, 0].mean(), statsb[:, 0].mean(), statsb[:, 15].mean()))


if __name__ == "__main__":
    main()
