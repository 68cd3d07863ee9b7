"""bench.py -- headline metric of BASELINE.json on MI355X.

metric : fp64 forward+adjoint ODE solves/sec at rtol=1e-8 (Lotka-Volterra, config 2:
         batch 65 536 random parameter draws per GPU, 2 differentiated parameters, 50 outputs,
         rtol=atol=1e-8 for the forward, backward and quadrature problems, grads = ones)
step   : one forward (sa_solve_forward_batch) + one adjoint (sa_solve_backward_batch) pass over the
         whole batch, inputs and outputs resident in HBM (torch tensors, SA_MEM_DEVICE).
N > 1  : one process per GPU (torch.distributed / RCCL only for the barrier and the max-reduce of
         the elapsed time); the batch shards by instance, no data-path collective -> weak scaling.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     : dominant kernel (sa_k_backward) algorithmic HBM bytes / HIP-event duration vs 8 TB/s
  cpu_baseline : the CPU oracle (oracle/cvodes_oracle.c, "port") timed on the host cores, bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def make_problem():
    from sunode_amd import SympyProblem
    from tools.problems import PROBLEMS
    s = PROBLEMS["lv"]
    return SympyProblem(s["params"], s["states"], s["rhs"], s["derivative_params"])


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    this same command (profiles/pmc_traffic.json, collected as MI355X_MICROARCH.md prescribes: separate
    passes, KiB units, calibrated on the known arena size); None when no such profile is in the tree.
    Counters cannot be collected from inside the timed run."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)[kernel]
        return float(rec["fetch"] + rec["write"])
    except (OSError, KeyError, ValueError):
        return None


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(prob, tol, target_seconds=12.0):
    """Oracle (CPU restatement of the CVODES path) on all host cores, bounded sample."""
    from oracle.harness import Oracle
    from tools.problems import lv_batch
    cores = usable_cores()
    orc = Oracle(prob, "lv", opt="-O2")
    cfg = orc.config(rtol=tol, atol=tol, rtolB=tol, atolB=tol, rtolQB=tol, atolQB=tol)

    def run(B):
        d = lv_batch(B)
        ps = d["params"][:, prob.params_subset.subset_index]
        pr = d["params"][:, prob.params_subset.remainder_index]
        g = np.ones((len(d["tvals"]), 2))
        t0 = time.perf_counter()
        _, st, _ = orc.solve_forward(cfg, d["y0"], ps, pr, 0.0, d["tvals"], nthreads=cores)
        _, _, st2, _ = orc.solve_backward(cfg, d["tvals"][-1], 0.0, d["tvals"], g, nthreads=cores)
        dt = time.perf_counter() - t0
        assert (st == 0).all() and (st2 == 0).all()
        return dt

    probe = 256 * cores
    dt = run(probe)
    B = int(min(4 * 65536, max(probe, probe * target_seconds / max(dt, 1e-6))))
    dt = run(B)
    return {"value": B / dt, "unit": "solves/s", "cores": cores, "kind": "port",
            "sample": "%d config-2 draws (same generator), fwd+adjoint, OpenMP over instances, gcc -O2, "
                      "threads = cgroup CPU quota" % B}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from sunode_amd import _native
    from tools.problems import lv_batch

    prob = make_problem()
    tol = 1e-8
    B = args.batch
    d = lv_batch(B * world)                       # global synthetic batch; this rank takes a contiguous shard
    sl = slice(rank * B, (rank + 1) * B)
    params = d["params"][sl]
    n, p, n_t = 2, 2, len(d["tvals"])
    ps = torch.tensor(np.ascontiguousarray(params[:, prob.params_subset.subset_index]), device=dev)
    pr = torch.tensor(np.ascontiguousarray(params[:, prob.params_subset.remainder_index]), device=dev)
    y0 = torch.tensor(np.ascontiguousarray(d["y0"][sl]), device=dev)
    tvals = torch.tensor(d["tvals"], device=dev)
    grads = torch.ones((n_t, n), dtype=torch.float64, device=dev)
    y_out = torch.empty((B, n_t, n), dtype=torch.float64, device=dev)
    grad_out = torch.empty((B, p), dtype=torch.float64, device=dev)
    lamda_out = torch.empty((B, n), dtype=torch.float64, device=dev)
    st_f = torch.empty(B, dtype=torch.int32, device=dev)
    st_b = torch.empty(B, dtype=torch.int32, device=dev)
    stats_f = torch.empty((B, 16), dtype=torch.int64, device=dev)
    stats_b = torch.empty((B, 16), dtype=torch.int64, device=dev)

    source = prob.native_source()
    if os.environ.get("SA_ABLATE"):          # timing experiments only (kernel results are then wrong)
        source += "".join("\n#define SA_ABLATE_%s 1\n" % k for k in os.environ["SA_ABLATE"].split(","))
    eng = _native.NativeSolver(source, device=local_rank, rtol=tol, atol=tol, rtolB=tol, atolB=tol,
                               rtolQB=tol, atolQB=tol, traj_capacity=512, n_states=n)
    torch.cuda.synchronize()

    def step():
        eng.solve(_native.SA_MEM_DEVICE, B, y0, ps, pr, 2, 0.0, tvals, n_t, y_out, st_f, stats_f, adjoint=True)
        eng.solve_backward(_native.SA_MEM_DEVICE, B, ps, pr, 2, float(d["tvals"][-1]), 0.0, tvals, n_t, grads, 0,
                           grad_out, lamda_out, st_b, stats_b)

    for _ in range(args.warmup):
        step()
    eng.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    fwd_ms, bwd_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        f, b = eng.last_kernel_ms()               # HIP events on the solver's stream (waits for this step)
        fwd_ms.append(f)
        bwd_ms.append(b)
    eng.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    failed = int((st_f != 0).sum().item() + (st_b != 0).sum().item())
    sf = stats_f.double().mean(dim=0).cpu().numpy()
    sb = stats_b.double().mean(dim=0).cpu().numpy()
    if rank == 0:
        value = world * B * args.steps / elapsed
        # algorithmic HBM bytes per launch:
        #   trajectory record per stored point: 8*(8+6n) B = {order, dt, T[6], Y[6][n]} (built by the forward
        #   kernel, read once by the backward kernel when the lane's table index moves onto it)
        #   backward, per instance: params 8*(p+r) + npts*rec + npts/fwd-status 8 + outputs 8*(p+n) + status 4
        #     + stats 128; shared: grads 8*n_t*n + tvals 8*n_t
        #   forward, per instance: y0+params 8*(n+p+r) + npts*rec + y_out 8*n_t*n + status 4 + np 4 + stats 128
        npts = float(sf[8])
        rec = 8 * (8 + 6 * n)
        bwd_bytes = B * (8 * 4 + npts * rec + 8 + 8 * (p + n) + 4 + 128) + 8 * n_t * n + 8 * n_t
        fwd_bytes = B * (8 * (n + 4) + npts * rec + 8 * n_t * n + 4 + 4 + 128) + 8 * n_t
        bwd_s = float(np.mean(bwd_ms)) * 1e-3
        fwd_s = float(np.mean(fwd_ms)) * 1e-3
        achieved = bwd_bytes / bwd_s / 1e9
        out = {
            "metric": "fp64 forward+adjoint ODE solves/sec at rtol=1e-8 (Lotka-Volterra, batch 65536/GPU)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: Lotka-Volterra forward+adjoint, 2 sens params, "
                                   "n_t=50, rtol=atol=1e-8 (fwd/bwd/quad), grads=ones",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": "instance-sharded x%d" % world,
                       "failed_instances": failed},
            "roofline": {"bound": "hbm", "kernel": "sa_k_backward", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic("sa_k_backward"),
                         "algorithmic_bytes_per_launch": bwd_bytes, "kernel_ms": 1e3 * bwd_s,
                         "forward_kernel_ms": 1e3 * fwd_s, "forward_bytes_per_launch": fwd_bytes,
                         "note": "path is latency/fp64-VALU bound, not HBM bound (SURVEY 8d)"},
            "work": {"fwd_steps_mean": float(sf[0]), "bwd_steps_mean": float(sb[0]),
                     "fwd_attempts_mean": float(sf[14]), "bwd_attempts_mean": float(sb[14]),
                     "bwd_wave_iterations_mean": float(sb[15]),
                     "bdf_steps_per_s": world * B * float(sf[0] + sb[0]) * args.steps / elapsed,
                     "rhs_evals_per_s": world * B * float(sf[1] + sb[1] + sb[9]) * args.steps / elapsed},
        }
        if "PROFILE" in os.environ.get("SA_ABLATE", ""):
            names = ["pre_step+post", "predict+set", "interp", "newton", "errtest+quad", "complete+prepare",
                     "post_step", "interval_setup"]
            tot = float(sb[8:16].sum())
            out["phase_cycles_per_lane"] = {k: [float(v), round(float(v) / tot, 4)] for k, v in zip(names, sb[8:16])}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(prob, tol)
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
