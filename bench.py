"""bench.py -- headline metric of BASELINE.json on MI355X.

metric : fp64 forward+adjoint ODE solves/sec at rtol=1e-8 (Lotka-Volterra, config 2:
         batch 65 536 random parameter draws per GPU, 2 differentiated parameters, 50 outputs,
         rtol=atol=1e-8 for the forward, backward and quadrature problems, grads = ones)
step   : one forward (sa_solve_forward_batch) + one adjoint (sa_solve_backward_batch) pass over the
         whole batch, inputs and outputs resident in HBM (torch tensors, SA_MEM_DEVICE).
N > 1  : `python bench.py --gpus N` starts N ranks itself (re-exec under torch.distributed.run on
         127.0.0.1) unless it already runs under one (WORLD_SIZE set, as the driver launches it);
         one process per GPU, RCCL only for the barrier, the max-reduce of the elapsed time and the
         sum of the failed-instance counts; the batch shards by instance through
         sunode_amd.parallel.shard_indices, no data-path collective -> weak scaling.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     : dominant kernel (sa_k_backward): SURVEY 8(d) algorithmic HBM bytes / HIP-event duration vs
                 8 TB/s, and a `valu` block (the roofline that actually binds: fp64 vector rate)
  cpu_baseline : the CPU oracle (oracle/cvodes_oracle.c, "port") timed on the host cores, bounded sample,
                 all-core and 1-core figures, and the libsundials_cvodes probe
  configs      : (N = 1) the other BASELINE configurations -- Robertson, SEIR, 100-state network -- a few
                 steps each with kernel times, failed-instance counts and their own bounded CPU baselines
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
FP64_VALU_PEAK_TFLOPS = 78.6          # MI355X_MICROARCH.md: fp64 vector peak


# ----------------------------------------------------------------------------------------------
# workloads (BASELINE.json configs 2-5; definitions in tools/problems.py)
# ----------------------------------------------------------------------------------------------
WORKLOADS = {
    "lv": dict(config=2, batch=65536, rtol=1e-8, atol=1e-8,
               label="BASELINE config 2: Lotka-Volterra forward+adjoint, 2 sens params, n_t=50, "
                     "rtol=atol=1e-8 (fwd/bwd/quad), grads=ones"),
    "robertson": dict(config=3, batch=262144, rtol=1e-8, atol=1e-10,
                      label="BASELINE config 3: Robertson stiff 3-state, rtol=1e-8 atol=1e-10, forward+adjoint "
                            "(3 sens params), T=4e4, 7 outputs"),
    "seir": dict(config=4, batch=16384, rtol=1e-8, atol=1e-8,
                 label="BASELINE config 4 (one GPU's share of 131 072): SEIR 16-state, forward+adjoint w.r.t. 8 "
                       "params, n_t=51"),
    "network100": dict(config=5, batch=1024, rtol=1e-8, atol=1e-8,
                       label="BASELINE config 5 (one GPU's share of 8 192): 100-state dense reaction network, "
                             "forward+adjoint w.r.t. 4 params, n_t=11"),
}


def make_problem(name="lv"):
    from tools.problem_cache import make_problem as cached      # (the symbolic work of a problem is cached on disk)
    return cached(name)


def make_batch(name, prob, B, idx=None):
    """Synthetic batch of B draws of workload `name`: dict(y0, ps, pr, rem_stride, tvals, grads).
    ``idx``: only these draws of it (the counter-based generator makes a rank's shard without the global batch)."""
    from tools.problems import lv_batch, network_batch, robertson_batch, seir_batch
    n = prob.n_states
    if name == "lv":
        d = lv_batch(B, idx=idx)
        ps = d["params"][:, prob.params_subset.subset_index]
        pr = d["params"][:, prob.params_subset.remainder_index]
        stride = pr.shape[1]
        grads = np.ones((len(d["tvals"]), n))
    elif name == "robertson":
        d = robertson_batch(B, idx=idx)
        ps, pr, stride = d["params"], np.zeros(1), 0
        grads = None
    else:
        d = seir_batch(B, idx=idx) if name == "seir" else network_batch(B, idx=idx)
        ps, pr, stride = d["ps"], d["pr"], 0
        grads = None
    if prob.n_remainder:          # hoisted fixed-parameter sub-expressions ride at the end of the remainder vector
        pr = np.asarray(prob.extend_remainder(pr))
        stride = pr.shape[-1] if stride else 0
    tv = d["tvals"]
    if grads is None:
        k = np.arange(len(tv))[:, None]
        i = np.arange(n)[None, :]
        grads = 1.0 + 0.5 * np.cos(1.7 * k + 0.9 * i)
    return dict(y0=np.ascontiguousarray(d["y0"]), ps=np.ascontiguousarray(ps), pr=np.ascontiguousarray(pr),
                rem_stride=stride, tvals=np.ascontiguousarray(tv), grads=np.ascontiguousarray(grads))


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


# ----------------------------------------------------------------------------------------------
# per-rank engines: the GPU engine (product path) and, for the CPU multi-process test, whatever
# `make_engine` the test injects.  An engine integrates the rank's shard: step() = forward + adjoint.
# ----------------------------------------------------------------------------------------------
class GpuEngine:
    """libsunode_amd.so on cuda:<local_rank>, device-resident tensors, work enqueued on a torch stream."""

    def __init__(self, name, prob, batch, tol, local_rank, arena_bytes=0):
        import torch
        from sunode_amd import _native
        self.torch, self._native = torch, _native
        self.dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(local_rank)
        n, p = prob.n_states, prob.n_params
        self.n, self.p, self.B, self.n_t = n, p, batch["y0"].shape[0], len(batch["tvals"])
        t = lambda a: torch.tensor(np.ascontiguousarray(a), device=self.dev)          # noqa: E731
        self.y0, self.ps, self.pr, self.tvals = t(batch["y0"]), t(batch["ps"]), t(batch["pr"]), t(batch["tvals"])
        self.grads = t(batch["grads"])
        self.rem_stride = batch["rem_stride"]
        self.t_end = float(batch["tvals"][-1])
        B, n_t = self.B, self.n_t
        f64, dev = torch.float64, self.dev
        self.y_out = torch.empty((B, n_t, n), dtype=f64, device=dev)
        self.grad_out = torch.empty((B, max(p, 1)), dtype=f64, device=dev)
        self.lamda_out = torch.empty((B, n), dtype=f64, device=dev)
        self.st_f = torch.empty(B, dtype=torch.int32, device=dev)
        self.st_b = torch.empty(B, dtype=torch.int32, device=dev)
        self.stats_f = torch.empty((B, 16), dtype=torch.int64, device=dev)
        self.stats_b = torch.empty((B, 16), dtype=torch.int64, device=dev)
        source = prob.native_source()
        if os.environ.get("SA_ABLATE"):          # timing experiments only (kernel results are then wrong)
            source += "".join("\n#define SA_ABLATE_%s 1\n" % k for k in os.environ["SA_ABLATE"].split(","))
        rt, at = tol
        self.eng = _native.NativeSolver(source, device=local_rank, rtol=rt, atol=at, rtolB=rt, atolB=at,
                                        rtolQB=rt, atolQB=at, n_states=n, arena_bytes=arena_bytes,
                                        compact=_native.default_compact_trajectory(source),   # = AdjointSolver's default
                                        guard_kinds=("adjoint",))                            # the kind this engine runs
        # stream ordering (include/sunode_amd.h): the solver launches on the torch stream that owns the tensors
        self.stream = torch.cuda.Stream(device=self.dev)
        self.eng.set_stream(self.stream.cuda_stream)
        torch.cuda.synchronize()
        if self.eng._guard_open:
            # differential guard (sunode_amd/_native.py NativeSolver): a code object without a verdict runs its first
            # 64 instances through the default AND the conservative build on its first forward + backward pair -- here,
            # outside the timed region, whatever --warmup says
            self.step()
            self.sync()
        self.build = {"code_object": _native.file_hash(self.eng.code_object),
                      "code_object_file": os.path.basename(self.eng.code_object), "toolchain": _native.toolchain_id()}

    def step(self):
        e, N = self.eng, self._native
        with self.torch.cuda.stream(self.stream):
            e.solve(N.SA_MEM_DEVICE, self.B, self.y0, self.ps, self.pr, self.rem_stride, 0.0, self.tvals, self.n_t,
                    self.y_out, self.st_f, self.stats_f, adjoint=True)
            e.solve_backward(N.SA_MEM_DEVICE, self.B, self.ps, self.pr, self.rem_stride, self.t_end, 0.0, self.tvals,
                             self.n_t, self.grads, 0, self.grad_out, self.lamda_out, self.st_b, self.stats_b)

    def kernel_ms(self):
        return self.eng.last_kernel_ms()          # HIP events on the solver's stream (waits for this step)

    def sync(self):
        self.eng.synchronize()
        self.torch.cuda.synchronize()

    def results(self):
        self.sync()
        return dict(failed=int((self.st_f != 0).sum().item() + (self.st_b != 0).sum().item()),
                    stats_f=self.stats_f.double().mean(dim=0).cpu().numpy(),
                    stats_b=self.stats_b.double().mean(dim=0).cpu().numpy(),
                    arena=self.eng.arena_info(), build=self.build, guard=dict(self.eng.guard_report),
                    head_grads=self.grad_out[:16, :max(self.p, 1)].cpu().numpy(),
                    head_lamda=self.lamda_out[:16].cpu().numpy())

    def close(self):
        self.eng.close()


class MultiDeviceEngine:
    """--single-process: ONE Python process drives several GPUs -- one GpuEngine (handle, stream, arena, device
    tensors) per entry of `devices`, every step issued from one host thread per engine (the native calls release
    the GIL).  The second launcher next to one-process-per-GPU; what AdjointSolver(devices=[...]) does for host
    arrays, with device-resident shards."""

    def __init__(self, name, prob, batch, tol, devices, factory=None):
        from concurrent.futures import ThreadPoolExecutor
        factory = factory or GpuEngine
        B = batch["y0"].shape[0]
        per = B // len(devices)
        assert per * len(devices) == B
        self.engines = []
        for k, d in enumerate(devices):
            rows = slice(k * per, (k + 1) * per)
            shard = dict(batch, y0=batch["y0"][rows], ps=batch["ps"][rows],
                         pr=batch["pr"][rows] if batch["rem_stride"] else batch["pr"])
            self.engines.append(factory(name, prob, shard, tol, d, arena_bytes=self._arena_share(devices, d)))
        self.dev = getattr(self.engines[0], "dev", "cpu")
        self.pool = ThreadPoolExecutor(max_workers=len(devices), thread_name_prefix="bench")

    @staticmethod
    def _arena_share(devices, d):
        k = devices.count(d)
        if k == 1:
            return 0                      # the library default (96 GiB, at most 60 % of the free HBM)
        from sunode_amd import _native
        free_b, _ = _native.device_memory(d)
        return min(96 << 30, int(0.6 * free_b)) // k

    def step(self):
        for f in [self.pool.submit(e.step) for e in self.engines]:
            f.result()

    def kernel_ms(self):
        ms = [e.kernel_ms() for e in self.engines]
        return max(m[0] for m in ms), max(m[1] for m in ms)

    def sync(self):
        for e in self.engines:
            e.sync()

    def results(self):
        rs = [e.results() for e in self.engines]
        out = dict(failed=sum(r["failed"] for r in rs),
                   stats_f=np.mean([r["stats_f"] for r in rs], axis=0), stats_b=np.mean([r["stats_b"] for r in rs], axis=0),
                   arena=(max(r["arena"][0] for r in rs), sum(r["arena"][1] for r in rs), any(r["arena"][2] for r in rs)))
        for key in ("build", "guard", "head_grads", "head_lamda"):      # (engine 0 integrates draws 0..B-1 of the global batch)
            if key in rs[0]:
                out[key] = rs[0][key]
        return out

    def close(self):
        for e in self.engines:
            e.close()
        self.pool.shutdown()


def run_single_process(args, *, make_engine=None):
    """`--gpus N --single-process [--devices 0,1,...]`: no process group; N engines in this process."""
    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(max(1, args.gpus)))
    if len(devices) != max(1, args.gpus):
        raise SystemExit("bench.py: --devices names %d device(s) but --gpus is %d" % (len(devices), args.gpus))
    if make_engine is None:
        # torch brings its own HIP runtime: it has to initialise before libsunode_amd.so touches the devices (the
        # arena-budget query below), or torch finds "no HIP GPUs" afterwards
        import torch
        torch.cuda.init()
        if torch.cuda.device_count() <= max(devices):
            raise SystemExit("bench.py: --devices %s but only %d GPU(s) visible" % (args.devices, torch.cuda.device_count()))
    name = args.workload
    prob = make_problem(name)
    w = WORKLOADS[name]
    B = args.batch or w["batch"]
    world = len(devices)
    batch = make_batch(name, prob, B * world)
    eng = MultiDeviceEngine(name, prob, batch, (w["rtol"], w["atol"]), devices, factory=make_engine)
    for _ in range(args.warmup):
        eng.step()
    eng.sync()
    fwd_ms, bwd_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step()
        f, b = eng.kernel_ms()
        fwd_ms.append(f)
        bwd_ms.append(b)
    eng.sync()
    elapsed = time.perf_counter() - t0
    res = eng.results()
    out = summarise(name, prob, w, B, world, args, elapsed, fwd_ms, bwd_ms, res, res["failed"])
    out["config"]["parallelism"] = "instance-sharded x%d, one process, one host thread per device (devices %s)" \
        % (world, ",".join(map(str, devices)))
    per_dev = [sum(e.kernel_ms()) for e in eng.engines]
    out["rank_ms_per_step"] = {"min": min(per_dev), "max": max(per_dev), "imbalance": max(per_dev) / min(per_dev) - 1.0,
                               "what": "kernel ms (forward + backward) of the last step per handle"}
    eng.close()
    return out


def run_rank(args, *, backend="nccl", make_engine=None):
    """One rank of the benchmark (rank / world from the torch.distributed.run environment).  Returns the result
    dict on rank 0, None elsewhere.  `backend` and `make_engine` are injectable so that the world-size-2 CPU
    test (tests/test_parallel_gloo.py) runs THIS function over gloo with a CPU stand-in for the per-rank solve."""
    from sunode_amd.parallel import shard_indices
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    use_dist = world > 1 or getattr(args, "force_dist", False)   # --force-dist: a world of one still forms its group
    if use_dist:
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            if torch.cuda.device_count() < world:
                raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible" % (world, torch.cuda.device_count()))
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    name = args.workload
    prob = make_problem(name)
    w = WORKLOADS[name]
    B = args.batch or w["batch"]
    # weak scaling: the global synthetic batch has B * world draws; every rank generates and integrates its own shard
    # (draw i depends only on (seed, i): no rank materialises the global batch)
    idx = shard_indices(B * world, rank, world)
    shard = make_batch(name, prob, B * world, idx=idx)
    factory = make_engine or GpuEngine
    eng = factory(name, prob, shard, (w["rtol"], w["atol"]), local_rank)

    def all_reduce(value, op):
        if not use_dist:
            return value
        import torch
        t = torch.tensor([value], dtype=torch.float64, device=getattr(eng, "dev", "cpu"))
        dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
        return float(t.item())

    for _ in range(args.warmup):
        eng.step()
    eng.sync()
    if use_dist:
        dist.barrier()
    fwd_ms, bwd_ms = [], []
    t0 = time.perf_counter()
    trace = []                                      # per-step wall clock (SA_BENCH_TRACE=1 prints it to stderr)
    for _ in range(args.steps):
        eng.step()
        f, b = eng.kernel_ms()
        fwd_ms.append(f)
        bwd_ms.append(b)
        trace.append(time.perf_counter())
    eng.sync()
    elapsed = time.perf_counter() - t0
    if os.environ.get("SA_BENCH_TRACE"):
        print("bench.py step wall ms: " + " ".join("%.3f" % (1e3 * (b_ - a_)) for a_, b_ in zip([t0] + trace, trace))
              + " | kernel ms: " + " ".join("%.3f" % (x + y) for x, y in zip(fwd_ms, bwd_ms)), file=sys.stderr, flush=True)
    own = elapsed                                   # this rank's time before it waited for the others
    if use_dist:
        dist.barrier()
    elapsed = all_reduce(own, "MAX")
    fastest = all_reduce(own, "MIN")
    res = eng.results()
    failed = int(all_reduce(float(res["failed"]), "SUM"))
    out = None
    if rank == 0:
        out = summarise(name, prob, w, B, world, args, elapsed, fwd_ms, bwd_ms, res, failed)
        # an imbalanced node (a slow GPU, a rank that drew the expensive instances) shows here the first time the
        # driver has eight GPUs: the line's value follows the SLOWEST rank
        out["rank_ms_per_step"] = {"min": 1e3 * fastest / args.steps, "max": 1e3 * elapsed / args.steps,
                                   "imbalance": elapsed / fastest - 1.0 if fastest > 0 else None}
    eng.close()
    if use_dist:
        dist.destroy_process_group()
    return out


def algorithmic_bytes(prob, n_t, npts, grads_broadcast=True, per_instance_rem=True):
    """B_alg per solve = 8*[n + P_inst + n_t*n (y_out) + n_t*n (grads; 0 if broadcast) + p + n]
    + 2 * S_f * (8*(n+1) + 8), split here by kernel: the forward kernel reads y0 + params, writes y_out and the
    S_f data points (t, y[n], order); the backward kernel reads params + the data points once, writes p + n."""
    n, p = prob.n_states, prob.n_params
    p_inst = p + (prob.n_remainder if per_instance_rem else 0)
    point = 8 * (n + 1) + 8
    fwd = 8 * (n + p_inst + n_t * n) + npts * point
    bwd = 8 * (p_inst + (0 if grads_broadcast else n_t * n) + p + n) + npts * point
    return fwd, bwd


def callback_flops(prob):
    """fp64 operations per call of each generated callback, counted in the generated source (one per
    + - * / and two per fma): the F_rhs / F_jac of SURVEY.md 8(d) without hand-written constants."""
    import re
    src = prob.native_source()
    cut = src.index("SA_FN int sa_rhs") if "SA_FN int sa_rhs" in src else 0
    marks = [(m.start(), m.group(1)) for m in re.finditer(r"SA_FN int (sa_[a-z_]+?)(?:_c\d+)?\(", src[cut:])]
    out = {}
    for k, (pos, fname) in enumerate(marks):
        end = marks[k + 1][0] if k + 1 < len(marks) else len(src) - cut
        body = src[cut + pos:cut + end]
        if "SA_CHUNK_CALL" in body:
            continue
        stmts = [ln for ln in body.splitlines() if "const double" in ln or "SA_STORE" in ln]
        text = "\n".join(stmts)
        ops = len(re.findall(r"(?<![eE(,])[-+*/](?![=/*])", text)) + 2 * text.count("fma(")
        out[fname] = out.get(fname, 0) + ops
    return out


def algorithmic_flops(prob, sf, sb):
    """SURVEY.md 8(d) F_alg per solve from the kernels' own counters (stats slots: nst 0, nfe 1, nsetups 2, nje 3,
    nni 4, nfqe 9, ninterp 11): callback calls x their operation counts + triangular solves 2n^2 per Newton
    iteration + Nordsieck work 6n(q+2) per step + LU 2/3 n^3 per setup + interpolation 2n(q+1) per call."""
    n, p = prob.n_states, prob.n_params
    f = callback_flops(prob)
    q = 4.0
    lu = (2.0 / 3.0) * n ** 3
    fwd = (sf[1] * f.get("sa_rhs", 0) + sf[3] * f.get("sa_jac", 0) + sf[4] * 2 * n * n + sf[0] * 6 * n * (q + 2)
           + sf[2] * lu)
    bwd = (sb[1] * f.get("sa_adj_rhs", 0) + sb[9] * f.get("sa_quad_rhs", 0) + sb[3] * f.get("sa_adj_jac", 0)
           + sb[4] * 2 * n * n + sb[0] * 6 * (n + p) * (q + 2) + sb[2] * lu + sb[11] * 2 * n * (q + 1))
    return float(fwd), float(bwd)


def pmc_profile(workload, kernel):
    """Committed rocprofv3 --pmc results of this command for `workload` (profiles/pmc_traffic.json, written by
    tools/make_pmc_traffic.py from the PMC summaries of tools/gpu_profiles.sh; collected in separate passes as
    MI355X_MICROARCH.md prescribes -- counters cannot be read from inside the timed run)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            return json.load(fh)[workload][kernel]
    except (OSError, KeyError, ValueError):
        return None


def pmc_context_matches(workload, B, rec, build):
    """(ok, note): do the committed counters describe THIS run?  The profile records the batch, the arena record size,
    the code-object hash and the toolchain it was taken with; counters of another build are not printed as
    if measured here."""
    ctx = pmc_profile(workload, "context")
    if ctx is None:
        return False, "profiles/pmc_traffic.json carries no context for this workload: counter ratios withheld"
    diff = [k for k, v in (("batch", B), ("record_bytes", rec), ("code_object", (build or {}).get("code_object")),
                           ("toolchain", ((build or {}).get("toolchain") or {}).get("hash"))) if ctx.get(k) != v]
    if diff:
        return False, "profiled with a different %s (profiles/pmc_traffic.json): counter ratios withheld" % " / ".join(diff)
    return True, "counters of this code object, batch and record format"


def truth_gradient_error(name, res):
    """max relative error of dL/dp and dL/dy0 of the timed batch's first 16 draws against the committed truth fixture
    (tests/golden/truth_lv.npz: DOP853 on the sensitivity equations, same generator, same cotangent) -- LV only."""
    if name != "lv" or "head_grads" not in res:
        return None
    try:
        t = np.load(os.path.join(ROOT, "tests", "golden", "truth_lv.npz"))
    except OSError:
        return None
    g, lam = res["head_grads"], res["head_lamda"]
    k = min(len(g), len(t["grad_params"]))
    eg = np.max(np.abs(g[:k] - t["grad_params"][:k]) / np.abs(t["grad_params"][:k]).max(axis=1, keepdims=True))
    el = np.max(np.abs(-lam[:k] - t["grad_y0"][:k]) / np.abs(t["grad_y0"][:k]).max(axis=1, keepdims=True))
    return {"draws": int(k), "dL_dp_max_rel": float(eg), "dL_dy0_max_rel": float(el),
            "fixture": "tests/golden/truth_lv.npz (rtol = atol = 1e-8 bar in the tests: 4e-6)",
            # north_star asks for 1e-6 relative vs CVODES.  CVODES is not on the box, so this compares with TRUTH
            # (DOP853 1e-13 on the sensitivity equations), which is stricter: it contains the BDF global error at
            # rtol = 1e-8 that CVODES carries as well (the controller equals DVODE counter for counter in both
            # directions: tests/test_oracle_pinning.py), i.e. device - CVODES is expected far below device - truth.
            "north_star_bar_vs_cvodes": 1e-6, "within_1e-6_of_truth": bool(max(eg, el) <= 1e-6),
            "note": "vs truth, not vs CVODES (unavailable): includes the integration error at rtol = 1e-8 that CVODES "
                    "itself carries; dL/dy0 of a few draws exceeds 1e-6 against truth for that reason"}


def summarise(name, prob, w, B, world, args, elapsed, fwd_ms, bwd_ms, res, failed):
    n, p = prob.n_states, prob.n_params
    sf, sb = res["stats_f"], res["stats_b"]
    n_t = {"lv": 50, "robertson": 7, "seir": 51, "network100": 11}[name]
    value = world * B * args.steps / elapsed
    npts = float(sf[8])
    fwd_alg, bwd_alg = algorithmic_bytes(prob, n_t, npts, grads_broadcast=True, per_instance_rem=(name == "lv"))
    bwd_s, fwd_s = float(np.mean(bwd_ms)) * 1e-3, float(np.mean(fwd_ms)) * 1e-3
    achieved = B * bwd_alg / bwd_s / 1e9
    f_fwd, f_bwd = algorithmic_flops(prob, sf, sb)
    pmc = pmc_profile(name, "sa_k_backward")
    pmc_f = pmc_profile(name, "sa_k_forward")
    arena_bytes, tiles, tiled = res.get("arena", (0, 0, False))
    from sunode_amd import _native
    compact = _native.default_compact_trajectory(prob.native_source())
    rec = 8 * (n + 2) if compact else 8 * (8 + 6 * n)
    build = res.get("build")
    ctx_ok, ctx_note = pmc_context_matches(name, B, rec, build)
    if not ctx_ok:
        pmc = pmc_f = None
    have = bool(pmc and "fetch" in pmc and "write" in pmc)
    traffic_b = (pmc["fetch"] + pmc["write"]) if have else None
    traffic_f = (pmc_f["fetch"] + pmc_f["write"]) if (pmc_f and "fetch" in pmc_f and "write" in pmc_f) else None
    ctx = pmc_profile(name, "context") if ctx_ok else None
    prof_ms = (ctx or {}).get("kernel_ms") or {}
    out = {
        "metric": "fp64 forward+adjoint ODE solves/sec at rtol=1e-8 (%s, batch %d/GPU)"
                  % ({"lv": "Lotka-Volterra"}.get(name, name), B),
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": w["label"], "batch_per_gpu": B, "global_batch": B * world,
                   "parallelism": "instance-sharded x%d" % world, "failed_instances": failed},
        "roofline": {
            # bound / achieved / peak / frac / traffic are ONE consistent set: the CONTRACT figures, algorithmic HBM bytes
            #  against the 8 TB/s peak.  The path is not HBM-bound (thousands of tiny sequential
            # solves): `binding_limiter` names what does bind -- the issue rate of the fp64 vector unit at one wavefront
            # per SIMD (one instruction of ANY kind per four cycles: profiles/r04_ubench_issue.txt) -- and `valu` holds
            # its figures.
            "bound": "hbm", "binding_limiter": "valu-issue",
            "kernel": "sa_k_backward", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic_b,
            "hbm": {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS},
            "traffic_source": ("profiles/pmc_traffic.json (separate rocprofv3 --pmc passes of this command): " + ctx_note)
                              if have else ctx_note,
            # kernel time of the profiled run next to this run's (a gap above 5 % means the counters describe a
            # different machine state: flagged)
            "profiled_kernel_ms": prof_ms.get("sa_k_backward"), "profiled_forward_kernel_ms": prof_ms.get("sa_k_forward"),
            "kernel_ms_gap": (abs(1e3 * bwd_s / prof_ms["sa_k_backward"] - 1.0) if prof_ms.get("sa_k_backward") else None),
            "kernel_ms_gap_flag": (bool(abs(1e3 * bwd_s / prof_ms["sa_k_backward"] - 1.0) > 0.05)
                                   if prof_ms.get("sa_k_backward") else None),
            # what the counters say the kernels really move, against the same peak (bytes of the committed PMC pass /
            # this run's kernel time): the figure to watch for wasted re-reads
            "hbm_frac_traffic": (traffic_b / bwd_s / 1e9 / HBM_PEAK_GBS) if have else None,
            "forward_traffic": traffic_f,
            "forward_hbm_frac_traffic": (traffic_f / fwd_s / 1e9 / HBM_PEAK_GBS) if traffic_f else None,
            "traffic_over_algorithmic": ((traffic_b + (traffic_f or 0.0)) / (B * (bwd_alg + (fwd_alg if traffic_f else 0.0)))) if have else None,
            "arena_record": "compact {order, t, y[n]}: %d B per stored step" % rec if compact else
                            "table {order, dt, T[6], Y[6][n]}: %d B per stored step" % rec,
            "algorithmic_bytes_per_launch": B * bwd_alg, "kernel_ms": 1e3 * bwd_s,
            "forward_kernel_ms": 1e3 * fwd_s, "forward_bytes_per_launch": B * fwd_alg,
            "algorithmic_bytes_per_solve_8d": fwd_alg + bwd_alg,
            # what the implementation moves by design: one {order, dt, T[6], Y[6][n]} record per stored point
            "traffic_model": B * (8 * (p + prob.n_remainder) + npts * rec + 8 * (p + n) + 140),
            "limiter": "neither contract roofline binds: the path is thousands of tiny sequential solves (SURVEY 8d); "
                       "what limits it is the instruction issue rate of ONE wavefront per SIMD (4 cycles per instruction "
                       "of any kind, 5 when dependent: profiles/r04_ubench_issue.txt) x lane utilisation -- see valu",
            "valu": {"kernel": "sa_k_backward", "unit": "TFLOP/s", "peak": FP64_VALU_PEAK_TFLOPS,
                     "algorithmic": B * f_bwd / bwd_s / 1e12, "frac_algorithmic": B * f_bwd / bwd_s / 1e12 / FP64_VALU_PEAK_TFLOPS,
                     "issued": (pmc["valu_fp64_flops"] / bwd_s / 1e12) if pmc and "valu_fp64_flops" in pmc else None,
                     "valu_busy": pmc.get("valu_busy") if pmc else None,
                     "lane_utilisation": pmc.get("lane_utilisation") if pmc else None,
                     # VALU wave-instructions per step attempt of a wavefront (the lever for configs 2 and 3 and for
                     # small batches: VERDICT r3 #4): counter VALU instructions / (wavefronts x mean attempts)
                     "valu_insts_per_attempt": (pmc["valu_insts"] / ((B + 63) // 64 * max(float(sb[14]), 1.0))
                                                if pmc and "valu_insts" in pmc and name in ("lv", "robertson") else None),
                     "note": "algorithmic = SURVEY 8(d) F_alg from the kernel's own counters; issued = fp64 "
                             "FMA/MUL/ADD lane-ops of the committed PMC pass (all 64 lanes counted) / this run's "
                             "kernel time"}},
        "work": {"fwd_steps_mean": float(sf[0]), "bwd_steps_mean": float(sb[0]),
                 "fwd_attempts_mean": float(sf[14]), "bwd_attempts_mean": float(sb[14]),
                 "stored_points_mean": npts,
                 "bdf_steps_per_s": world * B * float(sf[0] + sb[0]) * args.steps / elapsed,
                 "rhs_evals_per_s": world * B * float(sf[1] + sb[1] + sb[9]) * args.steps / elapsed,
                 "arena": {"bytes": arena_bytes, "tiled": bool(tiled), "reintegrated_tiles": tiles}},
    }
    if pmc and "valu_fp64_flops" in pmc:
        out["roofline"]["valu"]["frac_issued"] = out["roofline"]["valu"]["issued"] / FP64_VALU_PEAK_TFLOPS
    out["build"] = dict(build or {}, batch=B, record_bytes=rec)
    if res.get("guard") is not None:
        # what the differential guard found for this code object (verdict file next to it, or this run's check)
        out["guard"] = res["guard"]
    err = truth_gradient_error(name, res)
    if err:
        out["truth_gradient_error"] = err
    return out


# ----------------------------------------------------------------------------------------------
# CPU baseline (oracle = "port"); libsundials_cvodes probe
# ----------------------------------------------------------------------------------------------
def cvodes_probe(name="lv", prob=None, w=None, seconds=8.0):
    """BASELINE.md section 3 / SURVEY 8(d): time real CVODES when the box has it -- oracle/cvodes_driver.py drives
    libsundials_cvodes with the reference's own call sequence (solver.py:565-615, 682-784) and callbacks compiled
    from the generated C header, one draw at a time on one core like the reference, and reports it next to the
    restated oracle on the same draws (states / gradients / forward step counters: the one place this repository
    meets CVODES itself).  The image carries no SUNDIALS (conda-forge `sundials<6.0` is what the reference links),
    so normally the row just says so."""
    from oracle import cvodes_driver as drv
    libs = drv.find_libraries()
    if not libs:
        return {"available": False, "note": "libsundials_cvodes (+ nvecserial, sunmatrixdense, sunlinsoldense) not found "
                                            "on this host (ctypes.util.find_library); the reference's sunode+CVODES path "
                                            "cannot be timed here"}
    try:
        import ctypes
        from oracle.harness import Oracle
        prob = prob or make_problem(name)
        w = w or WORKLOADS[name]
        rt, at = w["rtol"], w["atol"]
        cb = ctypes.CDLL(drv.build_callbacks(prob.native_source(), name))
        d = drv.CvodesDriver(prob.n_states, prob.n_params, cb, drv.load(libs), rtol=rt, atol=at, rtolB=rt, atolB=at,
                             rtolQB=rt, atolQB=at)
        b = make_batch(name, prob, 64)
        ys, gs, ls, cnt, t0, done = [], [], [], [], time.perf_counter(), 0
        for i in range(64):
            d.set_params(b["ps"][i], b["pr"][i] if b["rem_stride"] else b["pr"])
            ys.append(d.solve_forward(0.0, b["tvals"], b["y0"][i]))
            cnt.append(d.counters())
            g, lam = d.solve_backward(b["tvals"][-1], 0.0, b["tvals"], b["grads"])
            gs.append(g); ls.append(lam)
            done += 1
            if time.perf_counter() - t0 > seconds:
                break
        dt = time.perf_counter() - t0
        orc = Oracle(prob, name)
        cfg = orc.config(rtol=rt, atol=at, rtolB=rt, atolB=at, rtolQB=rt, atolQB=at)
        pr = b["pr"][:done] if b["rem_stride"] else b["pr"]
        yo, _, so = orc.solve_forward(cfg, b["y0"][:done], b["ps"][:done], pr, 0.0, b["tvals"])
        go, lo, _, _ = orc.solve_backward(cfg, b["tvals"][-1], 0.0, b["tvals"], b["grads"])
        rel = lambda a, r: float(np.max(np.abs(np.asarray(a) - r) / np.maximum(np.abs(r).max(axis=-1, keepdims=True), 1e-300)))  # noqa: E731
        return {"available": True, "libraries": libs, "value": done / dt, "unit": "solves/s", "cores": 1,
                "kind": "reference", "sample": "%d %s draws, one at a time (the reference's call pattern)" % (done, name),
                "vs_oracle": {"states_max_rel": rel(ys, yo), "grad_max_rel": rel(gs, go), "lamda_max_rel": rel(ls, lo),
                              "forward_counters_equal": bool(np.array_equal(np.array(cnt), so[:, :7]))}}
    except Exception as exc:                          # a half-installed SUNDIALS must not take the bench line down
        return {"available": True, "libraries": libs, "error": "%s: %s" % (type(exc).__name__, exc)}


def cpu_baseline(name, prob, w, target_seconds=12.0, opt="-O3"):
    """Oracle (CPU restatement of the CVODES path, oracle/cvodes_oracle.c) on the host cores, bounded sample of the
    same workload: all usable cores (OpenMP over instances) and one core."""
    from oracle.harness import Oracle
    cores = usable_cores()
    orc = Oracle(prob, name, opt=opt)
    rt, at = w["rtol"], w["atol"]
    cfg = orc.config(rtol=rt, atol=at, rtolB=rt, atolB=at, rtolQB=rt, atolQB=at)

    def run(B, threads):
        d = make_batch(name, prob, B)
        t0 = time.perf_counter()
        _, st, _ = orc.solve_forward(cfg, d["y0"], d["ps"], d["pr"], 0.0, d["tvals"], nthreads=threads)
        _, _, st2, _ = orc.solve_backward(cfg, d["tvals"][-1], 0.0, d["tvals"], d["grads"], nthreads=threads)
        dt = time.perf_counter() - t0
        assert (st == 0).all() and (st2 == 0).all()
        return dt

    probe = {"lv": 256, "robertson": 32, "seir": 8, "network100": 1}[name] * cores
    dt = run(probe, cores)
    B = int(min(4 * w["batch"], max(probe, probe * target_seconds / max(dt, 1e-6))))
    dt = run(B, cores) if B > probe else dt
    B1 = max(1, B // (cores * 6))
    dt1 = run(B1, 1)
    return {"value": B / dt, "unit": "solves/s", "cores": cores, "kind": "port",
            "one_core": {"value": B1 / dt1, "sample": "%d draws" % B1},
            "sample": "%d %s draws (same generator), fwd+adjoint, OpenMP over instances, gcc %s, threads = cgroup "
                      "CPU quota; the port mirrors the kernels' arithmetic bit for bit (explicit FMAs, software pow, "
                      "tree sums): a stated baseline, not a tuned CPU code" % (B, name, opt),
            "cvodes": cvodes_probe(name, prob, w)}


def host_api(name, prob, w, B, steps=5):
    """The number a PyMC user sees: the SAME step through the product's Python API with numpy arrays in and out
    (`AdjointSolver.solve_forward_batch` + `solve_backward_batch`, the calls the pytensor Ops make,
    /root/reference/sunode/wrappers/as_pytensor.py:279-344): host -> device staging, kernels, results back in numpy
    arrays, Python overhead -- everything.  Never `value` (the contract times device-resident inputs); reported beside it."""
    from sunode_amd.solver import AdjointSolver
    b = make_batch(name, prob, B)
    rt, at = w["rtol"], w["atol"]
    sol = AdjointSolver(prob, abstol=at, reltol=rt, backward_abstol=at, backward_reltol=rt, quad_abstol=at, quad_reltol=rt)
    pr = b["pr"] if not b["rem_stride"] else b["pr"][:, :prob.n_remainder]
    if not b["rem_stride"] and prob.n_remainder:
        pr = b["pr"][:prob.n_remainder]                  # (the solver appends the hoisted values itself)
    tv, t_end = b["tvals"], float(b["tvals"][-1])

    # the reference's convention: the caller allocates the outputs once, the solver writes in place
    # (/root/reference/sunode/solver.py:682,723-724: solve_forward(..., y_out), solve_backward(..., grad_out, lamda_out))
    n, p = prob.n_states, prob.n_params
    fwd_out = {"y_out": np.empty((B, len(tv), n)), "status": np.empty(B, np.int32), "stats": np.empty((B, 16), np.int64)}
    bwd_out = {"grad_out": np.empty((B, p)), "lamda_out": np.empty((B, n)), "status": np.empty(B, np.int32),
               "stats": np.empty((B, 16), np.int64)}

    def step():
        y, st, _ = sol.solve_forward_batch(0.0, tv, b["y0"], b["ps"], pr, out=fwd_out)
        g, lam, stb, _ = sol.solve_backward_batch(t_end, 0.0, tv, b["grads"], out=bwd_out)
        return int((st != 0).sum() + (stb != 0).sum())
    step(); step()                                       # (guard check, arena sizing, first touch of the output pages)
    t0 = time.perf_counter()
    failed = 0
    for _ in range(steps):
        failed += step()
    dt = (time.perf_counter() - t0) / steps
    f_ms, b_ms = sol.last_kernel_ms()
    return {"solves_per_s": B / dt, "ms_per_step": 1e3 * dt, "kernel_ms": f_ms + b_ms, "batch": B, "steps": steps,
            "failed_instances": failed,
            "what": "AdjointSolver.solve_forward_batch + solve_backward_batch, numpy arrays in and out (SA_MEM_HOST), "
                    "wall clock; caller-allocated output arrays through out= (the reference's convention)"}


def first_use(workload):
    """What a NEW model costs before its first batch (VERDICT r5 missing #9): the symbolic derivation + C generation and
    the default + conservative gfx950 code objects, from a cold cache.  `measured_here`: the headline workload, rebuilt
    now on this box's host cores (force=True: the cached objects are replaced by identical ones); `recorded`: every
    BASELINE config as measured by tools/first_use.py in the build container (profiles/r06_first_use.json --
    network100 is a minute of sympy and more of clang: not repeated inside a bench run)."""
    from tools.first_use import measure
    out = {"measured_here": {workload: measure(workload)}, "host_cores": usable_cores()}
    try:
        with open(os.path.join(ROOT, "profiles", "r06_first_use.json")) as fh:
            out["recorded"] = json.load(fh)
    except (OSError, ValueError):
        out["recorded"] = None
    return out


def extra_configs(args):
    """BASELINE configs 3-5 on this GPU: a couple of steps each (N = 1 only; they are parity-test cases, the
    headline `value` is config 2)."""
    rows = {}
    for name in ("robertson", "seir", "network100"):
        w = WORKLOADS[name]
        try:
            prob = make_problem(name)
            sub = argparse.Namespace(**vars(args))
            # two untimed steps: the first call on a handle sizes the trajectory arena (and may re-integrate), the
            # second allocates its final size; from the third on the allocation is stable
            sub.workload, sub.batch, sub.steps, sub.warmup, sub.gpus = name, 0, 5, 2, 1
            r = run_rank(sub)
            row = {"workload": w["label"], "batch": w["batch"], "solves_per_s": r["value"],
                   "ms_per_step": r["ms_per_step"], "forward_kernel_ms": r["roofline"]["forward_kernel_ms"],
                   "backward_kernel_ms": r["roofline"]["kernel_ms"],
                   "failed_instances": r["config"]["failed_instances"],
                   "hbm_frac": r["roofline"]["frac"], "hbm_frac_traffic": r["roofline"]["hbm_frac_traffic"],
                   "forward_hbm_frac_traffic": r["roofline"]["forward_hbm_frac_traffic"],
                   "traffic_over_algorithmic": r["roofline"]["traffic_over_algorithmic"],
                   "traffic_source": r["roofline"]["traffic_source"],
                   "profiled_backward_kernel_ms": r["roofline"]["profiled_kernel_ms"],
                   "kernel_ms_gap_flag": r["roofline"]["kernel_ms_gap_flag"], "build": r.get("build"),
                   "arena_record": r["roofline"]["arena_record"],
                   "valu_frac_algorithmic": r["roofline"]["valu"]["frac_algorithmic"],
                   "valu_busy": r["roofline"]["valu"]["valu_busy"], "lane_utilisation": r["roofline"]["valu"]["lane_utilisation"],
                   "work": r["work"]}
            if not args.no_cpu_baseline:
                row["cpu_baseline"] = cpu_baseline(name, prob, w, target_seconds=6.0)
            rows[name] = row
        except Exception as exc:                      # a missing code object must not take the headline line down
            rows[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    return rows


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (default: the workload's BASELINE batch)")
    ap.add_argument("--workload", default="lv", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true")
    ap.add_argument("--single-process", action="store_true",
                    help="drive the --gpus N devices from THIS process (one engine + host thread per device) instead "
                         "of one process per GPU")
    ap.add_argument("--devices", default="", help="--single-process: comma-separated device ordinals "
                                                   "(default 0..N-1; an ordinal may repeat: two engines on one GPU)")
    ap.add_argument("--force-dist", action="store_true",
                    help="form the process group (nccl = RCCL), barrier and device all-reduce even with one rank")
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside torchrun: become `python -m torch.distributed.run ... bench.py ...`."""
    import socket
    import torch
    if torch.cuda.device_count() < args.gpus:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, torch.cuda.device_count()))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    os.execv(sys.executable, cmd)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.single_process:
        out = run_single_process(args)
    else:
        if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            launch_ranks(args, argv)                  # does not return
        out = run_rank(args)
    if out is not None:
        if args.gpus == 1:
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(args.workload, make_problem(args.workload), WORKLOADS[args.workload])
            if not args.no_extra_configs:
                try:
                    out["host_api"] = host_api(args.workload, make_problem(args.workload), WORKLOADS[args.workload],
                                               args.batch or WORKLOADS[args.workload]["batch"])
                except Exception as exc:        # noqa: BLE001 -- the headline line must not depend on it
                    out["host_api"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            if not args.no_extra_configs and args.workload == "lv":
                out["configs"] = extra_configs(args)
            if not args.no_extra_configs:
                try:
                    out["first_use"] = first_use(args.workload)
                except Exception as exc:        # noqa: BLE001 -- the headline line must not depend on it
                    out["first_use"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        else:
            # the contract times the CPU baseline at N = 1 only; keep the pointer so an N > 1 line is self-describing
            out["cpu_baseline"] = {"value": None, "unit": "solves/s", "cores": usable_cores(), "kind": "port",
                                   "sample": "not timed at N > 1 (contract: rank 0, N = 1 only); see the N = 1 line of "
                                             "the same round (BENCH_rNN.json / profiles/rNN_bench.json)"}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
