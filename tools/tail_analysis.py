"""What bounds a launch of ONE wavefront per SIMD?  (python tools/tail_analysis.py lv 65536 | robertson 262144 | seir 16384)

BASELINE's batch sizes put exactly 1 024 wavefronts of the forward / backward kernels on the 1 024 SIMDs of an MI355X
(LV: 65 536 lanes / 64; SEIR: 16 384 x 4 lanes / 64), one round, no second wavefront to fill a SIMD that has finished:
a launch lasts as long as its SLOWEST wavefront.  A wavefront's duration is (iterations of its attempt loop) x (time per
iteration), and the iteration count is a max over lanes -- in the forward kernel over the whole integration (lanes run
free), in the backward kernel per observation interval (lanes meet at every restart).  This tool reads the counters the
kernels already report (attempts per instance; the register kernel also reports the wave iterations of the backward
pass, stats[:, 15]) and prints

    mean attempts per instance        -> the work a perfectly packed machine would do
    mean / max iterations per wave    -> what the SIMDs are busy with / what the launch waits for
    kernel time / max iterations      -> time per wave iteration
    the hardest instance (forward + backward attempts): NO schedule of whole instances finishes before it does.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from sunode_amd import _native
    from sunode_amd.solver import AdjointSolver
    name, B = sys.argv[1], int(sys.argv[2])
    prob = bench.make_problem(name)
    w = bench.WORKLOADS[name]
    d = bench.make_batch(name, prob, B)                        # the batch of the bench line, same draws
    tol = dict(abstol=w["atol"], reltol=w["rtol"], backward_abstol=w["atol"], backward_reltol=w["rtol"],
               quad_abstol=w["atol"], quad_reltol=w["rtol"])
    if prob.n_remainder and d["pr"].shape[-1] != prob.n_remainder:     # AdjointSolver extends the remainder itself
        d["pr"] = d["pr"][..., :prob.n_remainder]
    sol = AdjointSolver(prob, **tol)
    tv = d["tvals"]
    for _ in range(3):
        y, st, sf = sol.solve_forward_batch(0.0, tv, d["y0"], d["ps"], d["pr"])
        g, lam, stb, sb = sol.solve_backward_batch(tv[-1], 0.0, tv, d["grads"])
    f_ms, b_ms = sol._engine().last_kernel_ms()
    fam = _native.kernel_variant(prob.native_source())
    lanes = fam[1]
    ipw = max(1, 64 // lanes)                                  # instances per wavefront (workgroup mapping: 1)
    nw = (B + ipw - 1) // ipw                                  # wavefronts that integrate (worker wavefronts not counted)
    slots = 1024 if lanes <= 32 else 256                       # SIMDs, or CUs for the one-workgroup-per-CU mapping
    rounds = nw / float(slots)
    af, ab = sf[:, 14].astype(float), sb[:, 14].astype(float)
    pad = nw * ipw - B
    wf = np.r_[af, np.zeros(pad)].reshape(nw, ipw).max(axis=1)         # forward: lanes run free -> max over the wave
    wb_lo = np.r_[ab, np.zeros(pad)].reshape(nw, ipw).max(axis=1)      # backward: at least the wave's hardest lane
    wb = sb[:, 15].astype(float)
    have_wi = bool(wb.max() > 0)
    wbw = np.r_[wb, np.zeros(pad)].reshape(nw, ipw).max(axis=1) if have_wi else wb_lo
    print("%s B = %d, %s, %d lane(s) per instance: %d integrating wavefronts of %d instance(s) on %d %s (%.2f rounds); "
          "forward %.3f ms, backward %.3f ms -> %.4g solves/s"
          % (name, B, fam[0], lanes, nw, ipw, slots, "SIMDs" if slots == 1024 else "CUs (one workgroup each)", rounds,
             f_ms, b_ms, B / ((f_ms + b_ms) * 1e-3)))
    note = " (sum over intervals of the max over lanes, device counter)" if have_wi else \
           " (>= max over lanes of the totals: this mapping does not report the per-interval count)"
    for tag, a, wv, ms, nt in (("forward ", af, wf, f_ms, " (max over lanes)"), ("backward", ab, wbw, b_ms, note)):
        # one round: the launch waits for its slowest wavefront; R rounds: every slot runs ~R wavefronts one after another
        serial = wv.max() if rounds <= 1.0 else max(wv.max(), rounds * wv.mean())
        print("  %s: attempts per instance mean %.1f, p99 %.0f, max %.0f | wave iterations%s mean %.1f, max %.0f | "
              "iterations a slot runs one after another ~%.0f -> %.2f us per wave iteration | lanes busy %.2f of the launch"
              % (tag, a.mean(), np.percentile(a, 99), a.max(), nt, wv.mean(), wv.max(), serial, 1e3 * ms / serial,
                 a.mean() * rounds / serial if rounds > 1.0 else a.mean() / serial))
    if rounds <= 1.0:
        crit = (af + ab).max()
        t_it = (f_ms + b_ms) * 1e3 / (wf.max() + wbw.max())
        print("  one round: a launch lasts as long as its slowest wavefront.  Hardest instance: %.0f forward + backward "
              "attempts = %.2f ms at %.2f us per iteration, of the %.2f ms of the two launches -- no placement of whole "
              "instances finishes before it; every wavefront at the MEAN iteration count would need %.2f ms"
              % (crit, crit * t_it * 1e-3, t_it, f_ms + b_ms, (wf.mean() + wbw.mean()) * t_it * 1e-3))


if __name__ == "__main__":
    main()
