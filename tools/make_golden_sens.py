"""Truth fixtures for forward sensitivities: dy(t_k)/dp for a few instances of the LV and Robertson
batches, from the sensitivity equations integrated by scipy at 1e-13 (same machinery as
make_golden_truth.py).  Output: tests/golden/truth_sens_{lv,robertson}.npz with
y0, ps, pr, t0, tvals, y_out [B,n_t,n], sens [B,n_t,p,n]."""
import os
import sys

import numpy as np
from scipy.integrate import solve_ivp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.make_golden_truth import GOLD, augmented_rhs, make  # noqa: E402
from tools.problems import lv_batch, robertson_batch  # noqa: E402


def sens_truth(prob, y0, ps, pr, t0, tvals, method, rtol, atol):
    n, p = prob.n_states, prob.n_params
    rhs = augmented_rhs(prob)
    B = len(y0)
    y_out = np.zeros((B, len(tvals), n))
    sens = np.zeros((B, len(tvals), p, n))
    for b in range(B):
        prb = pr if pr.ndim == 1 else pr[b]
        z0 = np.concatenate([y0[b], np.zeros(n * p), np.eye(n).ravel()])
        sol = solve_ivp(rhs, (t0, tvals[-1]), z0, method=method, t_eval=tvals, args=(ps[b], prb), rtol=rtol, atol=atol)
        assert sol.success, sol.message
        z = sol.y.T
        y_out[b] = z[:, :n]
        sens[b] = z[:, n:n + n * p].reshape(len(tvals), p, n)
        print("  instance", b, "nfev", sol.nfev, flush=True)
    return y_out, sens


def main():
    prob = make("lv")
    d = lv_batch(4)
    ps = d["params"][:, prob.params_subset.subset_index]
    pr = d["params"][:, prob.params_subset.remainder_index]
    y, S = sens_truth(prob, d["y0"], ps, pr, d["t0"], d["tvals"], "DOP853", 1e-13, 1e-15)
    np.savez(os.path.join(GOLD, "truth_sens_lv.npz"), y0=d["y0"], ps=ps, pr=pr, t0=d["t0"], tvals=d["tvals"],
             y_out=y, sens=S)
    prob = make("robertson")
    d = robertson_batch(2)
    y, S = sens_truth(prob, d["y0"], d["params"], np.zeros((2, 0)), d["t0"], d["tvals"], "Radau", 1e-12, 1e-16)
    np.savez(os.path.join(GOLD, "truth_sens_robertson.npz"), y0=d["y0"], ps=d["params"], pr=np.zeros((2, 0)),
             t0=d["t0"], tvals=d["tvals"], y_out=y, sens=S)


if __name__ == "__main__":
    main()
