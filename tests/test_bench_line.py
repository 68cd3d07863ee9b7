"""bench.py's bookkeeping that needs no GPU: the committed counter profile is only used when it describes the build,
batch and record format of the run at hand, and the truth-gradient error of the timed batch's first draws."""
import os

import numpy as np


def test_counter_profile_is_withheld_when_it_describes_another_build(monkeypatch):
    import bench
    ctx = {"batch": 65536, "record_bytes": 160, "code_object": "abc", "toolchain": "t1"}
    monkeypatch.setattr(bench, "pmc_profile", lambda w, k: ctx if k == "context" else {"fetch": 1.0, "write": 2.0})
    build = {"code_object": "abc", "toolchain": {"hash": "t1"}}
    assert bench.pmc_context_matches("lv", 65536, 160, build)[0]
    for kw, what in ((dict(B=4096), "batch"), (dict(rec=32), "record_bytes"),
                     (dict(build={"code_object": "zzz", "toolchain": {"hash": "t1"}}), "code_object"),
                     (dict(build={"code_object": "abc", "toolchain": {"hash": "t2"}}), "toolchain")):
        args = dict(B=65536, rec=160, build=build)
        args.update(kw)
        ok, note = bench.pmc_context_matches("lv", args["B"], args["rec"], args["build"])
        assert not ok and what in note
    monkeypatch.setattr(bench, "pmc_profile", lambda w, k: None)
    assert not bench.pmc_context_matches("lv", 65536, 160, build)[0]


def test_truth_gradient_error_of_the_first_draws(golden_dir):
    import bench
    t = np.load(os.path.join(golden_dir, "truth_lv.npz"))
    res = {"head_grads": t["grad_params"] * (1 + 1e-7), "head_lamda": -t["grad_y0"]}
    err = bench.truth_gradient_error("lv", res)
    assert err["draws"] == 16 and 5e-8 < err["dL_dp_max_rel"] < 2e-7 and err["dL_dy0_max_rel"] == 0.0
    assert bench.truth_gradient_error("seir", res) is None
