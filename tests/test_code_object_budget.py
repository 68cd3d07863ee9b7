"""Register / spill / scratch / LDS budget of the shipped code objects (profiles/code_object_budget.json,
tools/code_object_budget.py): the throughput of every configuration hangs on compiler flags and on the allocator's
mood (sunode_amd/_native.py *_CODEGEN_FLAGS), so a kernel edit or a toolchain update that turns SEIR's spill slots
into several hundred, or puts the Lotka-Volterra kernels back on scratch, must fail a CPU test -- not show up as a
slower bench line a round later."""
import json
import os

import pytest

from tools import code_object_budget as cob


def test_budget_file_covers_the_default_builds_of_every_baseline_problem():
    doc = cob.load()
    assert set(doc["budgets"]) == set(cob.BUILDS)
    for label, kernels in doc["budgets"].items():
        want = {"sa_k_forward", "sa_k_sens"} if label.endswith("/sens") else {"sa_k_forward", "sa_k_backward"}
        assert set(kernels) == want, label
        for k, row in kernels.items():
            assert set(row) == set(cob.FIELDS), (label, k)
    lv = doc["budgets"]["lv"]
    assert lv["sa_k_backward"]["vgpr_spill_count"] == 0 and lv["sa_k_backward"]["private_segment_fixed_size"] <= 64
    assert lv["sa_k_forward"]["vgpr_spill_count"] == 0
    assert doc["toolchain"]["hash"]


@pytest.mark.parametrize("label", ["lv", "robertson", "seir", "network24", "lv/sens", "robertson/sens", "seir/sens"])
def test_current_builds_stay_within_budget(label):
    """(cross-compiles for gfx950 on the CPU box; cached after the first build)"""
    hard, soft = cob.violations([label], split=True)
    assert hard == []               # spill slots / scratch / LDS (+ slack) and the occupancy class: ANY toolchain
    assert all("exceeds" in m or "informational" in m for m in soft)


def test_memory_resident_scratch_does_not_grow_with_the_state_count():
    """the controller's vector temporaries are workspace slots in bdf_mem.hip (bdf_core.h TMPV), not n-sized
    per-lane arrays: a 512-state model stays under a few KB of scratch per lane."""
    assert cob.mem_scratch_violations() == []


def test_occupancy_class_is_a_hard_limit(monkeypatch):
    from sunode_amd import _native
    assert cob.register_class({"vgpr_count": 128, "agpr_count": 0}) == 4
    assert cob.register_class({"vgpr_count": 256, "agpr_count": 0}) == 2
    assert cob.register_class({"vgpr_count": 200, "agpr_count": 57}) == 1
    doc = cob.load()
    fake = {"toolchain": {"hash": "another-toolchain"}, "budgets": {"lv": {
        k: dict(v, vgpr_count=100, agpr_count=0) for k, v in doc["budgets"]["lv"].items()}}}
    monkeypatch.setattr(cob, "load", lambda: fake)
    hard, soft = cob.violations(["lv"], split=True)          # recorded: <= 128 registers; built: more -> hard, even
    assert any("occupancy class" in m for m in hard)         # though the toolchain hash differs
    assert any("differs from the recorded one" in m for m in soft)


def test_budget_check_notices_a_regression(tmp_path, monkeypatch):
    """The check itself: a ceiling one below the current value is reported, kernel by kernel and field by field."""
    from sunode_amd import _native
    doc = cob.load()
    path = _native.build_code_object(cob.source_of("seir"), compact=True)
    notes = _native.code_object_notes(path)
    assert notes["sa_k_backward"]["vgpr_spill_count"] <= doc["budgets"]["seir"]["sa_k_backward"]["vgpr_spill_count"]
    tight = {"sa_k_backward": {"vgpr_count": notes["sa_k_backward"]["vgpr_count"] - 1},
             "sa_k_forward": {"group_segment_fixed_size": notes["sa_k_forward"]["group_segment_fixed_size"]}}
    bad = _native.check_code_object_budget("seir", path, tight)
    assert len(bad) == 1 and "sa_k_backward.vgpr_count" in bad[0]
    assert _native.check_code_object_budget("seir", path, {"sa_k_nonexistent": {}})
    tc = _native.toolchain_id()
    assert len(tc["hash"]) == 12 and "version" in tc["banner"].lower()
