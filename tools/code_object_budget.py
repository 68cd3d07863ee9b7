"""Register / spill / scratch / LDS budget of the shipped code objects (performance hangs on compiler
switches -- sunode_amd/_native.py DEFAULT/WAVE/ADJOINT/SMALL_GROUP_CODEGEN_FLAGS -- so a toolchain update that turns
179 spill slots into 600 must not pass silently).

    python tools/code_object_budget.py            compare the current builds with profiles/code_object_budget.json
    python tools/code_object_budget.py --write    record the current builds as the budget (after a deliberate change)

The budget bounds, per kernel, vgpr_spill_count / sgpr_spill_count / private_segment_fixed_size (scratch bytes per
lane) / group_segment_fixed_size (static LDS) and the register counts; `build()` (__graft_entry__.py) and
tests/test_code_object_budget.py fail above it.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUDGET = os.path.join(ROOT, "profiles", "code_object_budget.json")

#: label -> (problem, build_code_object keyword arguments); the builds the solvers select by default for the
#: BASELINE problems (compact records from three states on) and the forward-sensitivity builds
BUILDS = {
    "lv": ("lv", {}),
    "robertson": ("robertson", {"compact": True}),
    "seir": ("seir", {"compact": True}),
    "network24": ("network24", {"compact": True}),
    "network100": ("network100", {"compact": True}),
    "lv/sens": ("lv", {"sens": True}),
    "robertson/sens": ("robertson", {"sens": True}),
    "seir/sens": ("seir", {"sens": True}),
}
KERNELS = ("sa_k_forward", "sa_k_backward", "sa_k_sens")
FIELDS = ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size",
          "vgpr_count", "agpr_count")


def source_of(problem):
    from tools.problem_cache import make_problem
    return make_problem(problem).native_source()


def current(labels=None):
    from sunode_amd import _native
    rows = {}
    for label in (labels or BUILDS):
        problem, kw = BUILDS[label]
        path = _native.build_code_object(source_of(problem), **kw)
        notes = _native.code_object_notes(path)
        want = ("sa_k_forward", "sa_k_sens") if kw.get("sens") else ("sa_k_forward", "sa_k_backward")
        rows[label] = {k: {f: notes[k][f] for f in FIELDS if f in notes[k]} for k in want}
    return rows


def load():
    with open(BUDGET) as fh:
        return json.load(fh)


#: fields whose excess fails the build (with SLACK on top of the recorded value); the register counts only inform
HARD_FIELDS = ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")
SLACK = {"vgpr_spill_count": 16, "sgpr_spill_count": 16, "private_segment_fixed_size": 64, "group_segment_fixed_size": 0}
#: environment overrides of the code generation: builds made under them are not the builds the budget describes
OVERRIDES = ("SA_CLANG_FLAGS", "SA_KERNEL_DEFINES", "SA_VGPR_LIVERANGE_OPT", "SA_WAVES_PER_EU", "SA_FORCE_GROUP")


def register_class(row):
    """Wavefronts per SIMD the unified register file (512 per lane) allows: vgpr + agpr <= 256 -> 2, <= 512 -> 1
    (the thresholds that decide occupancy; below 128 nothing in this code base gains)."""
    total = int(row.get("vgpr_count", 0)) + int(row.get("agpr_count", 0))
    return 4 if total <= 128 else (2 if total <= 256 else 1)


def violations(labels=None, split=False):
    """Messages of every budget violation of the current builds.  ``split=True``: (hard, soft).

    hard (fails ``build()`` and the tests, with ANY toolchain -- ADVICE r5: a toolchain whose hash differs from the
    recorded one is exactly the case the budget exists for): a spill / scratch / LDS ceiling (+ SLACK) exceeded, or a
    kernel falling into a lower occupancy class of the register file than recorded (<= 256 registers -> > 256);
    soft: exact register counts, and everything when codegen overrides are set in the environment (such builds are
    not the builds the budget describes)."""
    from sunode_amd import _native
    doc = load()
    overridden = any(os.environ.get(k) for k in OVERRIDES)
    hard, soft = [], []
    for label in (labels or doc["budgets"]):
        problem, kw = BUILDS[label]
        path = _native.build_code_object(source_of(problem), **kw)
        limits = doc["budgets"][label]
        strict = {k: {f: v + SLACK[f] for f, v in row.items() if f in HARD_FIELDS} for k, row in limits.items()}
        loose = {k: {f: v for f, v in row.items() if f not in HARD_FIELDS} for k, row in limits.items()}
        found = _native.check_code_object_budget(label, path, strict)
        notes = _native.code_object_notes(path)
        for kernel, row in limits.items():
            if kernel in notes and register_class(notes[kernel]) < register_class(row):
                found.append("%s: %s uses %d + %d registers: occupancy class %d wavefront(s) per SIMD, recorded %d"
                             % (label, kernel, notes[kernel].get("vgpr_count", 0), notes[kernel].get("agpr_count", 0),
                                register_class(notes[kernel]), register_class(row)))
        (soft if overridden else hard).extend(found)
        soft += _native.check_code_object_budget(label, path, loose)
    if doc["toolchain"]["hash"] != _native.toolchain_id()["hash"]:
        soft.append("toolchain %s differs from the recorded one (%s): register counts are informational, the spill / "
                    "scratch / LDS ceilings and the occupancy classes still bind"
                    % (_native.toolchain_id()["hash"], doc["toolchain"]["hash"]))
    if overridden:
        soft.append("codegen overrides in the environment: informational only")
    return (hard, soft) if split else hard + soft


#: memory-resident mapping: scratch bytes per lane must not grow with the number of states (after the fold onto
#: bdf_core.h every vector temporary of the controller was an n-sized per-lane array -- 11 / 15 KB per lane at n = 129,
#: ~100 KB at n ~ 1000; they are numbered slots of the HBM workspace now, bdf_core.h TMPV).  Checked on a 512-state
#: model against this ceiling: an O(n) array alone would be 8 * 512 = 4 096 bytes per temporary.
MEM_SCRATCH_PROBLEM = "chain512"
MEM_SCRATCH_CEILING = {"sa_k_forward": 4096, "sa_k_backward": 6144}


def mem_scratch_violations():
    from sunode_amd import _native
    src = source_of(MEM_SCRATCH_PROBLEM)
    assert _native.kernel_variant(src)[0] == "bdf_mem.hip"
    notes = _native.code_object_notes(_native.build_code_object(src))
    return ["%s: %s scratch %s bytes per lane exceeds %d (an n-sized per-lane array?)"
            % (MEM_SCRATCH_PROBLEM, k, notes[k]["private_segment_fixed_size"], lim)
            for k, lim in MEM_SCRATCH_CEILING.items() if notes[k]["private_segment_fixed_size"] > lim]


def main():
    from sunode_amd import _native
    if "--write" in sys.argv:
        doc = {"toolchain": _native.toolchain_id(),
               "note": "ceilings = the values of the builds that were measured (profiles/r04_*); regenerate with "
                       "`python tools/code_object_budget.py --write` after a deliberate kernel / flag change",
               "budgets": current()}
        with open(BUDGET, "w") as fh:
            json.dump(doc, fh, indent=1, sort_keys=True)
        print("wrote", BUDGET)
        return 0
    bad = violations() + mem_scratch_violations()
    for msg in bad:
        print("OVER BUDGET:", msg)
    tc = _native.toolchain_id()
    if tc["hash"] != load()["toolchain"]["hash"]:
        print("note: toolchain %s differs from the one the budget was recorded with (%s)" % (tc["hash"], load()["toolchain"]["hash"]))
    print("%d violation(s)" % len(bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
