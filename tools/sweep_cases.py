"""The (n states, p differentiated parameters) shapes of the parity sweep over the DEFAULT mapping
(tests/test_shape_sweep.py; pre-compiled by ``__graft_entry__.build()``).

The reference accepts any sympy system (/root/reference/sunode/symode/problem.py:25-33; shapes exercised in
/root/reference/sunode/test_solve.py:7-78).  Here every model is a new code object and the kernel family is chosen
by (n, p) -- sunode_amd/_native.py ``kernel_variant`` / ``lane_group_size``: register kernel n <= 5 and p <= 8 |
4-lane lean groups n <= 16 | 8-lane lean groups n <= 21 | G = 8..32 lanes with the matrix in LDS up to 64 |
workgroup per instance up to 128 | memory-resident beyond.  The sweep sits on both sides of every boundary and
inside every band, with few / as many / more parameters than states.
"""

#: (problem name of tools/problem_cache.py, batch size) -- adjoint (forward + backward) through the default build
ADJOINT_CASES = [
    ("lv12", 70),          # n = 2, p = 12: few states, many parameters -> 4-lane groups with idle lanes
    ("rn5_8", 70),         # the register kernel's largest shape
    ("rn5_9", 70),         # p 8 | 9: leaves the register kernel
    ("rn6_1", 70), ("rn6_33", 40),          # n 5 | 6; p = 33 > 32: 8 lanes although n = 6
    ("rn7_4", 70), ("rnb7_9", 70),
    ("rn12_4", 70), ("rn12_17", 40),
    ("rnb15_9", 70), ("rn15_1", 70),
    ("rn16_4", 70),                         # the 4-lane groups' last size
    ("rn17_4", 70), ("rn17_33", 24),        # 16 | 17: 8-lane lean groups
    ("rnb20_9", 40),
    ("rn21_17", 40),                        # the lean groups' last size
    ("rn22_1", 40), ("rn22_33", 16),        # 21 | 22: matrix in LDS, G = 16 / 32
    ("rn33_4", 24), ("rnb33_9", 24),
    ("rn48_17", 12),
    ("rn64_1", 8), ("rn64_33", 6),          # the lane groups' last size
    # workgroup per instance: one workgroup per CU (LDS), 256 CUs -> batches beyond 256 so that workgroup slots are
    # taken a second time within one launch (the r5 batches of 3-5 instances never left the first round)
    ("rn65_4", 300),                        # 64 | 65: workgroup per instance
    ("rn96_9", 300),
    ("rn127_4", 260),
    ("rn128_17", 260),                      # the workgroup's last size
    ("rn129_1", 40),                        # 128 | 129: memory-resident
    ("chain256", 24),                       # far beyond it (bidiagonal Jacobian: zero runs emitted as loops, codegen.ZERO_RUN_MIN)
]

#: forward sensitivities (``Solver(sens_mode=...)``) through the default build of the same shapes
SENS_CASES = [
    ("lv12", 40), ("rn5_9", 40), ("rn6_1", 40), ("rn7_4", 40), ("rn12_4", 24), ("rnb15_9", 24), ("rn17_4", 24),
    ("rn21_17", 12), ("rn22_1", 24),
]


def batch_of(name, B):
    """Inputs of case ``name`` (tools/problems.py generators)."""
    import re
    from tools import problems as P
    if name == "lv12":
        return P.lv12_batch(B)
    m = re.fullmatch(r"chain(\d+)", name)
    if m:
        return P.chain_batch(B, int(m.group(1)))
    m = re.fullmatch(r"rnb?(\d+)_(\d+)", name)
    return P.random_network_batch(B, int(m.group(1)), int(m.group(2)))
