"""Re-runs tests/golden/make_golden_ops.py's construction on the inputs stored in a fixture (shared by tests/test_golden_ops.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def build_case(d):
    import make_golden_ops as G
    c = tuple(int(d[k]) for k in ("H", "W", "C", "f", "s", "M", "R", "N", "S"))
    idx = G.CASES.index(c)
    out = G.build(c, bool(d["white"]), seed=900 + idx)
    for k in ("X", "Z", "q_mu", "q_sqrt", "w", "Y"):
        np.testing.assert_array_equal(out[k], d[k])     # the seeded inputs themselves
    return out
