"""Constants of csrc/rt_device.h that stand for an expression of the reference, recomputed (no GPU)."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _define(name):
    src = open(os.path.join(ROOT, "ray_tracing_in_one_weekend_amd", "csrc", "rt_device.h")).read()
    return int(re.search(r"#define\s+%s\s+(0x[0-9a-fA-F]+)u" % name, src).group(1), 16)


def test_near_one_interval_is_the_reference_expression_over_every_float():
    """math.rs:13-15 `(v.length() - 1.0).abs() < 1e-6` with length = sqrt(dot) (glam), both correctly rounded in f32: the squared
    lengths that pass are one interval of floats, and near_one() compares with its two ends instead of taking the root."""
    f = np.float32
    lo_bits, hi_bits = np.array([0.0], dtype=f).view(np.uint32)[0], np.array([np.inf], dtype=f).view(np.uint32)[0]
    passing = []
    with np.errstate(all="ignore"):
        for start in range(int(lo_bits), int(hi_bits) + 1, 1 << 24):  # every non-negative float, inf included
            x = np.arange(start, min(start + (1 << 24), int(hi_bits) + 1), dtype=np.uint32).view(f)
            ok = np.abs((np.sqrt(x) - f(1.0)).astype(f)) < f(1e-6)
            passing.append(x[ok].view(np.uint32))
    passing = np.concatenate(passing)
    assert len(passing) == 50 and np.all(np.diff(passing.astype(np.int64)) == 1)        # one interval
    assert int(passing[0]) == _define("RT_NEAR_ONE_LO") and int(passing[-1]) == _define("RT_NEAR_ONE_HI")
    # (negative arguments and NaN: sqrt is NaN, the comparison is false; near_one's two comparisons are false for NaN as well,
    # and a squared length is never negative)
