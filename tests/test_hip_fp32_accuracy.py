"""Every exact-fp32 HIP op ON ITS OWN at the Code2 b256 / ER dimensions against a float64 evaluation of the same math,
with torch's fp32 evaluation of it as the yardstick (tools/fp32_accuracy.py): the HIP op's relative L2 error must be
within 3x of torch-fp32's, or below 1e-6.  This is the per-op half of the 1e-4 fp32 bar at real sizes; the model-level
half (tests/test_hip_configs.py) has to live with the chaos of ReLU gate flips, this one does not."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_every_fp32_op_is_at_fp32_roundoff():
    import fp32_accuracy

    rows = fp32_accuracy.all_cases()
    assert len(rows) >= 30
    bad = [(c, t, h, r) for c, t, h, r in rows if h > max(3 * r, 1e-6)]
    assert not bad, bad
