"""Host-side helpers of the product layer against the oracle's restatement (no GPU)."""
import numpy as np

from oracle import i3d_ref as R
from step_amd.tube_math import generate_anchors


def test_generate_anchors_matches_oracle():
    a = generate_anchors()
    assert a.shape == (34, 4)                          # 9 + 25 boxes of anchor mode "1" (data/data_utils.py:19-45)
    assert np.array_equal(a, R.anchors())
    assert (a >= 0).all() and (a <= 1 + 1e-6).all()
