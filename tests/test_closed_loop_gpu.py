"""The operators composed into a closed-loop stereo odometry (examples/closed_loop_stereo.py): detection,
bucketing, stereo matching, the chained frame operator, pose chaining — trajectory against the renderer's
ground truth. A behavioural test, not a parity test: it guards the semantics of the outputs (stages, pixel
arrays, pose convention) that a caller depends on."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_closed_loop_stereo_follows_ground_truth(vo):
    import closed_loop_stereo as E
    r = E.run(n_frames=25)
    assert r["path_m"] > 15.0
    assert r["mean_tracked"] > 300 and r["mean_inliers"] > 0.7 * r["mean_tracked"]
    assert r["end_error_m"] < 0.02 * r["path_m"]  # < 2 % drift over the run
    assert r["max_step_error"] < 0.05
    assert all(e["new"] > 0 for e in r["log"][:5])  # bins freed by lost tracks are refilled


def test_closed_operator_gives_the_same_odometry_as_the_operator_calls(vo):
    """The same closed loop with steps [9] + [10] inside the frame operator (per-bin candidates found before the frame,
    tracked speculatively, emitted behind the BA; strict-border mode 4: the replay runs wherever the operator decides)
    and as three operator calls after every frame: every new landmark, hence every track set and every pose of 20
    frames, must be the same bits."""
    import numpy as np
    import closed_loop_stereo as E
    a = E.run(n_frames=21)
    b = E.run(n_frames=21, closed=True, strict_border=4)
    assert [e["new"] for e in a["log"]] == [e["new"] for e in b["log"]]
    assert [e["inliers"] for e in a["log"]] == [e["inliers"] for e in b["log"]]
    assert np.array_equal(a["T_wc"], b["T_wc"])
    assert b["end_error_m"] < 0.02 * b["path_m"]


def test_closed_loop_with_local_bundle_adjustment(vo):
    """Every third frame a stereo keyframe, the window bundle-adjusted through vo_sba_solve (the numeric part of
    SparseBAParameters around it is in the example): every solve lowers the window's reprojection error, keyframe
    poses stay at the millimetre level of the odometry, the trajectory still follows the ground truth."""
    import closed_loop_stereo as E
    r = E.run(n_frames=22, lba=True)
    assert len(r["lba"]) >= 5
    for e in r["lba"]:
        assert e["err_last"] < e["err_first"] and e["err_last"] < 1.0
        assert e["kf_pos_err_after"] < 1.5 * e["kf_pos_err_before"] + 1e-3
    assert r["end_error_m"] < 0.02 * r["path_m"]
