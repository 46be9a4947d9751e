"""TEST INFRASTRUCTURE ONLY (see vo_oracle.h) — CPU restatement of the reference's track bookkeeping:
landmark / frame IDs and the mask-compaction constructors. Pure Python (small cases only).

Follows
  core/visual_odometry/landmark.h:64        inline static int landmark_counter_ = 0  (process-global)
  core/visual_odometry/landmark.cpp:5-52    both constructors: id_(landmark_counter_++), alive, tracked, !triangulated
  core/visual_odometry/landmark.cpp:65-74   setUntracked()
  core/visual_odometry/landmark.cpp:148-153 setDead()
  core/visual_odometry/landmark.cpp:194-231 LandmarkTracking(src, mask)
  core/visual_odometry/landmark.cpp:291-332 StereoLandmarkTracking(src, mask)
  core/visual_odometry/frame.h:53, frame.cpp:15,:35   id_ = frame_counter_++
  core/visual_odometry/frame.cpp:176-180    StereoFrame: left Frame first, then right
  core/visual_odometry/stereo_vo/stereo_vo.cpp:445 (StereoFrame per image pair), :538,:558,:571,:640,:670 (the five
  compactions of a steady-state frame), :716-736 (new landmarks, in candidate order, appended to lmtrack_final),
  :871-905 (first frame).
The reference holds no fixture for any of this: parity unpinned, the restatement is pinned by the
hand-worked example in tests/test_oracle.py::test_track_ids_hand_example.
"""
import numpy as np


class Process:
    """One process of the reference = one pair of static counters."""

    def __init__(self):
        self.landmark_counter = 0
        self.frame_counter = 0


class Landmark:
    def __init__(self, proc):
        self.id = proc.landmark_counter  # landmark.cpp:6 / :29
        proc.landmark_counter += 1
        self.alive = True
        self.tracked = True
        self.triangulated = False

    def set_untracked(self):  # landmark.cpp:65-74
        self.tracked = False

    def set_dead(self):  # landmark.cpp:148-153
        self.alive = False
        self.tracked = False


def new_frame_id(proc):  # frame.cpp:15 / :35
    i = proc.frame_counter
    proc.frame_counter += 1
    return i


def compact(lms, mask):
    """StereoLandmarkTracking(src, mask) / LandmarkTracking(src, mask): returns (index_valid, survivors)."""
    assert len(lms) == len(mask)  # the constructor throws otherwise (landmark.cpp:196-197, :293-295)
    index_valid = []
    for i, lm in enumerate(lms):
        if mask[i] and lm.alive and lm.tracked:
            index_valid.append(i)
        else:
            lm.set_untracked()
    return index_valid, [lms[i] for i in index_valid]


GATES = ("l0l1", "refine", "l1r1", "motion", "sampson")


def simulate_stereo_stream(seed, n_frames, n_first=24, n_cand=10, keep=(0.9, 0.93, 0.9, 0.88, 0.97), p_dead=0.05):
    """A stream of the stereo driver's bookkeeping with seeded random gate masks. Returns a list of per-frame
    dicts of plain int arrays: what the reference's data structures would hold, in a process of its own."""
    rng = np.random.default_rng(seed)
    proc = Process()
    frames = []
    # first frame (stereo_vo.cpp:445, :871-905): StereoFrame, then one landmark per accepted candidate, in order
    fid = [new_frame_id(proc), new_frame_id(proc)]
    accept = rng.random(n_first) < 0.8
    lms = [Landmark(proc) for a in accept if a]
    frames.append(dict(frame_ids=fid, accept=accept.astype(np.uint8), new_ids=[lm.id for lm in lms],
                       final_ids=[lm.id for lm in lms]))
    for _ in range(1, n_frames):
        f = {}
        # between frames the local BA may kill landmarks (setDead): they are still in the previous frame's list
        dead = rng.random(len(lms)) < p_dead
        for lm, d in zip(lms, dead):
            if d:
                lm.set_dead()
        f["dead"] = dead.astype(np.uint8)
        f["entry_ids"] = [lm.id for lm in lms]
        f["entry_alive"] = np.array([lm.alive for lm in lms], np.uint8)
        f["entry_tracked"] = np.array([lm.tracked for lm in lms], np.uint8)
        f["frame_ids"] = [new_frame_id(proc), new_frame_id(proc)]  # stereo_vo.cpp:445
        entry = lms
        cur = lms
        for g, kp in zip(GATES, keep):
            mask = rng.random(len(cur)) < kp
            idx, cur = compact(cur, mask)
            f["mask_" + g] = mask.astype(np.uint8)
            f["index_" + g] = idx
        f["exit_tracked"] = np.array([lm.tracked for lm in entry], np.uint8)
        accept = rng.random(n_cand) < 0.6  # mask_new[i] && Xl(2) > 0 && Xr(2) > 0 (:716-727)
        new = [Landmark(proc) for a in accept if a]
        f["accept"] = accept.astype(np.uint8)
        f["new_ids"] = [lm.id for lm in new]
        lms = cur + new  # lmtrack_final (:731-733), the next frame's lmtrack_prev (:466-472)
        f["final_ids"] = [lm.id for lm in lms]
        frames.append(f)
    return frames
