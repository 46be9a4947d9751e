"""T12: landmark / frame IDs and the mask-compaction constructors (landmark.cpp:5-52, :194-231, :291-332;
landmark.h:64; frame.h:53). CPU part: the oracle's restatement against a hand-worked example and the committed
fixture. GPU part (-m gpu): the product — vo_compact_tracks + the per-context counters — driven with the fixture's
masks for TWO streams interleaved in one process must reproduce the IDs each stream has in a process of its own."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "track_ids.npz")


def test_track_ids_hand_example():
    """Worked by hand from landmark.cpp: 4 landmarks, a mask ctor, a dead landmark, a second ctor, new landmarks."""
    from oracle import tracks as T
    proc = T.Process()
    assert [T.new_frame_id(proc), T.new_frame_id(proc)] == [0, 1]  # StereoFrame: left, right
    lms = [T.Landmark(proc) for _ in range(4)]
    assert [lm.id for lm in lms] == [0, 1, 2, 3]
    idx, cur = T.compact(lms, [1, 0, 1, 1])
    assert idx == [0, 2, 3] and [lm.id for lm in cur] == [0, 2, 3]
    assert [lm.tracked for lm in lms] == [True, False, True, True]  # the rejected one is setUntracked()
    cur[1].set_dead()  # id 2
    idx, cur2 = T.compact(cur, [1, 1, 0])
    assert idx == [0] and [lm.id for lm in cur2] == [0]  # id 2 is dead although its mask is true; id 3 masked out
    assert [lm.tracked for lm in lms] == [True, False, False, False]
    # an untracked landmark never comes back, whatever the mask says
    idx, _ = T.compact(lms, [1, 1, 1, 1])
    assert idx == [0]
    new = [T.Landmark(proc) for _ in range(2)]
    assert [lm.id for lm in new] == [4, 5] and proc.landmark_counter == 6
    assert [T.new_frame_id(proc), T.new_frame_id(proc)] == [2, 3]


def test_fixture_is_what_the_generator_makes():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_track_ids as M
    want = M.flatten()
    have = np.load(GOLD)
    assert sorted(want) == sorted(have.files)
    for k in want:
        assert np.array_equal(want[k], have[k]), k
    # the two streams are independent processes: both start at landmark 0 / frame 0
    assert have["s0_f0_frame_ids"].tolist() == [0, 1] == have["s1_f0_frame_ids"].tolist()
    assert have["s0_f0_new_ids"][0] == 0 == have["s1_f0_new_ids"][0]
    assert int(have["n_frames"]) >= 5


def _stage_from_index_lists(g, s, k, n):
    """stage[i] = number of gates entry feature i passed (what the fused frame operator reports)."""
    from oracle import tracks as T
    stage = np.zeros(n, np.int32)
    cur = np.arange(n)
    for lvl, gate in enumerate(T.GATES, 1):
        cur = cur[g[f"s{s}_f{k}_index_{gate}"]]
        stage[cur] = lvl
    return stage


@pytest.mark.gpu
def test_two_interleaved_streams_keep_their_own_ids(vo):
    from oracle import tracks as T
    g = np.load(GOLD)
    nf = int(g["n_frames"])
    ctxs = [vo.Context(device=0, max_width=64, max_height=64, max_points=256, n_slots=2, max_level=1) for _ in range(2)]
    try:
        tr = [vo.TrackIds(c) for c in ctxs]
        state = [None, None]  # per stream: (ids, alive, tracked) of the list a frame hands to the next
        for k in range(nf):
            for s in (0, 1):  # the two streams take turns, frame by frame, in ONE process
                key = f"s{s}_f{k}_"
                assert np.array_equal(tr[s].newFrames(2), g[key + "frame_ids"])
                if k == 0:
                    ids = tr[s].newLandmarks(g[key + "accept"])
                    assert np.array_equal(ids[ids >= 0], g[key + "new_ids"])
                    ids = ids[ids >= 0]
                    state[s] = (ids, np.ones(len(ids), np.uint8), np.ones(len(ids), np.uint8))
                    continue
                ids, alive, tracked = state[s]
                assert np.array_equal(ids, g[key + "entry_ids"])
                dead = g[key + "dead"].astype(bool)  # setDead() between the frames (landmark.cpp:148-153)
                alive = np.where(dead, 0, alive).astype(np.uint8)
                tracked = np.where(dead, 0, tracked).astype(np.uint8)
                assert np.array_equal(alive, g[key + "entry_alive"]) and np.array_equal(tracked, g[key + "entry_tracked"])
                n = len(ids)
                # (a) the five constructors one after the other, as the reference runs them
                cur_ids, cur_alive, cur_tr = ids, alive, tracked
                pos = np.arange(n)
                exit_tracked = tracked.copy()
                for gate in T.GATES:
                    idx, tr_after, ids_out = tr[s].compactTracks(g[key + "mask_" + gate], cur_alive, cur_tr, cur_ids)
                    assert np.array_equal(idx, g[key + "index_" + gate]), (s, k, gate)
                    exit_tracked[pos] = tr_after
                    pos, cur_ids, cur_alive, cur_tr = pos[idx], ids_out, cur_alive[idx], tr_after[idx]
                assert np.array_equal(exit_tracked, g[key + "exit_tracked"])
                # (b) one constructor on the frame operator's stage bytes gives the same survivors and flags
                stage = _stage_from_index_lists(g, s, k, n)
                idx_b, tr_b, ids_b = tr[s].compactTracks(stage == len(T.GATES), alive, tracked, ids)
                assert np.array_equal(idx_b, pos) and np.array_equal(ids_b, cur_ids)
                assert np.array_equal(tr_b, g[key + "exit_tracked"])
                new = tr[s].newLandmarks(g[key + "accept"])
                assert np.array_equal(new[new >= 0], g[key + "new_ids"])
                final = np.concatenate([cur_ids, new[new >= 0]])
                assert np.array_equal(final, g[key + "final_ids"])
                state[s] = (final, np.ones(len(final), np.uint8), np.ones(len(final), np.uint8))
        # the counters are the contexts': stream 0's never saw stream 1's landmarks
        for s in (0, 1):
            nl, nfr = tr[s].peek()
            made = max(int(g[f"s{s}_f{k}_new_ids"].max()) for k in range(nf) if g[f"s{s}_f{k}_new_ids"].size) + 1
            assert nfr == 2 * nf and nl == made
        tr[0].reset(100, 10)
        assert tr[0].newLandmarks([1, 0, 1]).tolist() == [100, -1, 101] and tr[0].newFrames(1).tolist() == [10]
        assert tr[1].peek()[1] == 2 * nf  # untouched by the other context's reset
    finally:
        for c in ctxs:
            c.close()
