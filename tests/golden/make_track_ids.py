"""Generates tests/golden/track_ids.npz: two stereo streams of track bookkeeping (gate masks, compaction index
lists, landmark and frame IDs over 7 frames), each simulated by oracle/tracks.py as the reference would run it in a
process of its own. Run from the repo root:  python tests/golden/make_track_ids.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tracks as T  # noqa: E402

N_FRAMES = 7
SEEDS = (11, 12)


def flatten():
    out = {"n_frames": np.int32(N_FRAMES), "seeds": np.array(SEEDS, np.int32)}
    for s, seed in enumerate(SEEDS):
        for k, f in enumerate(T.simulate_stereo_stream(seed, N_FRAMES)):
            for key, v in f.items():
                dt = np.uint8 if key.startswith(("mask_", "accept", "dead", "entry_alive", "entry_tracked",
                                                 "exit_tracked")) else np.int32
                out[f"s{s}_f{k}_{key}"] = np.asarray(v, dt)
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "track_ids.npz"), **flatten())
    print("wrote tests/golden/track_ids.npz")
