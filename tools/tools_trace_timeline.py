"""Per-frame GPU timeline from a rocprofv3 --kernel-trace CSV: start / end of every kernel of a frame relative to the
frame kernel's start (which stream runs what, and where the gaps are).
usage: python tools/tools_trace_timeline.py <kernel_trace.csv> [first_frame] [n_frames]"""
import csv
import sys


def short(name):
    for k in ("frame_track", "frame_replay", "frame_gate", "frame_fallback", "gn_pose", "pad_level0", "pyr_down", "orb_resize",
              "orb_score", "orb_count", "orb_plan", "orb_emit", "orb_harris", "orb_select", "orb_output", "bucket_key",
              "bucket_table", "remap_level0"):
        if k in name:
            return k
    return name[:24]


def main():
    path = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
    rows.sort()
    tracks = [i for i, r in enumerate(rows) if r[2] == "frame_track"]
    for fi in range(first, first + count):
        if fi + 1 >= len(tracks):
            break
        t0 = rows[tracks[fi]][0]
        tn = rows[tracks[fi + 1]][0]
        print(f"--- frame {fi}: next frame kernel starts at +{(tn - t0) / 1e3:.1f} us")
        agg = {}
        for s, e, n, q in rows:
            if t0 - 1000 <= s < tn - 1000:
                key = (n, q)
                a = agg.setdefault(key, [s, e, 0])
                a[0], a[1], a[2] = min(a[0], s), max(a[1], e), a[2] + 1
        for (n, q), (s, e, k) in sorted(agg.items(), key=lambda kv: kv[1][0]):
            print(f"   {n:16s} q{q:>3s} x{k:<3d} {(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} us")


if __name__ == "__main__":
    main()
