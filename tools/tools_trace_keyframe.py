"""Kernel-by-kernel GPU timeline of the first frame with a local-BA solve at or after a given frame, from a rocprofv3
--kernel-trace CSV: start, duration and the gap to the previous kernel of the same queue (where a keyframe's time goes).
usage: python tools/tools_trace_keyframe.py <kernel_trace.csv> [first_frame]"""
import csv
import sys


def main():
    path = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""),
                         r.get("Queue_Id", "?")))
    rows.sort()
    tracks = [i for i, r in enumerate(rows) if "frame_track" in r[2] or "mono_track" in r[2]]
    for fi in range(first, len(tracks) - 1):
        a, b = tracks[fi], tracks[fi + 1]
        if not any("sba_solve" in rows[k][2] for k in range(a, b)):
            continue
        t0 = rows[a][0]
        print(f"--- frame {fi}: next frame kernel starts at +{(rows[b][0] - t0) / 1e3:.1f} us")
        last_end = {}
        for s, e, n, q in rows[a:b]:
            gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
            last_end[q] = max(e, last_end.get(q, 0))
            print(f"   q{q:>3s} {n[:40]:40s} +{(s - t0) / 1e3:8.1f}  {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}")
        break


if __name__ == "__main__":
    main()
