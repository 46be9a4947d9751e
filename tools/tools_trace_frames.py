"""Kernel-by-kernel GPU timeline of a few consecutive frames from a rocprofv3 --kernel-trace CSV (frame = from one frame
kernel's start to the next one's): start, duration, queue.  usage: python tools/tools_trace_frames.py <csv> [first] [count]"""
import csv
import sys


def main():
    path = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    skip = ("__amd_rocclr",) if len(sys.argv) > 4 and sys.argv[4] == "all" else ("pyr_", "orb_", "bucket_", "__amd_rocclr", "remap_")
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""),
                         r.get("Queue_Id", "?")))
    rows.sort()
    tracks = [i for i, r in enumerate(rows) if "frame_track" in r[2] or "mono_track" in r[2]]
    for fi in range(first, min(first + count, len(tracks) - 1)):
        a, b = tracks[fi], tracks[fi + 1]
        t0 = rows[a][0]
        print(f"--- frame {fi}: next frame kernel starts at +{(rows[b][0] - t0) / 1e3:.1f} us")
        for s, e, n, q in rows[a:b]:
            if any(k in n for k in skip):
                continue
            print(f"   q{q:>3s} {n[:40]:40s} +{(s - t0) / 1e3:8.1f}  ->{(e - t0) / 1e3:8.1f}   {(e - s) / 1e3:7.1f} us")


if __name__ == "__main__":
    main()
