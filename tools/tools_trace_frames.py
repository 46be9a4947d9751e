"""Kernel-by-kernel GPU timeline of a few consecutive frames from a rocprofv3 --kernel-trace CSV (frame = from one frame
kernel's start to the next one's): start, duration, queue.  usage: python tools/tools_trace_frames.py <csv> [first] [count] [all]
       python tools/tools_trace_frames.py <csv> busy [tail_ms]   how full the device is over the last tail_ms of the trace (S streams
                                                                  on one GPU: what overlaps, what each queue does, who holds the time)"""
import csv
import sys


def busy(path, tail_ms):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?")))
    rows.sort()
    t1 = max(r[1] for r in rows)
    t0 = t1 - int(tail_ms * 1e6)
    rows = [r for r in rows if r[0] >= t0]
    win = (t1 - t0) / 1e3
    ev = sorted([(s, 1) for s, _, _, _ in rows] + [(e, -1) for _, e, _, _ in rows])
    hist, depth, last = {}, 0, t0
    for t, d in ev:
        hist[depth] = hist.get(depth, 0) + (t - last)
        depth += d
        last = t
    print(f"window {win:.0f} us, {len(rows)} kernels; time with k kernels running: " +
          ", ".join(f"k={k}: {100.0 * v / (t1 - t0):.1f}%" for k, v in sorted(hist.items())))
    per_q, per_k = {}, {}
    for s, e, n, q in rows:
        per_q[q] = per_q.get(q, 0) + (e - s)
        c = per_k.setdefault(n, [0, 0])
        c[0] += e - s
        c[1] += 1
    print("per queue (sum of kernel durations / window): " + ", ".join(f"q{q}: {100.0 * v / (t1 - t0):.0f}%" for q, v in sorted(per_q.items())))
    for q in sorted(per_q):
        names = {}
        for s, e, n, qq in rows:
            if qq == q:
                names[n] = names.get(n, 0) + (e - s)
        print(f"   q{q}: " + ", ".join(f"{n[:22]} {100.0 * v / (t1 - t0):.0f}%" for n, v in sorted(names.items(), key=lambda kv: -kv[1])[:6]))
    n_frames = sum(c[1] for n, c in per_k.items() if "frame_track" in n)
    print(f"{n_frames} frame kernels in the window = {n_frames / (win * 1e-6):.0f} frames/s")
    for n, c in sorted(per_k.items(), key=lambda kv: -kv[1][0])[:24]:
        print(f"   {n[:44]:44s} {c[1]:6d} x {c[0] / c[1] / 1e3:8.1f} us = {100.0 * c[0] / (t1 - t0):5.1f}% of the window")


def main():
    path = sys.argv[1]
    if len(sys.argv) > 2 and sys.argv[2] == "busy":
        return busy(path, float(sys.argv[3]) if len(sys.argv) > 3 else 30.0)
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
