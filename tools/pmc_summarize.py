"""Turns the two rocprofv3 --pmc passes of tools/run_profiles.sh (FETCH_SIZE, WRITE_SIZE; counter_collection CSVs)
into profiles/<tag>_frame_pmc.json: HBM bytes per launch of the dominant kernel and per frame over all kernels, with the
unit / gfx950 corrections calibrated by tools/pmccal.hip on this access pattern (1 GiB moved once with one dword per
lane: FETCH_SIZE counts KiB/2 for such loads, WRITE_SIZE KiB — the same x2 the microarchitecture guide gives).
usage: python tools/pmc_summarize.py <pmc_dir> <tag> <frames> [kernel substring]"""
import csv
import json
import os
import sys


def load(path):
    per, n = {}, {}
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
            n[k] = n.get(k, 0) + 1
    return per, n


def cal_factor(d, name, gib_kib):
    p = os.path.join(d, f"cal_{name}", "cal_counter_collection.csv")
    if not os.path.exists(p):
        return None
    per, _ = load(p)
    v = max(per.values())
    return gib_kib / v


def main():
    d, tag, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
    kern = sys.argv[4] if len(sys.argv) > 4 else "frame_track_kernel"
    ff = cal_factor(d, "FETCH_SIZE", 1048576.0) or 1.9999790193851346  # r01 calibration if pmccal was not run
    wf = cal_factor(d, "WRITE_SIZE", 1048576.0) or 1.0
    f_per, f_n = load(os.path.join(d, "bench_FETCH_SIZE", "bench_counter_collection.csv"))
    w_per, w_n = load(os.path.join(d, "bench_WRITE_SIZE", "bench_counter_collection.csv"))
    k = [x for x in f_per if kern in x][0]
    out = {
        "kernel": k.split("(")[0], "launches": f_n[k],
        "FETCH_SIZE_KiB_avg": f_per[k] / f_n[k], "WRITE_SIZE_KiB_avg": w_per[k] / w_n[k],
        "calibration": {"pattern": "one dword per lane, 1 GiB moved once (tools/pmccal.hip)", "fetch_factor": ff, "write_factor": wf,
                        "unit": "KiB"},
        "fetch_bytes_per_launch": f_per[k] / f_n[k] * 1024 * ff, "write_bytes_per_launch": w_per[k] / w_n[k] * 1024 * wf,
    }
    out["hbm_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]
    out["all_kernels_bytes_per_frame"] = {"fetch": sum(f_per.values()) * 1024 * ff / frames, "write": sum(w_per.values()) * 1024 * wf / frames,
                                          "note": f"sum over every kernel of the run / {frames} frames"}
    out["per_kernel_KiB_per_frame"] = {x.split("(")[0][:48]: {"fetch": round(f_per[x] * ff / frames, 1), "write": round(w_per.get(x, 0.0) * wf / frames, 1)}
                                       for x in sorted(f_per, key=lambda q: -f_per[q])[:14]}
    out["command"] = "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python3 bench.py --no-cpu-baseline --strict-border 1 --no-secondary --steps 60 --warmup 10 (two passes; counter collection serialises kernels across queues, hence the stream-ordered replay; tools/run_profiles.sh)"
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from visual_odometry_ros_amd import build as VB
    # bench.py quotes these numbers only while the kernel's sources still hash to this
    out["kernel_source_sha"] = VB.kernel_source_sha(VB.MONO_KERNEL_SOURCES if "mono_track" in kern else VB.FRAME_KERNEL_SOURCES)
    if len(sys.argv) > 5:
        out["command"] = sys.argv[5]
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_frame_pmc.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out)[:600])


if __name__ == "__main__":
    main()
