"""Per-kernel averages of the SQ counter passes of tools/run_profiles.sh / run_pmc_sq.sh (counter_collection CSVs) -> profiles/<tag>_frame_sq_counters.json
usage: python tools/sq_summarize.py <dir with a/ and b/> <tag>"""
import csv
import glob
import json
import os
import sys


def load(d):
    per = {}
    for path in glob.glob(os.path.join(d, "*counter_collection.csv")):
        with open(path) as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"].split("(")[0]
                e = per.setdefault(k, {})
                c = e.setdefault(r["Counter_Name"], [0.0, 0])
                c[0] += float(r["Counter_Value"])
                c[1] += 1
    return per


def main():
    d, tag = sys.argv[1], sys.argv[2]
    out = {"units": "per launch averages; SQ cycle counters in quad-cycles (MI355X_MICROARCH.md)", "kernels": {}}
    for sub in ("a", "b"):
        for k, cs in load(os.path.join(d, sub)).items():
            if not any(x in k for x in ("frame_track", "frame_replay", "mono_track", "mono_replay", "gn_pose", "orb_tile", "orb_finish", "pyr_build", "svo_", "mvo_", "sba_")):
                continue
            e = out["kernels"].setdefault(k, {})
            for name, (tot, n) in cs.items():
                e[name] = round(tot / n, 1)
                e["launches"] = n
    for k, e in out["kernels"].items():
        if "SQ_WAVES" in e and "SQ_WAVE_CYCLES" in e and e["SQ_WAVES"]:
            e["wave_lifetime_quadcycles"] = round(e["SQ_WAVE_CYCLES"] / e["SQ_WAVES"], 1)
            for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_LDS_BANK_CONFLICT"):
                if c in e:
                    e["frac_" + c] = round(e[c] / e["SQ_WAVE_CYCLES"], 4)
    out["command"] = "rocprofv3 --pmc <8 SQ counters> --kernel-trace --output-format csv -- python3 bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 10 (two passes; tools/run_pmc_sq.sh)"
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from visual_odometry_ros_amd import build as VB
    mono = any("mono_track" in k for k in out["kernels"])
    out["kernel_source_sha"] = VB.kernel_source_sha(VB.MONO_KERNEL_SOURCES if mono else VB.FRAME_KERNEL_SOURCES)
    if len(sys.argv) > 3:
        out["command"] = sys.argv[3]
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_frame_sq_counters.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out)[:1500])


if __name__ == "__main__":
    main()
