#!/usr/bin/env python3
"""Print the kernel timeline of one steady-state frame from a rocprofv3 kernel_trace.csv."""
import csv, sys, glob
path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob("gpurun_out/**/*kernel_trace.csv", recursive=True))[-1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "pad_level0" in r["Kernel_Name"]]
i0, i1 = idx[which], idx[which + 1]
t0 = int(rows[i0]["Start_Timestamp"]); prev = None; busy = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1000 if prev else 0
    busy += (e - s)
    print(f"{(s-t0)/1000:9.1f} us  dur {(e-s)/1000:8.1f}  gap {gap:7.1f}  {r['Kernel_Name'][:48]:48s} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']} vgpr {r['VGPR_Count']} lds {r['LDS_Block_Size']}")
    prev = e
print(f"frame span {(int(rows[i1]['Start_Timestamp'])-t0)/1000:.1f} us, busy {busy/1000:.1f} us")
