"""Mean duration of each RUN of consecutive launches of the same (kernel, grid) in a rocprofv3 --kernel-trace CSV (dispatch order).
    python tools/ktrace_summary.py <dir with *_kernel_trace.csv> [name filter]"""
import csv, glob, os, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
runs = []
for r in rows:
    name = r["Kernel_Name"]
    if flt and flt not in name:
        continue
    key = (name[:64], r.get("Grid_Size_X", r.get("Grid_Size", "")))
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    if runs and runs[-1][0] == key and len(runs[-1][1]) < int(os.environ.get("RUN_LEN", "1000000")):
        runs[-1][1].append(us)
    else:
        runs.append((key, [us]))
for (name, grid), v in runs:
    v2 = v[1:] if len(v) > 2 else v
    print(f"{name:64s} grid {grid:>9s}  n {len(v):3d}  mean {sum(v2) / len(v2):8.1f} us  min {min(v):8.1f}")
