"""Where is the chip under-filled?  From a rocprofv3 kernel trace (csv) of `bench.py --steps K`: the last K steps' wall span, the
idle time between kernels, and the time during which ONLY kernels of < `small` workgroups run (nothing else overlapping) -- by
kernel name.  That last figure is what running a branch on a second stream, or fusing launches, can still win.
   python tools/timeline_occupancy.py <kernel_trace.csv> [steps=5] [small=256]"""
import csv, re, sys, collections

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
small = int(sys.argv[3]) if len(sys.argv) > 3 else 256
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        wg = max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), grid // wg, r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0"))))
rows.sort()
# steps: the raster kernel runs once per step (forward); the window is [raster of step -steps-1, raster of the last step)
marks = [i for i, r in enumerate(rows) if r[3].startswith("sparse_raster_kernel")]
assert len(marks) > steps, f"{len(marks)} steps in the trace, {steps} wanted"
rows = rows[marks[-steps - 1]:marks[-1]]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for k, (s, e, wgs, name, q) in enumerate(rows):
    ev.append((s, 1, k)); ev.append((e, 0, k))
ev.sort()
active = set()
idle = 0
only_small = collections.Counter()
only_small_n = collections.Counter()
big_alone = 0
overl = 0
prev = t0
for t, kind, k in ev:
    dt = t - prev
    if dt > 0:
        if not active:
            idle += dt
        else:
            if len(active) > 1:
                overl += dt
            if all(rows[a][2] < small for a in active):
                nm = re.sub(r"<.*", "", rows[min(active)][3])[:70]
                only_small[nm] += dt
            elif len(active) == 1:
                big_alone += dt
    prev = t
    if kind:
        active.add(k)
    else:
        active.discard(k)
for (s, e, wgs, name, q) in rows:
    if wgs < small:
        only_small_n[re.sub(r"<.*", "", name)[:70]] += 1
span = (t1 - t0) / 1e6
print(f"span {span / steps:.2f} ms/step   idle {idle / 1e6 / steps:.2f}   >=2 kernels in flight {overl / 1e6 / steps:.2f}   "
      f"only <{small}-workgroup kernels in flight {sum(only_small.values()) / 1e6 / steps:.2f} ms/step   launches/step {len(rows) / steps:.0f}")
streams = collections.Counter(r[4] for r in rows)
print("launches by stream/queue:", dict(streams))
print(f"-- time with only small kernels in flight, by (first) kernel, ms/step (launches/step of that kernel below {small} WGs)")
for nm, v in only_small.most_common(40):
    print(f"{v / 1e6 / steps:7.3f}  ({only_small_n[nm] / steps:5.0f})  {nm}")
