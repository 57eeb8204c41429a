"""Post-processing of tools/prof_round2.sh: per-kernel HBM traffic + GB/s, conv traffic per launch, MFMA-busy fractions."""
import collections, csv, glob, json, os, sys
O = sys.argv[1]


def counters(d, names):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in names:
                out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def short(n):
    return n.split("(")[0].replace("void ", "")[:60]


# ---- calibration: FETCH_SIZE / WRITE_SIZE are reported in KiB
cal = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for k, v in counters(f"{O}/cal_{C}", {C}).items():
        vals = v[C]
        if "flow_warp_fwd" in k:
            cal[("warp", C)] = sum(vals[1:]) / max(len(vals) - 1, 1) * 1024
        if "vectorized_elementwise_kernel" in k or "copyBuffer" in k or "direct_copy" in k:
            big = [x for x in vals if x * 1024 > 1e8]
            if big:
                cal[("copy", C)] = sum(big) / len(big) * 1024
copy_bytes, warp_img, warp_flow = 64 * 1024 * 1024 * 4, 40 * 64 * 64 * 128 * 4, 40 * 2 * 64 * 128 * 4
calib = {
    "copy_16B_per_lane": {"true_read_bytes": copy_bytes, "FETCH_SIZE_bytes": cal.get(("copy", "FETCH_SIZE")),
                          "true_write_bytes": copy_bytes, "WRITE_SIZE_bytes": cal.get(("copy", "WRITE_SIZE"))},
    "flow_warp_fwd_4B_per_lane": {"true_read_bytes": warp_img + warp_flow, "FETCH_SIZE_bytes": cal.get(("warp", "FETCH_SIZE")),
                                  "true_write_bytes": warp_img, "WRITE_SIZE_bytes": cal.get(("warp", "WRITE_SIZE"))},
}
f16 = (copy_bytes / calib["copy_16B_per_lane"]["FETCH_SIZE_bytes"]) if calib["copy_16B_per_lane"]["FETCH_SIZE_bytes"] else None
f4 = ((warp_img + warp_flow) / calib["flow_warp_fwd_4B_per_lane"]["FETCH_SIZE_bytes"]) \
    if calib["flow_warp_fwd_4B_per_lane"]["FETCH_SIZE_bytes"] else None

# ---- kernel durations from the trace (5 + 2 steps), counters from the 2 + 1 step passes -> per launch averages
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f"{O}/kernel_trace.csv")):
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
fetch = counters(f"{O}/pmc_FETCH_SIZE", {"FETCH_SIZE"})
write = counters(f"{O}/pmc_WRITE_SIZE", {"WRITE_SIZE"})
rows = []
for k in dur:
    if k not in fetch or k not in write:
        continue
    rd = sum(fetch[k]["FETCH_SIZE"]) / len(fetch[k]["FETCH_SIZE"]) * 1024
    wr = sum(write[k]["WRITE_SIZE"]) / len(write[k]["WRITE_SIZE"]) * 1024
    t = sum(dur[k]) / len(dur[k])
    rows.append((sum(dur[k]) / 7.0, short(k), len(dur[k]) / 7.0, t, rd, wr))
rows.sort(reverse=True)
with open(f"{O}/hbm_per_kernel.txt", "w") as f:
    f.write("# per launch: FETCH_SIZE / WRITE_SIZE (KiB counters -> bytes), raw and with the measured read calibration\n")
    f.write(f"# calibration on this box: FETCH_SIZE x {f16 and round(f16, 2)} for 16-B/lane streaming reads (torch copy), x {f4 and round(f4, 2)} for "
            "4-B/lane coalesced gathers (flow_warp_fwd); WRITE_SIZE exact.  The conv MFMA kernels read with 4-B/lane gathers / 16-B loads mixed.\n")
    f.write("ms/step  kernel  launches/step  avg_us  read_MB(raw)  write_MB  GB/s(raw)  GB/s(read x2)\n")
    for ms, name, n, t, rd, wr in rows[:60]:
        f.write(f"{ms * 1e3:7.3f}  {name:60s} {n:6.1f} {t * 1e6:8.1f} {rd / 1e6:9.2f} {wr / 1e6:9.2f} "
                f"{(rd + wr) / t / 1e9:8.0f} {(2 * rd + wr) / t / 1e9:8.0f}\n")
conv = [r for r in rows if any(s in r[1] for s in ("conv_wino_kernel", "conv_wino4_kernel", "conv_igemm_kernel", "conv_patch3x3", "conv_wgrad_kernel",
                                                   "conv_wino_wgrad_kernel"))]
# the pad-ring launch of a reflect data gradient (round 5) belongs to the Winograd launch it follows: bench.py times the two as ONE
# conv launch, so its bytes go into the numerator and its launches stay out of the count
ring = [r for r in rows if "reflect_ring_dgrad_kernel" in r[1]]
nl = sum(r[2] for r in conv)
rd = sum(r[4] * r[2] for r in conv + ring) / nl
wr = sum(r[5] * r[2] for r in conv + ring) / nl
# the launch count bench.py's own profiler sees for this build (its staleness check compares against it)
bench_launches = None
try:
    for line in open(f"{O}/bench_final.json"):
        if line.startswith("{"):
            bench_launches = json.loads(line)["roofline"]["launches_per_step"]
except Exception:
    pass
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --steps 2 --warmup 1` (tools/prof_round.sh); "
               "counters are KiB; traffic_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction for wide reads, "
               f"measured on this box: x{f16 and round(f16, 2)} for 16-B/lane reads, x{f4 and round(f4, 2)} for 4-B/lane gathers); "
               "launch-weighted mean over all conv MFMA kernels",
       "calibration": calib,
       "conv": {"launches_per_step": bench_launches if bench_launches is not None else round(nl, 1),
                "launches_per_step_in_trace": round(nl, 1), "read_bytes_per_launch_raw": round(rd), "write_bytes_per_launch": round(wr),
                "traffic_bytes_per_launch": round(2 * rd + wr), "traffic_bytes_per_launch_uncorrected": round(rd + wr)}}
json.dump(res, open(f"{O}/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res["conv"]), "calib", f16, f4)

# ---- MFMA busy
m = counters(f"{O}/pmc_MFMA", {"GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
                               "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"})
with open(f"{O}/mfma_busy.txt", "w") as f:
    f.write("# per kernel (mean over launches of bench.py --steps 2 --warmup 1): MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x\n"
            "# kernel cycles per XCD), kernel cycles = GRBM_GUI_ACTIVE / 8\n")
    f.write("kernel  launches  mfma_busy_frac  wave_wait_frac(SQ_WAIT_ANY/SQ_WAVE_CYCLES)  issue_stall_frac(SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES)\n")
    out = []
    for k, v in m.items():
        if "conv_" not in k and "reflect_ring" not in k:
            continue
        mean = {c: sum(x) / len(x) for c, x in v.items()}
        gui = mean.get("GRBM_GUI_ACTIVE", 0) / 8
        if gui <= 0:
            continue
        out.append((sum(v["GRBM_GUI_ACTIVE"]), short(k), len(v["GRBM_GUI_ACTIVE"]), mean["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / gui,
                    mean["SQ_WAIT_ANY"] / max(mean["SQ_WAVE_CYCLES"], 1), mean["SQ_WAIT_INST_ANY"] / max(mean["SQ_WAVE_CYCLES"], 1)))
    for _, name, n, busy, w, s in sorted(out, reverse=True):
        f.write(f"{name:62s} {n:5d} {busy:7.3f} {w:7.3f} {s:7.3f}\n")
print(open(f"{O}/mfma_busy.txt").read()[:1500])
