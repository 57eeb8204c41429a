"""Post-processing of tools/prof_cfg.sh: MFMA-pipe busy fraction per conv kernel from the --pmc pass."""
import collections, csv, glob, sys
O = sys.argv[1]
m = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/pmc_MFMA/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for k, v in m.items():
    if "conv_" not in k or "GRBM_GUI_ACTIVE" not in v:
        continue
    mean = {c: sum(x) / len(x) for c, x in v.items()}
    gui = mean["GRBM_GUI_ACTIVE"] / 8
    if gui <= 0:
        continue
    out.append((sum(v["GRBM_GUI_ACTIVE"]), k.split("(")[0].replace("void ", "")[:70], len(v["GRBM_GUI_ACTIVE"]),
                mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / gui, mean.get("SQ_WAIT_ANY", 0) / max(mean.get("SQ_WAVE_CYCLES", 1), 1),
                mean.get("SQ_WAIT_INST_ANY", 0) / max(mean.get("SQ_WAVE_CYCLES", 1), 1)))
with open(f"{O}/mfma_busy.txt", "w") as f:
    f.write("# per kernel (mean over the launches of bench.py --config K --steps 2 --warmup 1): MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES /\n"
            "# (1024 SIMDs x kernel cycles per XCD), kernel cycles = GRBM_GUI_ACTIVE / 8\n")
    f.write("kernel  launches  mfma_busy_frac  wave_wait_frac(SQ_WAIT_ANY/SQ_WAVE_CYCLES)  issue_stall_frac(SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES)\n")
    for _, name, n, busy, w, s in sorted(out, reverse=True):
        f.write(f"{name:72s} {n:5d} {busy:7.3f} {w:7.3f} {s:7.3f}\n")
print(open(f"{O}/mfma_busy.txt").read()[:2500])
