# HBM-side traffic of the conv MFMA kernels during bench steps (separate --pmc passes, as the MI355X guide prescribes):
#   tools/pmc_traffic.sh  ->  gpurun_out/pmc_traffic.json   (copy to profiles/r01_pmc_traffic.json)
set -e
export TMPDIR=/tmp
R=$PWD
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$C
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_$C.log 2>&1 || tail -5 gpurun_out/pmc_$C.log
done
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{C}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != C:
            continue
        n = r["Kernel_Name"]
        key = "igemm" if ("conv_igemm_kernel" in n or "conv_patch3x3_kernel" in n or "conv_wino_kernel" in n) else \
              "wgrad" if "conv_wgrad_kernel" in n else None
        if key:
            agg[key][0] += 1
            agg[key][1] += float(r["Counter_Value"])
    out[C] = {k: {"launches": v[0], "sum_counter": v[1]} for k, v in agg.items()}
res = {"note": "rocprofv3 --pmc, bench.py --steps 2 --warmup 1 (3 steps counted); FETCH_SIZE/WRITE_SIZE are in KiB; "
               "gfx950 correction: FETCH_SIZE x2 for wide reads is NOT applied (4-byte-per-lane gathers are uncalibrated), "
               "so read traffic is a lower bound between 1x and 2x of the figure"}
for k in ("igemm", "wgrad"):
    n = out["FETCH_SIZE"][k]["launches"]
    rd = out["FETCH_SIZE"][k]["sum_counter"] * 1024 / n
    wr = out["WRITE_SIZE"][k]["sum_counter"] * 1024 / out["WRITE_SIZE"][k]["launches"]
    res[k] = {"launches": n, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
              "traffic_bytes_per_launch": round(rd + wr)}
json.dump(res, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
