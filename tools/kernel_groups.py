"""GPU time per step by kernel group from a rocprofv3 kernel_stats.csv of `bench.py --steps K --warmup W` (the tables of DESIGN 5).
    python tools/kernel_groups.py <kernel_stats.csv> <steps incl. warm-up>"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
GROUPS = collections.OrderedDict([
    ("conv MFMA fwd / dgrad, NC8 (patch, stride-2, 3x3x3, gather)", r"conv_patch_nc8_kernel|conv_s2_dgrad_nc8|conv_gather_nc8"),
    ("conv MFMA wgrad, NC8", r"conv_wgrad_nc8_kernel"),
    ("conv MFMA fwd / dgrad, NCHW (igemm, patch3x3, Winograd)", r"conv_igemm_kernel|conv_patch3x3|conv_wino_kernel|conv_wino4_kernel"),
    ("reflect pad-ring GEMMs of the data gradient (conv_ring.hip, round 5)", r"reflect_ring_dgrad"),
    ("conv MFMA wgrad, NCHW (direct, wide bf16, Winograd)", r"conv_wgrad_wide|conv_wgrad_kernel|conv_wino_wgrad_kernel"),
    ("thin <= 4-channel heads (vector ALU)", r"conv_thin"),
    ("NCHW -> NC8 layout pass", r"nchw_to_nc8"),
    ("split-K / slab reductions", r"splitk_reduce|wgrad_reduce|wgrad_nc8_reduce|wino_slab_sum|wino_wgrad_finish|wino_wgrad_reduce"),
    ("norm statistics / apply / backward", r"norm_"),
    ("reflect folds", r"reflect_"),
    ("x2 up-sampling / resize / pooling", r"upsample|resize|maxpool"),
    ("weight packing, filter transforms, Adam", r"pack_|adam|wino_filter|wino4_filter|ring_pack"),
    ("act_bwd, tap backward, losses", r"act_bwd|relu_tap|grad_to_nc8|l1_|ssim|final_sum"),
    ("warp / raster / splat / RoI", r"flow_warp|warp_inv|splat|raster|roi_"),
    ("hipBLASLt / rocBLAS GEMMs", r"Cijk|rocblas"),
    ("ATen copies / cat / memcpy", r"direct_copy|copyBuffer|CatArray|bfloat16_copy|bfloat16tofloat32|Memcpy|copy_kernel"),
    ("ATen elementwise / reductions / fills", r"at::native"),
])
tot = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    for g, p in GROUPS.items():
        if re.search(p, r["Name"]):
            break
    else:
        g = "other"
    tot[g][0] += float(r["TotalDurationNs"])
    tot[g][1] += int(r["Calls"])
s = n = 0
for g in list(GROUPS) + ["other"]:
    if g in tot:
        ns, c = tot[g]
        print(f"| {g} | {ns / 1e6 / steps:.2f} ({c / steps:.0f}) |")
        s += ns; n += c
print(f"| total | {s / 1e6 / steps:.2f} ({n / steps:.0f}) |")
