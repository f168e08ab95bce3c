#!/usr/bin/env python3
"""Per-launch HBM-side traffic of the kernel families of the forward plan from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: summarize_pmc.py FETCH_DIR WRITE_DIR PLAN_EXECUTIONS > profiles/rNN_traffic.json
PLAN_EXECUTIONS = how many times the profiled command ran the launch plan (bench.py --no-graph: warm-ups + timed steps + 1 result step +
3 instrumented replays), so that launches_per_step is per PLAN execution.
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE counts
128-byte requests at 64 bytes, so wide coalesced reads (16 B per lane: every read of these kernels) are DOUBLED; WRITE_SIZE
is exact for 16-byte-per-lane stores.  Infinity-Cache hits are included (memory-side of L2), so this is an upper bound on HBM bytes."""
import csv, glob, json, os, sys

FAMILIES = {"conv": ("conv_igemm_kernel", "conv3x3_direct_kernel", "conv3x3_rr_kernel", "upconv_fused_kernel", "node_gemm_kernel", "pw_stream_kernel"), "dwconv": ("dwconv_kernel",),
            "mlp_fused": ("mlp_fused_kernel", "mlp_pair_kernel"), "bifpn_fuse": ("fuse_kernel",), "layernorm": ("layernorm_kernel",), "stem": ("stem_mfma_kernel", "stem_kernel"),
            "mask_assembly": ("mask_x4_kernel",)}
if "--train" in sys.argv:      # families of the training step (bench.py --mode train): the names bench.py's roofline_families uses
    sys.argv.remove("--train")
    FAMILIES = {"wgrad": ("wgrad_kernel", "wgrad_f32_kernel", "wgrad_reduce_kernel", "stem_wgrad_kernel", "wgrad3x3"), "conv_fwd_dgrad": ("conv_igemm_kernel", "conv3x3_rr_kernel", "conv3x3_direct_kernel", "pw_stream_kernel"),
                "batchnorm": ("bn_block_stats", "bn_combine_stats", "bn_apply", "bn_bwd_partial", "bn_bwd_final", "bn_bwd_apply", "bn_stats_from", "bn_copy_stats", "colsum_tile"),
                "depthwise": ("dwconv_kernel", "dw_wgrad"), "layernorm": ("layernorm_kernel", "layernorm_bwd"), "channel_sums": ("channel_sum",),
                "weight_prep": ("weight_prep_kernel",)}


def totals(d, counter):
    out = {k: [0, 0.0] for k in FAMILIES}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for fam, pats in FAMILIES.items():
                if any(p in r["Kernel_Name"] for p in pats):
                    out[fam][0] += 1
                    out[fam][1] += float(r["Counter_Value"])
                    break
    return out


fetch, write = totals(sys.argv[1], "FETCH_SIZE"), totals(sys.argv[2], "WRITE_SIZE")
execs = int(sys.argv[3])
fams = {}
for k in FAMILIES:
    nf, fk = fetch[k]
    nw, wk = write[k]
    if not nf or not nw:
        continue
    fb, wb = 2.0 * fk * 1024 / nf, wk * 1024 / nw
    fams[k] = {"dispatches_fetch_pass": nf, "dispatches_write_pass": nw, "launches_per_step": nf / execs, "fetch_bytes_per_launch": fb,
               "write_bytes_per_launch": wb, "traffic_bytes_per_launch": fb + wb}
out = {"kernel": "conv_igemm_kernel + conv3x3_rr_kernel + upconv_fused_kernel + node_gemm_kernel + pw_stream_kernel (all tiles)", "plan_executions": execs,
       "correction": "FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B), WRITE_SIZE x1; KiB units; Infinity-Cache hits included",
       "families": fams}
out.update({k: v for k, v in fams.get("conv", {}).items()})
json.dump(out, sys.stdout, indent=1)
print()
