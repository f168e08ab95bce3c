#!/usr/bin/env python3
"""MFMA utilisation per 3x3 shape class from `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE` runs of
tools/conv_one.py (tools/profile_round.sh).  utilisation = MFMA busy cycles / (kernel cycles x 1024 SIMDs), kernel cycles =
GRBM_GUI_ACTIVE / 8 (summed over the 8 XCDs) -- the counter-based figure the north star's 0.70 target is read against."""
import csv, glob, os, sys
out, tag = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob(os.path.join(out, f"{tag}_sq_*"))):
    if not os.path.isdir(d):
        continue
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if not any(t in r["Kernel_Name"] for t in ("conv3x3_direct_kernel", "conv3x3_rr_kernel", "conv_igemm_kernel")):
                continue
            e = acc.setdefault((r["Kernel_Name"][:60], r["Counter_Name"]), [0, 0.0])
            e[0] += 1
            e[1] += float(r["Counter_Value"])
    kernels = sorted({k for k, _ in acc})
    for k in kernels:
        g = lambda c: acc.get((k, c), [1, 0.0])
        n = g("SQ_VALU_MFMA_BUSY_CYCLES")[0]
        busy, gui = g("SQ_VALU_MFMA_BUSY_CYCLES")[1] / n, g("GRBM_GUI_ACTIVE")[1] / max(g("GRBM_GUI_ACTIVE")[0], 1)
        cyc = gui / 8.0
        print(f"{os.path.basename(d):40s} {k:60s} dispatches {n:3d}  MFMA busy {busy:14.0f}  kernel cycles {cyc:10.0f}  utilisation {busy / (cyc * 1024) if cyc else 0:.3f}")
