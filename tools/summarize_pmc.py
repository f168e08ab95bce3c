#!/usr/bin/env python3
"""Per-launch HBM-side traffic of the conv kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: summarize_pmc.py FETCH_DIR WRITE_DIR STEPS_PROFILED > profiles/rNN_traffic.json
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE counts
128-byte requests at 64 bytes, so wide coalesced reads (16 B per lane: every read of these kernels) are DOUBLED; WRITE_SIZE
is exact for 16-byte-per-lane stores.  Infinity-Cache hits are included (memory-side of L2), so this is an upper bound on HBM bytes."""
import csv, glob, json, os, sys


def total(d, counter):
    n, s = 0, 0.0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and ("conv_igemm_kernel" in r["Kernel_Name"] or "conv3x3_direct_kernel" in r["Kernel_Name"] or "conv3x3_rr_kernel" in r["Kernel_Name"]):
                n += 1
                s += float(r["Counter_Value"])
    return n, s


nf, fetch_kib = total(sys.argv[1], "FETCH_SIZE")
nw, write_kib = total(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
out = {"kernel": "conv_igemm_kernel + conv3x3_direct_kernel (all tiles)", "dispatches_fetch_pass": nf, "dispatches_write_pass": nw,
       "fetch_bytes_per_launch": 2.0 * fetch_kib * 1024 / max(nf, 1), "write_bytes_per_launch": write_kib * 1024 / max(nw, 1),
       "correction": "FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B), WRITE_SIZE x1; KiB units; Infinity-Cache hits included",
       "launches_per_step": nf / max(steps, 1)}
out["traffic_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]
json.dump(out, sys.stdout, indent=1)
print()
