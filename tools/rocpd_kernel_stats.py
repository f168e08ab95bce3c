#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, min, max) from the rocpd database this rocprofv3 build writes for
`--kernel-trace`.  usage: rocpd_kernel_stats.py RESULTS.db > profiles/rNN_kernel_stats.csv"""
import csv, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for n, k, s, a, mn, mx in rows:
    w.writerow([n, k, s, round(a, 3), round(100 * s / tot, 2), mn, mx])
conv = [r for r in rows if any(t in r[0] for t in ("conv_igemm_kernel", "conv3x3_direct_kernel", "conv3x3_rr_kernel"))]
print(f"conv family: {sum(r[1] for r in conv)} calls, average {sum(r[2] for r in conv) / max(sum(r[1] for r in conv), 1) / 1e3:.2f} us", file=sys.stderr)
