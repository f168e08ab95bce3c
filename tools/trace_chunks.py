#!/usr/bin/env python3
"""Median kernel duration per consecutive chunk of N launches in a rocprofv3 kernel trace csv: trace_chunks.py CSV N"""
import csv, sys, statistics
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
rows = [r for r in rows if "conv" in r["Kernel_Name"] or "dwconv" in r["Kernel_Name"]]
for i in range(0, len(rows), n):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i:i + n]]
    print(f"chunk {i // n:3d}: median {statistics.median(d):7.2f} us  min {min(d):7.2f}  n={len(d)}")
