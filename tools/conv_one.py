#!/usr/bin/env python3
"""Run ONE conv configuration repeatedly (for rocprofv3 --pmc passes).  usage: conv_one.py NAME NBUF TC TP [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conv_tune import SHAPES, bench
name = [k for k in SHAPES if sys.argv[1] in k][0]
nb, tc, tp = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
print(name, bench(SHAPES[name], (nb << 28) | (tc << 16) | tp, iters=iters))
