#!/bin/bash
echo "stress lanes=4: $(timeout -k 10 400 python3 tools/lane_stress.py 200 4 2>&1 | grep steps)"
timeout -k 10 300 python tools/pair_stress.py 62,63,64,65,93,94 61,67,68,72,73,74,78,80,82,83,85 40 2>&1 | tail -5
