#!/bin/bash
# Quick per-kernel trace of the default bench (through gpurun from the repo root): gpurun_out/$1_kernel_stats.csv
set -o pipefail
TAG=${1:-kt}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/${TAG}_kt.log 2>&1
DB=$(find $OUT/${TAG}_kt -name "*.db" | head -1); [ -n "$DB" ] && python3 tools/rocpd_kernel_stats.py "$DB" > $OUT/${TAG}_kernel_stats.csv 2> $OUT/${TAG}_kernel_stats.txt
rm -rf $OUT/${TAG}_kt
tail -1 $OUT/${TAG}_kt.log | cut -c1-300
