#!/bin/bash
# rocprofv3 kernel trace of the training step (configs[2]) on the GPU box: per-kernel calls / total / average into gpurun_out/$1_train_kernel_stats.csv
set -o pipefail
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
MTBT_TRAIN_LANES=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_tkt -- python3 bench.py --mode train --steps 5 --warmup 2 > $OUT/${TAG}_train_kt.log 2>&1
DB=$(find $OUT/${TAG}_tkt -name "*.db" | head -1); [ -n "$DB" ] && python3 tools/rocpd_kernel_stats.py "$DB" > $OUT/${TAG}_train_kernel_stats.csv 2> /dev/null
rm -rf $OUT/${TAG}_tkt
timeout -k 10 400 python3 bench.py --mode train --steps 10 --warmup 3 > $OUT/${TAG}_train_bench.json 2> $OUT/${TAG}_train_bench.err
# HBM-side traffic per kernel family of the training step: two PMC passes (FETCH_SIZE, WRITE_SIZE), 3 plan executions each (1 warm-up + 1 step + the instrumented replay;
# single stream).  tools/summarize_pmc.py --train applies the guide's gfx950 corrections
MTBT_TRAIN_LANES=0 timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_tpmc_f -- python3 bench.py --mode train --steps 1 --warmup 1 > $OUT/${TAG}_tpmc_f.log 2>&1
MTBT_TRAIN_LANES=0 timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_tpmc_w -- python3 bench.py --mode train --steps 1 --warmup 1 > $OUT/${TAG}_tpmc_w.log 2>&1
python3 tools/summarize_pmc.py --train $OUT/${TAG}_tpmc_f $OUT/${TAG}_tpmc_w 3 > $OUT/${TAG}_train_traffic.json
rm -rf $OUT/${TAG}_tpmc_f $OUT/${TAG}_tpmc_w
head -8 $OUT/${TAG}_train_kernel_stats.csv | cut -c1-200; cut -c1-300 $OUT/${TAG}_train_bench.json
