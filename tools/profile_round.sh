#!/bin/bash
# One profiling round on the GPU box (run through gpurun from the repo root): bench line, per-kernel trace, HBM traffic counters, MFMA-busy
# counters of the 3x3 shape classes.  The traced / counted runs skip the head calibration (--raw-heads: its one train-mode forward would add
# ~270 conv dispatches of other shapes to the statistics) and the schedule autotune (plans of other options); kernel durations do not depend on either.  Everything lands under gpurun_out/$1_*; copy what is to be judged into profiles/.
set -o pipefail
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
timeout -k 10 400 python3 bench.py --kernel-table > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err && cp $OUT/layer_times.json $OUT/${TAG}_layer_times.json
MTBT_LANES=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-autotune --raw-heads > $OUT/${TAG}_kt.log 2>&1
DB=$(find $OUT/${TAG}_kt -name "*.db" | head -1); [ -n "$DB" ] && python3 tools/rocpd_kernel_stats.py "$DB" > $OUT/${TAG}_kernel_stats.csv 2> $OUT/${TAG}_kernel_stats.txt
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_f -- python3 bench.py --no-graph --steps 1 --warmup 1 --no-cpu-baseline --raw-heads > $OUT/${TAG}_pmc_f.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_w -- python3 bench.py --no-graph --steps 1 --warmup 1 --no-cpu-baseline --raw-heads > $OUT/${TAG}_pmc_w.log 2>&1
python3 tools/summarize_pmc.py $OUT/${TAG}_pmc_f $OUT/${TAG}_pmc_w 6 > $OUT/${TAG}_traffic.json
for shape in "proto.cv2" "c2f_p3.m" "head 3x3 256->64 @80" "c2f_p4.m"; do
  s=$(echo "$shape" | tr -c 'a-zA-Z0-9' '_')
  timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_sq_$s -- python3 tools/conv_one.py "$shape" 0 0 0 20 > $OUT/${TAG}_sq_$s.log 2>&1
done
python3 tools/summarize_sq.py $OUT ${TAG} > $OUT/${TAG}_mfma_busy.txt 2>&1
rm -rf $OUT/${TAG}_kt $OUT/${TAG}_pmc_f $OUT/${TAG}_pmc_w $OUT/${TAG}_sq_*/ 2>/dev/null
tail -3 $OUT/${TAG}_mfma_busy.txt; cat $OUT/${TAG}_bench.json | cut -c1-400
