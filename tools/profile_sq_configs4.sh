#!/bin/bash
# MFMA-busy counters of the 3x3 shape classes at BASELINE configs[4]'s size (batch 64, 1280 x 1280 input: maps twice as wide, fp16), through gpurun
# from the repo root:  bash tools/profile_sq_configs4.sh r03   ->  gpurun_out/r03_c4_mfma_busy.txt
set -o pipefail
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
export MTBT_TUNE_BATCH=64 MTBT_TUNE_SCALE=2 MTBT_TUNE_DTYPE=f16
for shape in "proto.cv2" "c2f_p3.m" "head 3x3 256->64 @80" "c2f_p4.m" "bifpn.m 3x3 128->128 @40"; do
  s=$(echo "$shape" | tr -c 'a-zA-Z0-9' '_')
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_c4_sq_$s -- python3 tools/conv_one.py "$shape" 0 0 0 10 > $OUT/${TAG}_c4_sq_$s.log 2>&1
done
python3 tools/summarize_sq.py $OUT ${TAG}_c4 > $OUT/${TAG}_c4_mfma_busy.txt 2>&1
rm -rf $OUT/${TAG}_c4_sq_*/ 2>/dev/null
cat $OUT/${TAG}_c4_mfma_busy.txt
