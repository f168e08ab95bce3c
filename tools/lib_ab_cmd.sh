#!/bin/bash
# Run one command under the previous (tools/probes/libmtbt_prev.so) and the current build of libmtbt_hip.so on ONE box, alternating.
#   bash tools/lib_ab_cmd.sh python3 tools/mlp384_probe.py
set -o pipefail
cd "$GRAFT_REPO_ROOT"
LIB=multitask_bonetumor_yolo_amd/csrc/libmtbt_hip.so
cp $LIB /tmp/lib_new.so
for which in prev new prev new; do
  if [ $which = prev ]; then cp tools/probes/libmtbt_prev.so $LIB; else cp /tmp/lib_new.so $LIB; fi
  echo "== $which"
  timeout -k 10 300 "$@" 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_new.so $LIB
