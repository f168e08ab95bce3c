#!/bin/bash
# A/B of two builds of libmtbt_hip.so on ONE box: the in-tree library against tools/probes/libmtbt_prev.so (an earlier commit built in a
# worktree), alternating, separate processes.  Through gpurun from the repo root; the swap only touches the box's scratch copy.
#   bash tools/lib_ab.sh [bench.py args...]
set -o pipefail
cd "$GRAFT_REPO_ROOT"
LIB=multitask_bonetumor_yolo_amd/csrc/libmtbt_hip.so
cp $LIB /tmp/lib_new.so
for round in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then cp tools/probes/libmtbt_prev.so $LIB; else cp /tmp/lib_new.so $LIB; fi
    timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" > /tmp/ab.log 2>&1 || { tail -5 /tmp/ab.log; cp /tmp/lib_new.so $LIB; exit 1; }
    python3 -c "
import json,sys
d=json.loads([l for l in open('/tmp/ab.log') if l.startswith('{')][-1])
print('$which', d['ms_per_step'], 'ms/step', d['value'], d['unit'])"
  done
done
cp /tmp/lib_new.so $LIB
