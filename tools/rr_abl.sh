#!/bin/bash
# dev: ablation sweep of the direct 3x3 kernels (needs a -DMTBT_CONV_ABLATION build)
for shape in "proto.cv2" "${1:-c2f_p3.m}"; do
for pol in 7 15; do
for dbg in 0 1 4 8 9 5; do
  echo -n "pol $pol dbg $dbg: "
  MTBT_CONV_POLICY=$pol MTBT_CONV_DEBUG=$dbg python tools/conv_one.py "$shape" 0 0 0 20 2>&1 | tail -1
done; done; done
