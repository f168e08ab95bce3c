#!/bin/bash
timeout -k 10 120 python -X faulthandler tools/capture_topo.py viamain > gpurun_out/topo_viamain.log 2>&1; echo "topo viamain rc=$? $(tail -1 gpurun_out/topo_viamain.log)"
for cfg in "3 60" "4 60" "3 1e9" "4 1e9" "6 1e9"; do set -- $cfg
  MTBT_LANES=$1 MTBT_LANE_WIDE_US=$2 timeout -k 10 120 python -X faulthandler tools/capture_probe.py step > gpurun_out/cap_$1_$2.log 2>&1; echo "lanes $1 wide $2 rc=$?"
done
