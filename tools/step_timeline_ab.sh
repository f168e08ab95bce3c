#!/bin/bash
# Per-kernel timelines of ONE replayed inference step for several environment settings (A/B of scheduling knobs), through gpurun from the repo
# root:  bash tools/step_timeline_ab.sh "MTBT_SEG_GATE=0 MTBT_NODE_FUSED=0" "MTBT_SEG_GATE=1" ...   -> gpurun_out/step_timeline_<k>.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
k=0
for cfg in "$@"; do
  for kv in $cfg; do export "$kv"; done
  rm -rf gpurun_out/tl && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/tl_$k.log 2>&1
  CFG="$cfg" K=$k python3 - <<'PY'
import csv, glob, os
f = glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
stems = [i for i, r in enumerate(rows) if "stem" in r["Kernel_Name"]] + [len(rows)]
spans = [(int(rows[stems[k + 1] - 1]["End_Timestamp"]) - int(rows[stems[k]]["Start_Timestamp"]), k) for k in range(len(stems) - 1)]
k = min(spans)[1]
step = rows[stems[k]:stems[k + 1]]
t0 = int(step[0]["Start_Timestamp"])
out = open(f"gpurun_out/step_timeline_{os.environ['K']}.txt", "w")
out.write(f"# {os.environ['CFG']}\n")
for r in step:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    out.write(f"{s:9.1f} {e:9.1f} {e - s:7.1f} q{r.get('Queue_Id', '?'):>3s} {r['Kernel_Name'][:90]}\n")
end = max(int(r["End_Timestamp"]) for r in step)
print(f"cfg {os.environ['K']} [{os.environ['CFG']}]: {len(step)} kernels, step span {(end - t0) / 1e3:.1f} us")
PY
  for kv in $cfg; do unset "${kv%%=*}"; done
  k=$((k+1))
done
rm -rf gpurun_out/tl
