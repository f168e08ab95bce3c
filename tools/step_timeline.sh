#!/bin/bash
# Timeline of ONE replayed inference step (lanes as they ran): rocprofv3 kernel trace of a short bench, last step's kernels with start / end
# relative to the step's first kernel.  Through gpurun from the repo root; writes gpurun_out/step_timeline.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tl && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/tl.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last stem kernel marks the start of the last step
stems = [i for i, r in enumerate(rows) if "stem" in r["Kernel_Name"]] + [len(rows)]
# the graph replays are the shortest stem-to-stem spans; take the last complete one
spans = [(int(rows[stems[k + 1] - 1]["End_Timestamp"]) - int(rows[stems[k]]["Start_Timestamp"]), k) for k in range(len(stems) - 1)]
k = min(spans)[1]
step = rows[stems[k]:stems[k + 1]]
t0 = int(step[0]["Start_Timestamp"])
out = open("gpurun_out/step_timeline.txt", "w")
for r in step:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    out.write(f"{s:9.1f} {e:9.1f} {e - s:7.1f} q{r.get('Queue_Id', '?'):>3s} {r['Kernel_Name'][:90]}\n")
print(len(step), "kernels, step span", (int(step[-1]["End_Timestamp"]) - t0) / 1e3, "us")
PY
rm -rf gpurun_out/tl
