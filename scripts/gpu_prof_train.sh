#!/bin/bash
# rocprofv3 kernel stats of the B=128 training step (13 steps: 3 warm + 10 timed): sum of kernel time vs wall time per step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/proft2
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/proft2 -o t -- python3 scripts/bench_train.py ${1:-128} ${2:-f32} > gpurun_out/proft2/out.log 2>&1
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/proft2/t_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 2), "calls", calls, "-> per step (13 steps): kernel ms", round(tot / 1e6 / 13, 3), "launches", round(calls / 13, 1))
for r in rows[:16]:
    print(r["Name"][:72].ljust(72), r["Calls"].rjust(6), str(round(float(r["TotalDurationNs"]) / 1e6, 2)).rjust(8), "ms  avg us", round(float(r["AverageNs"]) / 1e3, 1))
PY
tail -1 gpurun_out/proft2/out.log | cut -c1-200
