#!/bin/bash
# rocprofv3 kernel trace of the B=128 training step, split by hardware queue: which stream carries the critical path?
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/proft4; mkdir -p gpurun_out/proft4
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/proft4 -o t -- python3 scripts/bench_train.py ${1:-128} ${2:-bf16} > gpurun_out/proft4/out.log 2>&1
python3 - <<'PY' > gpurun_out/proft4/streams.txt
import csv, glob, collections
f = glob.glob("gpurun_out/proft4/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("columns", list(rows[0].keys()))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows)
idx = [i for i, e in enumerate(ev) if e[2].startswith("opt_adam_ema_kernel")]
a, b = idx[-11], idx[-1]
seg = ev[a + 1:b + 1]
wall = (seg[-1][1] - seg[0][0]) / 10
print("per step wall %.3f ms, launches %.1f" % (wall / 1e6, len(seg) / 10))
for key_i, key_n in ((3, "queue"), (4, "stream")):
    per = collections.defaultdict(lambda: [0, 0])
    for e in seg:
        per[e[key_i]][0] += e[1] - e[0]; per[e[key_i]][1] += 1
    for k, (t, n) in sorted(per.items()):
        print("%s %s: kernel time %.3f ms per step, %.1f launches per step" % (key_n, k, t / 10 / 1e6, n / 10))
# per queue: kernel-name totals
for q in sorted(set(e[3] for e in seg)):
    agg = collections.defaultdict(lambda: [0, 0])
    for e in seg:
        if e[3] == q:
            agg[e[2][:60]][0] += e[1] - e[0]; agg[e[2][:60]][1] += 1
    print("--- queue", q)
    for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
        print("   %8.1f us/step %5.1f launches  %s" % (t / 10 / 1e3, c / 10, n))
PY
cat gpurun_out/proft4/streams.txt
tail -1 gpurun_out/proft4/out.log | cut -c1-300
