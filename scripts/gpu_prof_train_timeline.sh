#!/bin/bash
# rocprofv3 kernel trace of the B=128 training step: per-step kernel time, GPU-busy time (union over both streams), wall time,
# launch count and the largest idle gaps (what is the step bound by?)
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/proft3; mkdir -p gpurun_out/proft3
export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/proft3 -o t -- python3 scripts/bench_train.py ${1:-128} ${2:-bf16} > gpurun_out/proft3/out.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/proft3/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# the 10 timed steps = the last 10/13 of the launches: find step boundaries by the optimizer kernel
idx = [i for i, e in enumerate(ev) if e[2].startswith("opt_adam_ema_kernel")]
print("steps seen", len(idx), "kernels", len(ev))
a, b = idx[-11], idx[-1]          # ten whole steps
seg = ev[a + 1:b + 1]
wall = (seg[-1][1] - seg[0][0]) / 10
ksum = sum(e[1] - e[0] for e in seg) / 10
busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
gaps = []
for s_, e_, n in seg[1:]:
    if s_ > cur_e:
        busy += cur_e - cur_s; gaps.append((s_ - cur_e, n)); cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
busy += cur_e - cur_s
print("per step: wall %.3f ms, kernel-time sum %.3f ms, GPU busy (union) %.3f ms, idle %.3f ms, launches %.1f" % (wall / 1e6, ksum / 1e6, busy / 10 / 1e6, (wall - busy / 10) / 1e6, len(seg) / 10))
gaps.sort(reverse=True)
print("largest idle gaps (us, kernel that ended the gap):")
for g, n in gaps[:12]:
    print("  %8.1f  %s" % (g / 1e3, n[:90]))
import collections
hist = collections.Counter(min(int(g / 1e3), 50) for g, n in gaps)
print("gap histogram (us -> count per step):", {k: round(v / 10, 1) for k, v in sorted(hist.items())})
tot_small = sum(g for g, n in gaps if g < 20e3) / 10 / 1e6
print("idle in gaps < 20 us: %.3f ms per step; in gaps >= 20 us: %.3f ms" % (tot_small, sum(g for g, n in gaps if g >= 20e3) / 10 / 1e6))
PY
tail -1 gpurun_out/proft3/out.log | cut -c1-200
