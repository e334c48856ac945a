#!/bin/bash
# parity subset + bench + I-cache counters on the current main library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4 || exit 1
python bench.py --steps 1 --warmup 0 --num-scales 200 --no-cpu-baseline --no-variants 2>/dev/null | tail -1 > gpurun_out/probe/bench5.json
python3 -c "
import json; d=json.loads(open('gpurun_out/probe/bench5.json').read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
BENCH="python3 bench.py --steps 1 --warmup 0 --num-scales 40 --no-cpu-baseline --no-roofline --no-variants"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/probe -o ic5 -- $BENCH > /dev/null 2> gpurun_out/probe/stderr_ic5.log
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/probe/ic5_counter_collection.csv')):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'unet_wg' in r['Kernel_Name']:
            per[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in per.items():
        print(k, 'n=', len(v), 'mean=', sum(v) / len(v))
PY
