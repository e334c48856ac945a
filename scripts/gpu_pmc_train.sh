#!/bin/bash
# PMC passes over the B=128 training step (bench_train.py: 3 warm + 10 timed steps): SQ busy / MFMA busy, instruction-cache, HBM bytes.
cd $GRAFT_REPO_ROOT
B=${1:-128}; DT=${2:-bf16}
O=gpurun_out/pmct
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_WAIT_IFETCH[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*" | sort -u > $O/avail.txt
cat $O/avail.txt | tr '\n' ' '; echo
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o sq1 -- python3 scripts/bench_train.py $B $DT > $O/out1.log 2>&1 && \
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O -o sq2 -- python3 scripts/bench_train.py $B $DT > $O/out2.log 2>&1 ; \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o fetch -- python3 scripts/bench_train.py $B $DT > $O/out3.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O -o write -- python3 scripts/bench_train.py $B $DT > $O/out4.log 2>&1
python3 - <<'PY'
import csv, glob, collections
O = 'gpurun_out/pmct'
tab = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in sorted(glob.glob(O + '/*_counter_collection.csv')):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:48]
        tab[k][r['Counter_Name']] += float(r['Counter_Value'])
        if f.endswith('sq1_counter_collection.csv') and r['Counter_Name'] == 'GRBM_GUI_ACTIVE': calls[k] += 1
names = sorted({c for v in tab.values() for c in v})
print('per kernel, summed over 13 steps; FETCH/WRITE_SIZE in KB (FETCH x2 for gfx950 per the guide)')
print('kernel'.ljust(50), 'calls', ' '.join(n[:14].rjust(14) for n in names))
tot = collections.defaultdict(float)
for k, v in sorted(tab.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0)):
    print(k.ljust(50), str(calls[k]).rjust(5), ' '.join(('%.3g' % v.get(n, 0)).rjust(14) for n in names))
    for n in names: tot[n] += v.get(n, 0)
print('TOTAL'.ljust(56), ' '.join(('%.4g' % tot[n]).rjust(14) for n in names))
hbm = (2 * tot.get('FETCH_SIZE', 0) + tot.get('WRITE_SIZE', 0)) * 1024 / 13
print('HBM bytes per step: %.1f MB' % (hbm / 1e6))
PY
for i in 1 2 3 4; do tail -1 $O/out$i.log | cut -c1-220; done
