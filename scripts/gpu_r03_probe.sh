#!/bin/bash
# round-3 probe: per-wave fine stamps of the co-operative kernel, instruction-cache / instruction-mix counters, build variants
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
export TMPDIR=/tmp
for w in 0 4 3 7; do
  RDMI_UDBG=$((w << 16)) timeout -k 10 200 python scripts/gpu_stamps.py 128 > gpurun_out/probe/stamps_wave$w.txt 2>&1 || exit 1
done
BENCH="python3 bench.py --steps 1 --warmup 0 --num-scales 40 --no-cpu-baseline --no-roofline --no-variants"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d gpurun_out/probe -o ic -- $BENCH > /dev/null 2> gpurun_out/probe/stderr_ic.log && \
rocprofv3 --pmc SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/probe -o mix -- $BENCH > /dev/null 2> gpurun_out/probe/stderr_mix.log
python3 - <<'PY' > gpurun_out/probe/pmc_summary.txt
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/probe/*_counter_collection.csv')):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'unet_wg' in r['Kernel_Name']:
            per[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in per.items():
        print(f, k, 'n=', len(v), 'mean=', sum(v) / len(v))
PY
cat gpurun_out/probe/pmc_summary.txt
bash scripts/gpu_variants.sh > gpurun_out/probe/variants.txt 2>&1
cat gpurun_out/probe/variants.txt
