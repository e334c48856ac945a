#!/bin/bash
# SQ / MFMA counters of the S = 4 samples-per-workgroup program (B = 1024 with guidance: 2048 forwards = 512 workgroups per launch)
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcm1024
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o sq1 -- python3 bench.py --batch 1024 --steps 1 --warmup 0 --num-scales 20 --no-cpu-baseline --no-roofline --no-variants > /dev/null 2> $O/stderr1.log && \
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O -o sq2 -- python3 bench.py --batch 1024 --steps 1 --warmup 0 --num-scales 20 --no-cpu-baseline --no-roofline --no-variants > /dev/null 2> $O/stderr2.log
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmcm1024/sq*_counter_collection.csv')):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'unet_wg' in r['Kernel_Name']:
            per[(r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in per.items():
        print(k[0], k[1], 'n=', len(v), 'mean=', sum(v) / len(v))
rows = list(csv.DictReader(open('gpurun_out/pmcm1024/sq1_kernel_trace.csv')))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'unet_wg' in r['Kernel_Name']]
print('unet_wg launches', len(d), 'mean us', sum(d) / max(len(d), 1))
PY
