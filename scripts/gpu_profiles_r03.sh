#!/bin/bash
# Round-3 profile collection (one gpurun call): everything lands in gpurun_out/r03/, the summaries are copied to profiles/ afterwards.
#  1. rocprofv3 --kernel-trace --stats of a short bench.py run (same command shape as rounds 1-2)
#  2. PMC passes on the dominant kernel: matrix-pipe busy / wave states (two passes), FETCH_SIZE, WRITE_SIZE (one pass each)
#  3. per-op cycle stamps of the co-operative kernel
#  4. training step (B = 128 bf16): kernel stats + timeline, FETCH_SIZE / WRITE_SIZE passes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; rm -rf $O; mkdir -p $O gpurun_out/prof gpurun_out/pmct
export TMPDIR=/tmp
set -x
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- python3 bench.py --steps 1 --warmup 1 --num-scales 200 --no-cpu-baseline --no-variants > $O/bench_under_rocprof.json 2> $O/stderr_stats.log
echo step1 done
BENCH40="python3 bench.py --steps 1 --warmup 0 --num-scales 40 --no-cpu-baseline --no-roofline --no-variants"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o sq1 -- $BENCH40 > /dev/null 2> $O/stderr_sq1.log && \
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O -o sq2 -- $BENCH40 > /dev/null 2> $O/stderr_sq2.log && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof -o pmc_fetch -- $BENCH40 > /dev/null 2> $O/stderr_fetch.log && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof -o pmc_write -- $BENCH40 > /dev/null 2> $O/stderr_write.log
echo step2 done
python3 - <<'PY' > gpurun_out/r03/pmc_unet_summary.txt
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r03/sq*_counter_collection.csv')) + sorted(glob.glob('gpurun_out/prof/pmc_*_counter_collection.csv')):
    per = collections.defaultdict(list); names = set()
    for r in csv.DictReader(open(f)):
        if 'unet_wg' in r['Kernel_Name']:
            per[r['Counter_Name']].append(float(r['Counter_Value'])); names.add(r['Kernel_Name'][:60])
    for k, v in per.items():
        print(f.split('/')[-1], k, 'n=', len(v), 'mean=', sum(v) / len(v), sorted(names))
PY
cat $O/pmc_unet_summary.txt
python3 scripts/gpu_stamps.py 128 > $O/coop_kernel_per_op_cycles.txt 2>&1
echo step3 done
./scripts/gpu_prof_train_timeline.sh 128 bf16 > $O/train_b128_bf16_timeline.txt 2>&1
cp gpurun_out/proft3/t_kernel_stats.csv $O/train_b128_bf16_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmct -o fetch -- python3 scripts/bench_train.py 128 bf16 > $O/train_fetch.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmct -o write -- python3 scripts/bench_train.py 128 bf16 > $O/train_write.log 2>&1
python3 - <<'PY' > gpurun_out/r03/train_traffic_per_kernel.txt
import csv, collections
tab = collections.defaultdict(lambda: collections.defaultdict(float))
for tag in ('fetch', 'write'):
    for r in csv.DictReader(open(f'gpurun_out/pmct/{tag}_counter_collection.csv')):
        tab[r['Kernel_Name'].split('(')[0][:60]][r['Counter_Name']] += float(r['Counter_Value'])
print('per kernel, KB per step (13 steps: 3 warm + 10 timed); FETCH_SIZE counts 64-B requests as 32 B on gfx950: x2 for bytes')
tf = tw = 0
for k, v in sorted(tab.items(), key=lambda kv: -(kv[1].get('WRITE_SIZE', 0) + 2 * kv[1].get('FETCH_SIZE', 0))):
    print(k.ljust(62), 'FETCH %10.1f  WRITE %10.1f' % (v.get('FETCH_SIZE', 0) / 13, v.get('WRITE_SIZE', 0) / 13)); tf += v.get('FETCH_SIZE', 0); tw += v.get('WRITE_SIZE', 0)
print('TOTAL per step: FETCH_SIZE %.1f KB, WRITE_SIZE %.1f KB -> HBM bytes %.1f MB' % (tf / 13, tw / 13, (2 * tf + tw) * 1024 / 13 / 1e6))
PY
tail -3 $O/train_traffic_per_kernel.txt
echo step4 done
