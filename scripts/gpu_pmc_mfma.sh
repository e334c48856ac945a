cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcm
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_BUSY_CYCLES\|SQ_BUSY_CU_CYCLES\|SQ_WAVE_CYCLES\|SQ_WAIT_ANY\|SQ_WAIT_INST_ANY\|SQ_ACTIVE_INST_ANY\|SQ_ACTIVE_INST_VALU\|SQ_ACTIVE_INST_LDS\|SQ_WAIT_INST_LDS\|GRBM_GUI_ACTIVE\|SQ_INSTS_VALU\b" | sort -u > gpurun_out/pmcm/avail.txt
cat gpurun_out/pmcm/avail.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmcm -o sq1 -- python3 bench.py --steps 1 --warmup 0 --num-scales 40 --no-cpu-baseline --no-roofline --no-variants > /dev/null 2> gpurun_out/pmcm/stderr1.log && \
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmcm -o sq2 -- python3 bench.py --steps 1 --warmup 0 --num-scales 40 --no-cpu-baseline --no-roofline --no-variants > /dev/null 2> gpurun_out/pmcm/stderr2.log
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmcm/sq*_counter_collection.csv')):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'unet_wg' in r['Kernel_Name']:
            per[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in per.items():
        print(f, k, 'n=', len(v), 'mean=', sum(v) / len(v))
PY
tail -3 gpurun_out/pmcm/stderr2.log
