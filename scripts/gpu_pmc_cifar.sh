#!/bin/bash
# PMC passes over the CIFAR-shape score evaluation (scripts/gpu_cifar.py, B=64 with guidance = 128 forwards): where tconv_kernel's time goes
cd $GRAFT_REPO_ROOT
DT=${1:-bf16}
O=gpurun_out/pmcc
mkdir -p $O
export TMPDIR=/tmp
export CIFAR_DTYPE=$DT CIFAR_PROF=64
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o sq1 -- python3 scripts/gpu_cifar.py > $O/out1.log 2>&1 && \
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O -o sq2 -- python3 scripts/gpu_cifar.py > $O/out2.log 2>&1 ; \
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O -o sq3 -- python3 scripts/gpu_cifar.py > $O/out3.log 2>&1 ; \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o fetch -- python3 scripts/gpu_cifar.py > $O/out4.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O -o write -- python3 scripts/gpu_cifar.py > $O/out5.log 2>&1
python3 - <<'PY'
import csv, glob, collections
O = 'gpurun_out/pmcc'
tab = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in sorted(glob.glob(O + '/*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:44]
        tab[k][r['Counter_Name']] += float(r['Counter_Value'])
        if f.endswith('sq1_counter_collection.csv') and r['Counter_Name'] == 'GRBM_GUI_ACTIVE': calls[k] += 1
names = sorted({c for v in tab.values() for c in v})
print('per kernel, summed over the run (3 score evaluations of 128 forwards); FETCH/WRITE_SIZE in KB')
for k, v in sorted(tab.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0))[:8]:
    print(k, 'calls', calls[k])
    for n in names: print('   ', n.ljust(28), '%.4g' % v.get(n, 0))
PY
