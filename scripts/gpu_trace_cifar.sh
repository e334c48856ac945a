#!/bin/bash
# rocprofv3 kernel trace of the CIFAR-shape score evaluation, aggregated per (kernel, grid): which layers cost what
cd $GRAFT_REPO_ROOT
O=gpurun_out/trc
mkdir -p $O
export TMPDIR=/tmp
export CIFAR_DTYPE=${1:-bf16} CIFAR_PROF=64
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 scripts/gpu_cifar.py > $O/out.log 2>&1
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/trc/t_kernel_trace.csv')))
d = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
    d[(n, r['Grid_Size_X'], r['Grid_Size_Y'], r['LDS_Block_Size'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in d.values())
print('total kernel ms (3 evaluations)', tot / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print(k[0].ljust(42), ('grid %s x %s lds %s' % (k[1], k[2], k[3])).ljust(34), 'n=%4d' % len(v), 'avg %8.1f us' % (sum(v) / len(v)), 'sum %8.2f ms' % (sum(v) / 1e3))
PY
