#!/bin/bash
# rocprofv3 kernel trace of the CIFAR-shape bf16 score evaluation (B = 64 with guidance), aggregated per (kernel, grid), for two plans:
# tag "on": default plan; tag "off": RDMI_ICONV=0 (identical now that the implicit-GEMM conv is off by default; TAGS selects)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export CIFAR_DTYPE=bf16 CIFAR_PROF=64
for tag in ${TAGS:-on off}; do
  O=gpurun_out/trc_$tag; rm -rf $O; mkdir -p $O
  if [ $tag = off ]; then export RDMI_ICONV=0; fi
  rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 scripts/gpu_cifar.py > $O/out.log 2>&1 || { echo "rocprof failed"; tail -5 $O/out.log; exit 1; }
  python3 - $O <<'PY' > $O/summary.txt
import csv, collections, sys
O = sys.argv[1]
rows = list(csv.DictReader(open(O + '/t_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# keep the LAST score evaluation: kernels after the last pack_kernel
last = max(i for i, r in enumerate(rows) if 'pack_kernel' in r['Kernel_Name'])
rows = rows[last + 1:]
d = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
    d[(n, r['Grid_Size_X'], r['Grid_Size_Y'], r['LDS_Block_Size'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in d.values())
span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6
print('last evaluation: %d kernels, kernel time %.2f ms, span first start -> last end %.2f ms' % (len(rows), tot / 1e3, span))
byk = collections.defaultdict(float)
for k, v in d.items(): byk[k[0]] += sum(v)
for k, v in sorted(byk.items(), key=lambda kv: -kv[1]): print('  %-42s %8.2f ms' % (k, v / 1e3))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:45]:
    print(k[0].ljust(42), ('grid %s x %s lds %s' % (k[1], k[2], k[3])).ljust(34), 'n=%4d' % len(v), 'avg %8.1f us' % (sum(v) / len(v)), 'sum %8.2f ms' % (sum(v) / 1e3))
PY
  head -16 $O/summary.txt
done
