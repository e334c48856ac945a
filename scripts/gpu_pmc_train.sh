set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmct
export TMPDIR=/tmp
B=${1:-128}
# separate PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass); kernel-trace only
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmct -o fetch -- python3 scripts/bench_train.py $B > gpurun_out/pmct/out_fetch.json 2> gpurun_out/pmct/stderr_fetch.log && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmct -o write -- python3 scripts/bench_train.py $B > gpurun_out/pmct/out_write.json 2> gpurun_out/pmct/stderr_write.log && \
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmct -o stats -- python3 scripts/bench_train.py $B > gpurun_out/pmct/out_stats.json 2> gpurun_out/pmct/stderr_stats.log
python3 - <<'PY'
import csv, glob, collections
tot = {}
for tag in ('fetch', 'write'):
    for f in glob.glob(f'gpurun_out/pmct/{tag}_counter_collection.csv'):
        per = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0][:60]
            per[k][0] += 1; per[k][1] += float(r['Counter_Value'])
        tot[tag] = per
        print(tag, 'total KB', sum(v[1] for v in per.values()))
        for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:12]:
            print(f'  {tag} {k:62s} n={v[0]:5d} KB={v[1]:.0f}')
PY
cat gpurun_out/pmct/out_stats.json
