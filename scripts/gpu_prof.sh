set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1 -- python3 bench.py --steps 1 --warmup 1 --num-scales 200 --no-cpu-baseline --no-variants > gpurun_out/prof/bench_under_prof.json 2> gpurun_out/prof/stderr.log
head -12 gpurun_out/prof/r1_kernel_stats.csv | cut -c1-200
# HBM traffic of the dominant kernel: separate PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof -o pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --num-scales 40 --no-cpu-baseline --no-roofline --no-variants > /dev/null 2> gpurun_out/prof/stderr_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof -o pmc_write -- python3 bench.py --steps 1 --warmup 0 --num-scales 40 --no-cpu-baseline --no-roofline --no-variants > /dev/null 2> gpurun_out/prof/stderr_write.log
ls gpurun_out/prof
python3 - <<'PY'
import csv, glob
for tag in ('fetch','write'):
    for f in glob.glob(f'gpurun_out/prof/pmc_{tag}_counter_collection.csv'):
        rows=[r for r in csv.DictReader(open(f)) if 'unet_wg' in r.get('Kernel_Name','')]
        if rows:
            vals=[float(r['Counter_Value']) for r in rows]
            print(tag, rows[0]['Counter_Name'], 'n=',len(vals),'mean=',sum(vals)/len(vals))
PY
