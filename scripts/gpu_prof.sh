set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1 -- python3 bench.py --steps 1 --warmup 1 --num-scales 200 --no-cpu-baseline --no-roofline > gpurun_out/prof/bench_under_prof.json 2> gpurun_out/prof/stderr.log
ls -la gpurun_out/prof | head -20
find gpurun_out/prof -name "*stats*" | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
head -30 $f
