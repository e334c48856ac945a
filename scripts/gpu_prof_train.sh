cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/proft
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/proft -o t1 -- python3 scripts/bench_train.py 128 > gpurun_out/proft/out.json 2> gpurun_out/proft/stderr.log
head -16 gpurun_out/proft/t1_kernel_stats.csv | cut -c1-150
