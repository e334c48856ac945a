cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/proft
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "train" > gpurun_out/proft/tests.log 2>&1 && \
python3 scripts/bench_train.py 128 > gpurun_out/proft/b128.json 2> gpurun_out/proft/b128.err && \
python3 scripts/bench_train.py 1024 > gpurun_out/proft/b1024.json 2> gpurun_out/proft/b1024.err && \
python3 scripts/bench_train.py 4096 > gpurun_out/proft/b4096.json 2> gpurun_out/proft/b4096.err && \
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/proft -o t4k -- python3 scripts/bench_train.py 4096 > gpurun_out/proft/out4k.json 2> gpurun_out/proft/stderr4k.log
tail -3 gpurun_out/proft/tests.log
cat gpurun_out/proft/b128.json gpurun_out/proft/b1024.json gpurun_out/proft/b4096.json
head -14 gpurun_out/proft/t4k_kernel_stats.csv | cut -c1-130
