set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout -k 10 500 python bench.py --steps 2 --warmup 1 2>&1 | tail -1 | tee gpurun_out/bench_none.json
