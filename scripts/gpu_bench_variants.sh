set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --corrector langevin --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1100 | tee gpurun_out/bench_langevin.json
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --height 8 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-700 | tee gpurun_out/bench_8x9.json
timeout -k 10 400 python bench.py --steps 1 --warmup 1 --batch 1024 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-400 | tee gpurun_out/bench_b1024.json
RDMI_PATH=layers timeout -k 10 400 python bench.py --steps 1 --warmup 1 --batch 1024 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-400 | tee gpurun_out/bench_b1024_layers.json
