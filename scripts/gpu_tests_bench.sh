#!/bin/bash
mkdir -p gpurun_out/tests_bench
python -m pytest tests -x -q -m gpu > gpurun_out/tests_bench/tests.log 2>&1 && tail -3 gpurun_out/tests_bench/tests.log && python bench.py --steps 3 --warmup 1 > gpurun_out/tests_bench/bench.json 2> gpurun_out/tests_bench/bench.err && cat gpurun_out/tests_bench/bench.json
