#!/bin/bash
mkdir -p gpurun_out/r2z
python -m pytest tests -x -q -m gpu > gpurun_out/r2z/tests.log 2>&1 && tail -3 gpurun_out/r2z/tests.log && python bench.py --steps 3 --warmup 1 > gpurun_out/r2z/bench.json 2> gpurun_out/r2z/bench.err && cat gpurun_out/r2z/bench.json
