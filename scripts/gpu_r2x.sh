#!/bin/bash
mkdir -p gpurun_out/r2x
python -m pytest tests -x -q -m gpu > gpurun_out/r2x/tests.log 2>&1 && tail -3 gpurun_out/r2x/tests.log && python scripts/gpu_train_bf16.py > gpurun_out/r2x/train_bf16.log 2>&1 && cat gpurun_out/r2x/train_bf16.log
