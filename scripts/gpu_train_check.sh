#!/bin/bash
mkdir -p gpurun_out/train_check
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "train or optimizer or bf16 or loss" > gpurun_out/train_check/tests.log 2>&1 && tail -3 gpurun_out/train_check/tests.log && python scripts/bench_train.py 128 bf16 && python scripts/bench_train.py 128 bf16 && python scripts/bench_train.py 128 f32 && python scripts/bench_train.py 1024 bf16 && python scripts/bench_train.py 4096 bf16 && python scripts/gpu_train_bf16.py 2>&1 | head -6
