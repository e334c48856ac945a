#!/bin/bash
# training-step timing: recorded launch graphs (default) against plain launches, fp32 and bf16, B = 128; then the training parity tests
set -e
mkdir -p gpurun_out
for dt in bf16 f32; do
  for g in 1 0; do
    echo "== train_dtype=$dt RDMI_TRAIN_GRAPH=$g"
    RDMI_TRAIN_GRAPH=$g timeout -k 10 200 python scripts/bench_train.py 128 $dt
  done
done
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "train or optimizer or bf16 or loss" 2>&1 | tail -5
