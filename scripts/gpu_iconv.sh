#!/bin/bash
# CIFAR-shape bf16 plan: parity tests, then the per-kernel table of one CFG score evaluation at B = 64 (128 forwards)
mkdir -p gpurun_out/iconv
timeout -k 10 600 python -m pytest tests/test_gpu_cifar.py -x -q -m gpu -k "${1:-implicit_gemm or bf16_forward or batch8}" > gpurun_out/iconv/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/iconv/tests.log
RDMI_PROF_SHAPES=1 CIFAR_DTYPE=bf16 CIFAR_PROF=64 timeout -k 10 300 python scripts/gpu_cifar.py 64 6 > gpurun_out/iconv/prof_on.txt 2>&1 && tail -${2:-27} gpurun_out/iconv/prof_on.txt
