#!/bin/bash
CIFAR_DTYPE=bf16 CIFAR_PROF=64 python scripts/gpu_cifar.py > gpurun_out/cifar_prof_bf16.txt 2>&1; tail -16 gpurun_out/cifar_prof_bf16.txt
RDMI_NO_PREACT=1 CIFAR_DTYPE=bf16 CIFAR_PROF=64 python scripts/gpu_cifar.py > gpurun_out/cifar_prof_bf16_nopre.txt 2>&1; tail -13 gpurun_out/cifar_prof_bf16_nopre.txt
python -m pytest tests/test_gpu_cifar.py -x -q -m gpu 2>&1 | tail -5
