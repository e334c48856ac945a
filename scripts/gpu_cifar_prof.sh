#!/bin/bash
CIFAR_DTYPE=bf16 CIFAR_PROF=64 python scripts/gpu_cifar.py > gpurun_out/cifar_prof_bf16.txt 2>&1; tail -14 gpurun_out/cifar_prof_bf16.txt
CIFAR_DTYPE=f32 CIFAR_PROF=64 python scripts/gpu_cifar.py > gpurun_out/cifar_prof_f32.txt 2>&1; tail -12 gpurun_out/cifar_prof_f32.txt
python -m pytest tests/test_gpu_cifar.py -x -q -m gpu 2>&1 | tail -5
