#!/bin/bash
bash scripts/gpu_pmc_train.sh 128 bf16 > gpurun_out/pmct_summary.txt 2>&1
bash scripts/gpu_prof_train.sh 128 bf16 > gpurun_out/proft2_summary.txt 2>&1
cp gpurun_out/proft2/t_kernel_stats.csv gpurun_out/proft2_b128_bf16_kernel_stats.csv
bash scripts/gpu_prof_train.sh 4096 bf16 > gpurun_out/proft2_summary_4096.txt 2>&1
cp gpurun_out/proft2/t_kernel_stats.csv gpurun_out/proft2_b4096_bf16_kernel_stats.csv
tail -5 gpurun_out/pmct_summary.txt; head -12 gpurun_out/proft2_summary.txt; head -12 gpurun_out/proft2_summary_4096.txt
CIFAR_DTYPE=bf16 CIFAR_PROF=64 python scripts/gpu_cifar.py > gpurun_out/cifar_prof_bf16.txt 2>&1; tail -12 gpurun_out/cifar_prof_bf16.txt
CIFAR_DTYPE=f32 CIFAR_PROF=64 python scripts/gpu_cifar.py > gpurun_out/cifar_prof_f32.txt 2>&1; tail -12 gpurun_out/cifar_prof_f32.txt
