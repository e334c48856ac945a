#!/bin/bash
mkdir -p gpurun_out/iconv
RDMI_ICONV_MIN_WGS=${1:-128} RDMI_PROF_SHAPES=1 CIFAR_DTYPE=bf16 CIFAR_PROF=64 timeout -k 10 300 python scripts/gpu_cifar.py > gpurun_out/iconv/prof_shapes_min$1.txt 2>&1 && tail -${2:-26} gpurun_out/iconv/prof_shapes_min$1.txt
