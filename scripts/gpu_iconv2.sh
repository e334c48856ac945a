#!/bin/bash
# iconv_kernel: per-shape profile rows of one CFG score evaluation at B = 64 (128 forwards), default build and the 4-waves variant
mkdir -p gpurun_out/iconv
RDMI_PROF_SHAPES=1 CIFAR_DTYPE=bf16 CIFAR_PROF=64 timeout -k 10 300 python scripts/gpu_cifar.py > gpurun_out/iconv/prof_shapes.txt 2>&1 && tail -22 gpurun_out/iconv/prof_shapes.txt
RDMI_LIB=$PWD/variants/librdmi_w4.so RDMI_PROF_SHAPES=1 CIFAR_DTYPE=bf16 CIFAR_PROF=64 timeout -k 10 300 python scripts/gpu_cifar.py > gpurun_out/iconv/prof_shapes_w4.txt 2>&1 && tail -22 gpurun_out/iconv/prof_shapes_w4.txt
