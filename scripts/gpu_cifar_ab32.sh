#!/bin/bash
# A/B of the CIFAR-shape sampler update in both dtypes: default library against RDMI_LIB (env), B = 64 with guidance
for dt in f32 bf16; do for lib in "" "$1"; do echo "== $dt ${lib:-main}"; RDMI_LIB=$lib CIFAR_DTYPE=$dt timeout -k 10 300 python scripts/gpu_cifar.py 64 ${2:-9} 2>&1 | grep "ms/update\|rel"; done; done
