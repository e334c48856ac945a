#!/bin/bash
# end-to-end A/B of the CIFAR-shape bf16 sampler update (B = 64 with guidance, 12 updates): environment switches of the tiled plan
mkdir -p gpurun_out/iconv
for cfg in "X=1" "${@}"; do
  echo "== $cfg"
  env $cfg CIFAR_DTYPE=bf16 timeout -k 10 300 python scripts/gpu_cifar.py 64 13 2>&1 | grep "ms/update"
done
