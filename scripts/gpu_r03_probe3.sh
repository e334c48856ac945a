#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
bash scripts/gpu_variants.sh > gpurun_out/probe/variants3.txt 2>&1
cat gpurun_out/probe/variants3.txt
V=$(ls variants/librdmi_*.so | head -1)
for w in 0 4; do
  RDMI_VARIANT_LIB=$PWD/$V RDMI_UDBG=$((w << 16)) timeout -k 10 200 python scripts/gpu_stamps.py 128 > gpurun_out/probe/pre_wave$w.txt 2>&1 || exit 1
done
