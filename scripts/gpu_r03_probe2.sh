#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
for w in 0 4; do
  RDMI_VARIANT_LIB=$PWD/variants/librdmi_probe.so RDMI_UDBG=$((w << 16)) timeout -k 10 200 python scripts/gpu_stamps.py 128 > gpurun_out/probe/entry_wave$w.txt 2>&1 || exit 1
done
