#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
V=$(ls variants/librdmi_*.so | head -1)
for d in 0 1024; do
  RDMI_VARIANT_LIB=$PWD/$V RDMI_UDBG=$d timeout -k 10 200 python scripts/gpu_stamps.py 128 > gpurun_out/probe/abl_$d.txt 2>&1 || exit 1
done
