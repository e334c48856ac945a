#!/bin/bash
# parity file + bench line + stamps on the current main library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4 || exit 1
bash scripts/gpu_r03_probe7.sh
