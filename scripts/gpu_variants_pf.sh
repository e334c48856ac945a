#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in pf4 pf6 pf8 pf12 pf4; do timeout -k 10 200 python scripts/gpu_variant_bench.py variants/librdmi_$v.so 200 128 2>&1 | tail -1; done
