#!/bin/bash
# run every variants/librdmi_*.so (or the names given) through gpu_variant_bench.py, one process each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/variant_ref_*.pt
names="$@"
if [ -z "$names" ]; then names=$(ls variants/librdmi_*.so | sed 's/.*librdmi_//; s/\.so//'); fi
for n in $names; do
  timeout -k 10 120 python scripts/gpu_variant_bench.py variants/librdmi_$n.so ${NSCALES:-100} ${BATCH:-128} 2>&1 | grep -v amdgpu.ids | tail -2
done
