#!/bin/bash
# build_variant.sh NAME [-D...]: experimental build of librdmi into build/variants/librdmi_NAME.so (never loaded by the product)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -Wno-unused-value -Wno-pass-failed "$@" \
    $ROOT/optimized-diffusion-model_amd/csrc/rdmi.hip -o $ROOT/variants/librdmi_$NAME.so
echo built $NAME
