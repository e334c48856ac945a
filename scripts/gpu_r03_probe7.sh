#!/bin/bash
# quick: bench line + stamps (wave 0 / 4) on the current main library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
python bench.py --steps 1 --warmup 0 --num-scales 200 --no-cpu-baseline --no-variants 2>/dev/null | tail -1 > gpurun_out/probe/bench7.json
python3 -c "
import json; d=json.loads(open('gpurun_out/probe/bench7.json').read()); print('N=200 value', d['value'], 'launch us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
for w in 0 4; do
  RDMI_UDBG=$((w << 16)) timeout -k 10 200 python scripts/gpu_stamps.py 128 > gpurun_out/probe/s7_wave$w.txt 2>&1 || exit 1
done
