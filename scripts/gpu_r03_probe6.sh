#!/bin/bash
# parity (whole GPU parity file) + bench line on the current main library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4 || exit 1
python bench.py --steps 1 --warmup 0 --num-scales 200 --no-cpu-baseline --no-variants 2>/dev/null | tail -1 > gpurun_out/probe/bench6.json
python3 -c "
import json; d=json.loads(open('gpurun_out/probe/bench6.json').read()); print('N=200 value', d['value'], 'launch us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])"
