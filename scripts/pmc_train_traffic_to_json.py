"""Turn the FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_pmc_train.sh (gpurun_out/pmct/{fetch,write}_counter_collection.csv: every
kernel of 13 training steps) into profiles/train_traffic.json, from which bench.py's train variant reports HBM GB/s.
usage: pmc_train_traffic_to_json.py [B] [dtype]"""
import csv, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha16

def total(tag):
    s = 0.0
    for r in csv.DictReader(open(os.path.join(ROOT, 'gpurun_out', 'pmct', f'{tag}_counter_collection.csv'))):
        s += float(r['Counter_Value'])
    return s

STEPS = 13                                   # scripts/bench_train.py: 3 warm + 10 timed
fetch, write = total('fetch'), total('write')
commit = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
out = {'batch': int(sys.argv[1]) if len(sys.argv) > 1 else 128, 'dtype': sys.argv[2] if len(sys.argv) > 2 else 'bf16',
       'fetch_size_kb_per_step': round(fetch / STEPS, 1), 'write_size_kb_per_step': round(write / STEPS, 1),
       'bytes_per_step': int((2 * fetch + write) * 1024 / STEPS), 'csrc_sha16': csrc_sha16(), 'commit': commit,
       'source': 'scripts/gpu_pmc_train.sh (separate --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/bench_train.py; FETCH x2 per the gfx950 correction)'}
json.dump(out, open(os.path.join(ROOT, 'profiles', 'train_traffic.json'), 'w'), indent=1)
print(out)
