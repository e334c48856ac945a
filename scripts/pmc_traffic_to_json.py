"""Turn the rocprofv3 --pmc passes of scripts/gpu_prof.sh (gpurun_out/prof/pmc_{fetch,write}_counter_collection.csv) into
profiles/unet_traffic.json, the file bench.py's roofline.traffic is read from.  Records the kernel-source hash and commit the
passes were taken on: bench.py emits traffic=null when the sources have changed since."""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha16

def mean_of(tag, kernel='unet_wg'):
    vals = []
    for f in glob.glob(os.path.join(ROOT, 'gpurun_out', 'prof', f'pmc_{tag}_counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            if kernel in r.get('Kernel_Name', ''):
                vals.append(float(r['Counter_Value']))
    if not vals:
        raise SystemExit(f'no {tag} rows for {kernel}')
    return sum(vals) / len(vals), len(vals)

fetch, nf = mean_of('fetch')
write, nw = mean_of('write')
commit = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
out = {'kernel': 'unet_wg_kernel', 'batch': int(sys.argv[1]) if len(sys.argv) > 1 else 128, 'pixels': 81,
       'fetch_size_kb_mean': round(fetch, 1), 'write_size_kb_mean': round(write, 1), 'launches_fetch_pass': nf, 'launches_write_pass': nw,
       'bytes_per_launch': int((2 * fetch + write) * 1024), 'csrc_sha16': csrc_sha16(), 'commit': commit,
       'source': 'scripts/gpu_prof.sh (separate --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --num-scales 40)'}
json.dump(out, open(os.path.join(ROOT, 'profiles', 'unet_traffic.json'), 'w'), indent=1)
print(out)
