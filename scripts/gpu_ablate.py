"""Kernel-time ablation (profiling aid): per-kernel mean times of 40 sampler updates under RDMI_DBG settings."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, json
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'optimized-diffusion-model_amd'))
import torch
import __graft_entry__ as ge
from rdmi import sampling, sde_lib
dev = torch.device('cuda:0')
model, cfg, _ = ge.make_model(dev, num_scales=41, corrector='none')
B = 128
sde = sde_lib.RVESDE(0.01, 5, N=41)
lab = torch.rand(B, 1, device=dev)
fn = sampling.get_pc_sampler(sde, (B,1,9,9), sampling.get_predictor('euler_maruyama'), sampling.get_corrector('none'), sampling.get_denoiser('none'), 0.01, 1, 1e-5, dev, seed=3)
fn(model, weight=0.0, class_labels=lab)
ctx = model._ctx[(str(dev), 9, 9)]
import time
torch.cuda.synchronize(); t0=time.perf_counter()
fn(model, weight=0.0, class_labels=lab)
torch.cuda.synchronize(); wall=time.perf_counter()-t0
ctx.set_profiling(True)
fn(model, weight=0.0, class_labels=lab)
torch.cuda.synchronize()
prof = ctx.get_profile()
print(json.dumps({'wall_ms_per_update': wall*1e3/40, 'k': {p['kernel']: round(1e3*p['ms']/max(p['launches'],1),1) for p in prof}}))
''' % (ROOT, ROOT)
for dbg in sys.argv[1:] or ['0', '1', '2', '3']:
    env = dict(os.environ, RDMI_UDBG=dbg)
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
    print('RDMI_UDBG=' + dbg, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-2000:], flush=True)
