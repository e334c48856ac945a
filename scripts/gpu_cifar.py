"""CIFAR-shape NCSN++ (BASELINE config #5) through the tiled plan: golden forward check, then a short sampler timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'optimized-diffusion-model_amd'))
import numpy as np, torch
import __graft_entry__ as ge
if os.environ.get('RDMI_LIB'):
    from rdmi import _native
    _native.use_library(os.environ['RDMI_LIB'])
from rdmi import sampling, sde_lib
from rdmi.models import utils as mutils
dev = torch.device('cuda:0')
t0 = time.time()
DT = os.environ.get('CIFAR_DTYPE', 'f32')
model, cfg, params = ge.make_cifar_model(dev, compute_dtype=DT)
print('model built', time.time() - t0, flush=True)
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'forward_cifar.npz'))
sde = sde_lib.RVESDE(0.01, 50, N=1000)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
with torch.no_grad():
    s = mutils.get_score_fn(sde, model)(T(g['x']), T(g['t']), class_labels=T(g['labels']))
torch.cuda.synchronize()
print(model._ctx[(str(dev), 32, 32)].path_info())
s = s.cpu().numpy()
ref = g['score']
for n in range(2):
    print('sample', n, 'max |ref|', float(np.abs(ref[n]).max()), 'max err', float(np.abs(s[n] - ref[n]).max()), 'rel', float(np.abs(s[n] - ref[n]).max() / np.abs(ref[n]).max()))
if len(sys.argv) > 1:
    B, N = int(sys.argv[1]), int(sys.argv[2])
    m2, cfg2, _ = ge.make_cifar_model(dev, num_scales=N, compute_dtype=DT)
    sde2 = sde_lib.RVESDE(0.01, 50, N=N)
    lab = torch.zeros(B, 1, device=dev)
    fn = sampling.get_sampling_fn(cfg2, sde2, (B, 3, 32, 32), 1e-5, dev)
    fn(m2, weight=0.0, class_labels=lab); torch.cuda.synchronize()
    t0 = time.time(); x, nfe = fn(m2, weight=0.0, class_labels=lab); torch.cuda.synchronize(); dt = time.time() - t0
    print(f'B={B} N={N}: {dt/(N-1)*1e3:.2f} ms/update -> {B/(dt/(N-1)*999):.3f} traj/s at 1000 scales; finite {bool(torch.isfinite(x).all())}')
if os.environ.get('CIFAR_PROF'):
    B = int(os.environ['CIFAR_PROF'])
    x = torch.rand(B, 3, 32, 32, device=dev); t = torch.full((B,), 0.5, device=dev); lab = torch.zeros(B, 1, device=dev)
    fn = mutils.get_cf_score_fn(sde, model, lab, 0.0)
    with torch.no_grad():
        fn(x, t); torch.cuda.synchronize()
        ctx = model._ctx[(str(dev), 32, 32)]
        ctx.set_profiling(True)
        fn(x, t); torch.cuda.synchronize()
    tot = 0
    for p_ in sorted(ctx.get_profile(), key=lambda p: -p['ms']):
        tot += p_['ms']
        print(f"{p_['kernel']:44s} {p_['launches']:5d} launches {p_['ms']:8.3f} ms  {(p_['flops'] / (p_['ms'] * 1e-3) / 1e12) if p_['flops'] else 0:7.1f} TFLOP/s")
    print('total', tot)
