"""Large-batch check + timing of the multi-sample programs: oracle parity on a few samples, then traj/s at the given batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'optimized-diffusion-model_amd'))
import numpy as np, torch
import __graft_entry__ as ge
from rdmi import sampling, sde_lib
from rdmi.models import utils as mutils
from oracle import rd_oracle as O
dev = torch.device('cuda:0')
B = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
model, cfg, params = ge.make_model(dev, num_scales=N)
sde = sde_lib.RVESDE(0.01, 5, N=N)
g = torch.Generator().manual_seed(9)
x = torch.rand(2 * B, 1, 9, 9, generator=g); t = torch.rand(2 * B, generator=g) * 0.99 + 0.01; lab = torch.rand(2 * B, 1, generator=g)
with torch.no_grad():
    s = mutils.get_score_fn(sde, model)(x.to(dev), t.to(dev), class_labels=lab.to(dev)).cpu().numpy()
print(model._ctx[(str(dev), 9, 9)].path_info())
idx = [0, 1, 2, 3, 4, 5, 2 * B - 1, 2 * B - 2, B, B + 1]
ref = O.score_fn(params, O.RVESDE(0.01, 5, N=N), x.numpy()[idx], t.numpy()[idx], lab.numpy()[idx])
print('B', B, 'max err vs oracle', float(np.abs(s[idx] - ref).max()), 'finite', bool(np.isfinite(s).all()))
labs = lab[:B].to(dev)
fn = sampling.get_pc_sampler(sde, (B, 1, 9, 9), sampling.get_predictor('euler_maruyama'), sampling.get_corrector('none'),
                             sampling.get_denoiser('none'), 0.01, 1, 1e-5, dev, seed=99)
fn(model, weight=0.0, class_labels=labs); torch.cuda.synchronize()
t0 = time.time(); xs, _ = fn(model, weight=0.0, class_labels=labs); torch.cuda.synchronize(); dt = time.time() - t0
print(f'B={B}: {dt / (N - 1) * 1e6:.1f} us/update -> {B / (dt / (N - 1) * 999):.1f} traj/s at 1000 scales', flush=True)
