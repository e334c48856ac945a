"""GPU check of the co-operative program: golden forward parity under the default group stride (8: one XCD per group), stride 1
(groups spanning XCDs: the exchange must be placement-independent) and with the program disabled, then a short B=128 timing."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'optimized-diffusion-model_amd'))
import __graft_entry__ as ge
ge.build()
from rdmi import sampling, sde_lib
from rdmi.models import utils as mutils
dev = torch.device('cuda:0')
g = np.load(os.path.join(ROOT, 'tests/golden/forward_9x9.npz'))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sde = sde_lib.RVESDE(0.01, 5, N=1000)
for env in ({}, {'RDMI_COOP_STRIDE': '1'}, {'RDMI_COOP': '0'}):
    os.environ.update(env)
    model, cfg, params = ge.make_model(dev)
    with torch.no_grad():
        s = mutils.get_score_fn(sde, model)(T(g['x']), T(g['t']), class_labels=T(g['labels']))
    ctx = list(model._ctx.values())[0]
    print(env, 'golden err', float(np.abs(s.cpu().numpy() - g['score']).max()), 'gave_up', ctx.coop_gave_up(), flush=True)
    # B = 128 with guidance: 256 forwards per update
    B, N = 128, 60
    sde2 = sde_lib.RVESDE(0.01, 5, N=N)
    lab = torch.rand(B, 1, device=dev)
    fn = sampling.get_pc_sampler(sde2, (B, 1, 9, 9), sampling.get_predictor('euler_maruyama'), sampling.get_corrector('none'),
                                 sampling.get_denoiser('none'), 0.01, 1, 1e-5, dev, seed=3)
    for rep in range(3):
        torch.manual_seed(0)
        torch.cuda.synchronize(); t0 = time.time()
        x, nfe = fn(model, weight=0.0, class_labels=lab)
        torch.cuda.synchronize(); dt = time.time() - t0
    ctx = [c for c in model._ctx.values()][-1]
    print('   ', ctx.path_info())
    print('    B=128: %.1f us per update, finite %s, in cube %s, gave_up %s, checksum %.6f' % (dt / (N - 1) * 1e6, bool(torch.isfinite(x).all()),
          bool((x.min() >= 0) & (x.max() <= 1)), ctx.coop_gave_up(), float(x.double().sum())), flush=True)
    for k in env: os.environ.pop(k)
