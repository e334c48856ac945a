import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'optimized-diffusion-model_amd'))
import __graft_entry__ as ge
ge.build()
from rdmi import sde_lib
from rdmi.models import utils as mutils
dev = torch.device('cuda:0')
g = np.load(os.path.join(ROOT, 'tests/golden/forward_9x9.npz'))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sde = sde_lib.RVESDE(0.01, 5, N=1000)
model, cfg, params = ge.make_model(dev)
for idx in ([0], [0, 1, 2, 3], list(range(8)), [0, 0, 0, 0], [0, 1]):
    for rep in range(2):
        with torch.no_grad():
            s = mutils.get_score_fn(sde, model)(T(g['x'][idx]), T(g['t'][idx]), class_labels=T(g['labels'][idx]))
        e = np.abs(s.cpu().numpy() - g['score'][idx]).reshape(len(idx), -1)
        print(idx, 'rep', rep, 'per-sample max err', np.round(e.max(1), 5), 'frac bad pixels', np.round((e > 1e-3).mean(1), 2), flush=True)
ctx = list(model._ctx.values())[0]
print(ctx.path_info(), ctx.coop_gave_up())
