"""score_hk error statistics against the reference fixture (to set the test tolerance from measurement)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'optimized-diffusion-model_amd'))
import numpy as np, torch
from rdmi import cube
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'cube_sde.npz'))
dev = torch.device('cuda:0')
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
out = cube.score_hk(T(g['hk_x']), T(g['hk_x0']), T(g['hk_sigma'])).cpu().numpy().astype(np.float64)
ref = g['hk_score'].astype(np.float64)
err = np.abs(out - ref)
print('shape', ref.shape, 'max|ref|', np.abs(ref).max(), 'median|ref|', np.median(np.abs(ref)))
print('max abs err', err.max(), 'at |ref| =', np.abs(ref).ravel()[err.argmax()], 'sigma', np.broadcast_to(g['hk_sigma'].reshape(-1, *([1] * (ref.ndim - 1))), ref.shape).ravel()[err.argmax()] if g['hk_sigma'].ndim else g['hk_sigma'])
rel = err / np.maximum(np.abs(ref), 1e-30)
big = np.abs(ref) > 1e-3 * np.abs(ref).max()
print('max rel err over |ref| > 1e-3 max:', rel[big].max(), ' max err / max|ref|:', err.max() / np.abs(ref).max())
sig = np.broadcast_to(g['hk_sigma'].reshape(-1, *([1] * (ref.ndim - 1))), ref.shape)
print('max err * sigma^2:', (err * sig ** 2).max(), ' max |ref| * sigma^2:', (np.abs(ref) * sig ** 2).max())
