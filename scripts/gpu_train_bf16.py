"""bf16 training step vs the fp32 reference fixture: losses, gradient norms; and step timing at a few batch sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'optimized-diffusion-model_amd'))
import numpy as np, torch
import __graft_entry__ as ge
from tests import test_emu_parity as TE
dev = torch.device('cuda:0')
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'train_step.npz'))
_mk = ge.make_model
for dt in ('f32', 'bf16'):
    def mk(device, **kw):
        m, cfg, p = _mk(device, **kw); m.train_dtype = dt; m._ctx.clear(); return m, cfg, p
    ge.make_model = mk
    out = TE._train_two_steps(ge, dev, g)
    names = list(g['param_names'])
    gn = np.array([np.sqrt((out['grads'][n].astype(np.float64) ** 2).sum()) for n in names]); ref = g['step0.grad_norms'].astype(np.float64)
    big = ref > 1e-7 * ref.max()
    print(dt, 'loss0', out['loss0'], float(g['step0.loss']), 'loss1', out['loss1'], float(g['step1.loss']), 'grad-norm max rel err', float(np.abs(gn[big] / ref[big] - 1).max()),
          'median', float(np.median(np.abs(gn[big] / ref[big] - 1))))
    r = g['step0.grad.out_conv.weight']; print('   out_conv.weight grad max err / max', float(np.abs(out['grads']['out_conv.weight'] - r).max() / np.abs(r).max()))
ge.make_model = _mk
from rdmi import losses, sde_lib
from rdmi.models.ema import ExponentialMovingAverage
for dt in ('f32', 'bf16'):
    for B in (128, 1024, 4096):
        model, cfg, _ = ge.make_model(dev); model.train_dtype = dt; model.train()
        sde = sde_lib.RVESDE(0.01, 5, N=1000)
        opt = losses.get_optimizer(cfg, model.parameters()); ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
        state = dict(optimizer=opt, model=model, ema=ema, step=0, scaler=None)
        fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=False, likelihood_weighting=False)
        batch = torch.rand(B, 1, 9, 9, device=dev); lab = torch.rand(B, 1, device=dev)
        for _ in range(3): l = fn(state, batch, class_labels=lab)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): l = fn(state, batch, class_labels=lab)
        torch.cuda.synchronize(); d = (time.perf_counter() - t0) / 5
        print(f'{dt} B={B}: {d*1e3:.2f} ms/step  {B/d:.0f} samples/s  loss {float(l.detach()):.3f}', flush=True)
        del model, opt, ema, state
