"""Host enqueue time vs device time of the B=128 training step: is the step bound by the Python / launch path or by the GPU?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'optimized-diffusion-model_amd'))
import torch
import __graft_entry__ as ge
from rdmi import losses, sde_lib
from rdmi.models.ema import ExponentialMovingAverage
dev = torch.device('cuda:0')
for DT in ('bf16', 'f32'):
    model, cfg, _ = ge.make_model(dev); model.train_dtype = DT; model.train()
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    opt = losses.get_optimizer(cfg, model.parameters()); ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    state = dict(optimizer=opt, model=model, ema=ema, step=0, scaler=None)
    fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=False, likelihood_weighting=False)
    batch = torch.rand(128, 1, 9, 9, device=dev); lab = torch.rand(128, 1, device=dev)
    for _ in range(5): fn(state, batch, class_labels=lab)
    torch.cuda.synchronize()
    K = 30
    t0 = time.perf_counter()
    for _ in range(K): fn(state, batch, class_labels=lab)
    t_host = (time.perf_counter() - t0) / K
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / K
    print(f'{DT}: host enqueue {t_host*1e3:.2f} ms/step, with final sync {t_all*1e3:.2f} ms/step', flush=True)
    import cProfile, pstats, io
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10): fn(state, batch, class_labels=lab)
    pr.disable(); torch.cuda.synchronize()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(14); print(s.getvalue()[:3000])
