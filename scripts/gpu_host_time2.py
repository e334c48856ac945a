"""Host-side cost of the B=128 training step, by piece (perf_counter around the pieces; a cProfile by own time)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'optimized-diffusion-model_amd'))
import torch
import __graft_entry__ as ge
from rdmi import losses, sde_lib
from rdmi.models.ema import ExponentialMovingAverage
dev = torch.device('cuda:0')
DT = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
model, cfg, _ = ge.make_model(dev); model.train_dtype = DT; model.train()
sde = sde_lib.RVESDE(0.01, 5, N=1000)
opt = losses.get_optimizer(cfg, model.parameters()); ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
state = dict(optimizer=opt, model=model, ema=ema, step=0, scaler=None)
ofn = losses.optimization_manager(cfg)
loss_fn = losses.get_sde_loss_fn(sde, True, reduce_mean=False, likelihood_weighting=False)
batch = torch.rand(128, 1, 9, 9, device=dev); lab = torch.rand(128, 1, device=dev)
T = {}
def tick(k, t0):
    T[k] = T.get(k, 0.0) + time.perf_counter() - t0
def step(sync_each=False):
    t0 = time.perf_counter(); opt.zero_grad(); tick('zero_grad', t0)
    t0 = time.perf_counter(); loss = loss_fn(model, batch, class_labels=lab); tick('forward+loss', t0)
    if sync_each: torch.cuda.synchronize()
    t0 = time.perf_counter(); loss.backward(); tick('backward', t0)
    if sync_each: torch.cuda.synchronize()
    t0 = time.perf_counter(); done = ofn(opt, model.parameters(), step=state['step'], scaler=None, ema=ema); tick('optimize_fn', t0)
    if sync_each: torch.cuda.synchronize()
    state['step'] += 1
    return loss
for _ in range(5): step()
torch.cuda.synchronize()
for mode in (False, True):
    T.clear(); K = 30
    t0 = time.perf_counter()
    for _ in range(K): step(mode)
    th = (time.perf_counter() - t0) / K
    torch.cuda.synchronize(); ta = (time.perf_counter() - t0) / K
    print(f'{DT} sync_each={mode}: host {th*1e3:.2f} ms/step, total {ta*1e3:.2f} ms/step; pieces (ms): ' + ', '.join(f'{k} {v/K*1e3:.2f}' for k, v in T.items()), flush=True)
print('graph stats', model._ctx[('train', str(dev), 9, 9)].train_graph_stats())
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step(True)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:5000])
# the C entry points alone (GPU idle before each call): host cost of one recorded-graph launch vs plain launches
tctx = model._ctx[('train', str(dev), 9, 9)]
x = torch.rand(128, 1, 9, 9, device=dev); sig = torch.rand(128, device=dev) + 0.1; out = torch.empty_like(x)
flat = torch.empty(sum(p.numel() for p in model.parameters()), device=dev); gout = torch.randn_like(x)
for name, fn in (('train_forward', lambda: tctx.train_forward(x, sig, lab, out, 0.2, 123)), ('backward', lambda: tctx.backward(gout, flat, x))):
    for _ in range(3): fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0); torch.cuda.synchronize()
    print(f'{name}: host {sorted(ts)[5]*1e3:.3f} ms per call (median of 10)', flush=True)
print('graph stats', tctx.train_graph_stats())
