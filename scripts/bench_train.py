"""Secondary measurement (BASELINE config #4 shape): score-matching training steps/s on one GPU.
usage: bench_train.py [B] [train_dtype f32|bf16].  Not the driver's bench.py; prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'optimized-diffusion-model_amd'))
import torch
import __graft_entry__ as ge
from rdmi import losses, sde_lib
from rdmi.models.ema import ExponentialMovingAverage
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
DT = sys.argv[2] if len(sys.argv) > 2 else 'f32'
model, cfg, _ = ge.make_model(dev)
model.train_dtype = DT
model.train()
sde = sde_lib.RVESDE(0.01, 5, N=1000)
opt = losses.get_optimizer(cfg, model.parameters())
ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
state = dict(optimizer=opt, model=model, ema=ema, step=0, scaler=None)
step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=False, likelihood_weighting=False)
batch = torch.rand(B, 1, 9, 9, device=dev); labels = torch.rand(B, 1, device=dev)
for _ in range(3):
    l = step_fn(state, batch, class_labels=labels)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 10
for _ in range(K):
    l = step_fn(state, batch, class_labels=labels)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(json.dumps({'metric': 'score-matching training step (dropout 0.2, label drop 0.5, Adam+clip+EMA)', 'train_dtype': DT, 'batch': B,
                  'ms_per_step': dt * 1e3, 'samples_per_s': B / dt, 'loss': float(l.detach())}))
