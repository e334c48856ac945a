"""Launch-floor experiment: same sampler on the legacy null stream vs a non-blocking stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'optimized-diffusion-model_amd'))
import torch
import __graft_entry__ as ge
from rdmi import sampling, sde_lib
dev = torch.device('cuda:0')
model, cfg, _ = ge.make_model(dev, num_scales=101, corrector='none')
B = 128
sde = sde_lib.RVESDE(0.01, 5, N=101)
lab = torch.rand(B, 1, device=dev)
fn = sampling.get_pc_sampler(sde, (B,1,9,9), sampling.get_predictor('euler_maruyama'), sampling.get_corrector('none'), sampling.get_denoiser('none'), 0.01, 1, 1e-5, dev, seed=3)
def run(tag):
    fn(model, weight=0.0, class_labels=lab)
    torch.cuda.synchronize(); t0=time.perf_counter()
    fn(model, weight=0.0, class_labels=lab)
    torch.cuda.synchronize(); print(tag, 'ms/update', (time.perf_counter()-t0)*1e3/100, flush=True)
run('null stream')
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run('torch.cuda.Stream')
