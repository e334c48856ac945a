"""Experiment driver: time the fused U-Net kernel of one experimental build (variants/librdmi_NAME.so) at the bench shape.
usage: gpu_variant_bench.py LIB [num_scales] [B]  -> one line: name, unet avg us, traj/s-equivalent, max |diff| vs default lib output file"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'optimized-diffusion-model_amd'))
import torch
from rdmi import _native
lib = sys.argv[1]
_native.use_library(lib)
import __graft_entry__ as ge
from rdmi import sampling, sde_lib
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device('cuda:0')
model, cfg, _ = ge.make_model(dev, num_scales=N)
sde = sde_lib.RVESDE(0.01, 5, N=N)
g = torch.Generator().manual_seed(1234)
lab = torch.rand(B, 1, generator=g).to(dev)
prior = torch.rand(B, 1, 9, 9, generator=g)
fn = sampling.get_pc_sampler(sde, (B, 1, 9, 9), sampling.get_predictor('euler_maruyama'), sampling.get_corrector('none'),
                             sampling.get_denoiser('none'), 0.01, 1, 1e-5, dev, seed=99)
_rand = torch.rand
torch.rand = lambda *a, **k: prior.clone()
x, _ = fn(model, weight=0.0, class_labels=lab)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.time(); x, _ = fn(model, weight=0.0, class_labels=lab); torch.cuda.synchronize(); ts.append(time.time() - t0)
torch.rand = _rand
best = min(ts)
ref_path = os.path.join(ROOT, 'gpurun_out', f'variant_ref_{N}_{B}.pt')
diff = float('nan')
if os.path.exists(ref_path):
    diff = float((torch.load(ref_path) - x.cpu()).abs().max())
else:
    torch.save(x.cpu(), ref_path)
print(f'{os.path.basename(lib):40s} {best / (N - 1) * 1e6:9.2f} us/update  {B / (best / (N - 1) * 999):8.2f} traj/s@1000  maxdiff_vs_first {diff:.3e}', flush=True)
