"""Diagnostic: per-op cycles of the fused U-Net kernel (workgroup 0) at the bench shape."""
import os, sys
os.environ['RDMI_STAMPS'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'optimized-diffusion-model_amd'))
import torch
from rdmi import _native
if os.environ.get('RDMI_VARIANT_LIB'): _native.use_library(os.environ['RDMI_VARIANT_LIB'])     # experimental build (scripts/build_variant.sh)
import __graft_entry__ as ge
from rdmi import sde_lib
from rdmi.models import utils as mutils
dev = torch.device('cuda:0')
model, cfg, _ = ge.make_model(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
QUIET = len(sys.argv) > 2
sde = sde_lib.RVESDE(0.01, 5, N=1000)
x = torch.rand(B, 1, 9, 9, device=dev); t = torch.full((B,), 0.5, device=dev); lab = torch.rand(B, 1, device=dev)
fn = mutils.get_cf_score_fn(sde, model, lab, 0.0)
with torch.no_grad():
    for _ in range(5):
        fn(x, t)
torch.cuda.synchronize()
ctx = model._ctx[(str(dev), 9, 9)]
print(ctx.path_info())
ops = ctx.op_cycles()
tot = sum(c for _, c in ops)
agg = {}
for i, (d, c) in enumerate(ops):
    if not QUIET: print(f'{i:3d} {c:8d} {100*c/tot:5.1f}%  {d}' + (f'   fine(entry,ring,kind,prefetch,main,epi,gnbar,gnapply)={ctx.fine[i]}' if d.startswith('CONV') and i < len(ctx.fine) else ''))
    k = d.split()[0] + (' ' + d.split()[1] if d.startswith('CONV') else '')
    agg[k] = agg.get(k, 0) + c
print('total cycles', tot, ' (100 MHz ticks?)')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f'{k:24s} {v:9d} {100*v/tot:5.1f}%')
