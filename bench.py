"""bench.py -- trajectories/sec of the 1000-step reflected PC sampler (NCSN++ on [B,1,9,9] GTO-Halo latents).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched under torch.distributed.run,
one rank per GPU (RCCL).  A "step" = one full sampling_fn call: N_scales-1 = 999 reflected PC updates of a
batch of 128 trajectories per GPU (BASELINE.json configs[1]; classifier-free-guidance path on => 256 network
forwards per update), followed for N>1 by the single all-gather of the samples (configs[2], weak scaling:
128 trajectories per GPU).  Prints ONE JSON line on rank 0.

Extra objects on that line:
  roofline     -- dominant kernel (the fp32-MFMA implicit-GEMM conv) algorithmic FLOP / its mean launch time,
                  measured with HIP events on the launch stream in an instrumented pass of the same workload
                  right after the timed region (per-launch events inside the timed region would perturb it);
                  peak = 157.3 TFLOP/s fp32 matrix (MI355X_MICROARCH.md).  bound = "mfma": the path is
                  compute-bound (48.6 FLOP/B layer-granular), so the HBM fraction is reported beside it.
  cpu_baseline -- oracle/rd_oracle_torch.py (a torch-CPU restatement pinned to the reference's fixtures; the
                  reference itself cannot travel to the GPU box) timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'optimized-diffusion-model_amd'))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0
GFLOP_PER_FORWARD = 0.220438528        # 110 219 264 MAC per sample-forward at 9x9
ACT_BYTES_PER_FORWARD = 4440464        # layer-granular activation bytes per sample-forward
WEIGHT_BYTES = 25019652
PMC_TRAFFIC_BYTES_UNET_B128 = (2 * 98287 + 15576) * 1024   # profiles/r01_pmc_hbm_traffic_unet_wg_kernel.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=128, help='trajectories per GPU')
    ap.add_argument('--num-scales', type=int, default=1000)
    ap.add_argument('--corrector', default='none', choices=['none', 'langevin'])
    ap.add_argument('--height', type=int, default=9)
    ap.add_argument('--width', type=int, default=9)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # Rehearsal on a box with fewer GPUs than ranks (never what the driver runs): RDMI_BENCH_BACKEND=gloo puts every rank
    # on cuda:(local % device_count) and exchanges through gloo, to exercise this file's N>1 control flow.
    backend = os.environ.get('RDMI_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(f'cuda:{local}'))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 under torch.distributed.run')
    torch.cuda.set_device(local)
    dev = torch.device(f'cuda:{local}')

    import __graft_entry__ as ge
    if int(os.environ.get('LOCAL_RANK', '0')) == 0:
        ge.build()                    # one rank per node compiles (no-op when librdmi.so is fresh) ...
    if world > 1:
        dist.barrier()
    ge.build()                        # ... everyone loads
    from rdmi import sampling, sde_lib
    from rdmi.parallel import sharded_sampling_fn

    model, cfg, _ = ge.make_model(dev, corrector=args.corrector, num_scales=args.num_scales,
                                  image_size=9, image_width=args.width)
    B = args.batch
    shape = (B, 1, args.height, args.width)
    sde = sde_lib.RVESDE(cfg.sde.sigma_min, cfg.sde.sigma_max, N=cfg.sde.num_scales)
    torch.manual_seed(1234)
    labels_all = torch.rand(B * world, 1)                       # Benchmark harness: class_labels = U[0,1](B,1)
    labels = labels_all[rank * B:(rank + 1) * B].to(dev)
    sampling_fn = sharded_sampling_fn(cfg, sde, shape, 1e-5, dev, seed=1000, rank=rank, world=world)

    def step():
        return sampling_fn(model, weight=0.0, class_labels=labels)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x, nfe = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert x.shape[0] == B * world and bool(torch.isfinite(x).all()) and float(x.min()) >= 0 and float(x.max()) <= 1

    n_corr = 1 if args.corrector == 'langevin' else 0
    evals = (args.num_scales - 1) * (1 + n_corr)               # score evaluations per trajectory
    fwd_per_traj = 2 * evals                                    # CFG: two forwards per evaluation
    value = B * world * args.steps / elapsed
    out = {
        'metric': 'trajectories/sec (1000-step PC sampler, NCSN++ [1,9,9], bs=128/GPU)',
        'value': value, 'unit': 'trajectories/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic (seeded non-degenerate weights, U[0,1] prior and labels, Philox noise)',
        'config': {'workload': f'NCSN++ nf=64 ch_mult=[1,2,2] on [{B},1,{args.height},{args.width}] per GPU, '
                               f'{args.num_scales}-step reflected PC sampler (predictor euler_maruyama, corrector '
                               f'{args.corrector}), classifier-free guidance on (2B-sample forwards), fp32',
                   'global_batch': B * world, 'num_scales': args.num_scales, 'score_evals_per_traj': evals,
                   'parallelism': f'batch-sharded x{world}, one all-gather per sampling call' if world > 1 else 'single GPU'},
        'tflops_algorithmic': value * fwd_per_traj * GFLOP_PER_FORWARD / 1e3 / world,
    }
    if rank == 0 and not args.no_roofline:
        out['roofline'] = roofline(ge, model, cfg, sde, shape, labels, dev, args, value / world, fwd_per_traj)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(args, B)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def roofline(ge, model, cfg, sde, shape, labels, dev, args, traj_per_s_per_gpu, fwd_per_traj):
    """Instrumented pass: HIP events around every launch (csrc ProfScope), 60 updates of the same batch."""
    from rdmi import sampling, sde_lib
    N = 61
    sde2 = sde_lib.RVESDE(cfg.sde.sigma_min, cfg.sde.sigma_max, N=N)
    fn = sampling.get_pc_sampler(sde2, shape, sampling.get_predictor('euler_maruyama'),
                                 sampling.get_corrector(args.corrector), sampling.get_denoiser('none'), cfg.sampling.snr,
                                 cfg.sampling.n_steps_each, 1e-5, dev, seed=7)
    fn(model, weight=0.0, class_labels=labels)                 # warm
    ctx = model._ctx[(str(dev), shape[2], shape[3])]
    ctx.set_profiling(True)
    fn(model, weight=0.0, class_labels=labels)
    torch.cuda.synchronize()
    prof = ctx.get_profile()
    ctx.set_profiling(False)
    plan = ctx.path_info()
    total_ms = sum(p['ms'] for p in prof)
    dom = max(prof, key=lambda p: p['ms'])
    achieved = dom['flops'] / (dom['ms'] * 1e-3) / 1e12 if dom['ms'] > 0 else 0.0
    kernels = {p['kernel']: {'launches': p['launches'], 'avg_us': 1e3 * p['ms'] / max(p['launches'], 1),
                             'share': p['ms'] / total_ms, 'tflops': (p['flops'] / (p['ms'] * 1e-3) / 1e12) if p['flops'] else None}
               for p in sorted(prof, key=lambda p: -p['ms'])}
    hbm_bytes_per_traj = fwd_per_traj * ACT_BYTES_PER_FORWARD + (fwd_per_traj / 2) * WEIGHT_BYTES / shape[0]
    # HBM-side bytes per launch of the dominant kernel come from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
    # they cannot be collected from inside this process): profiles/r01_pmc_hbm_traffic_unet_wg_kernel.md, B=128 shape only.
    traffic = PMC_TRAFFIC_BYTES_UNET_B128 if (dom['kernel'] == 'unet_wg_kernel' and shape[0] == 128 and shape[2] * shape[3] == 81) else None
    return {'bound': 'mfma', 'kernel': dom['kernel'], 'achieved': achieved, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': achieved / PEAK_FP32_MFMA_TFLOPS, 'traffic': traffic,
            'traffic_note': 'bytes/launch from rocprofv3 PMC passes of this command, FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE; see profiles/',
            'avg_launch_us': 1e3 * dom['ms'] / max(dom['launches'], 1), 'launches': dom['launches'], 'plan': plan,
            'whole_path_frac_of_fp32_peak': traj_per_s_per_gpu * fwd_per_traj * GFLOP_PER_FORWARD / 1e3 / PEAK_FP32_MFMA_TFLOPS,
            'whole_path_layer_granular_hbm_frac': traj_per_s_per_gpu * hbm_bytes_per_traj / 1e9 / PEAK_HBM_GBPS,
            'kernels': kernels}


def cpu_baseline(args, B):
    """torch-CPU restatement (oracle/rd_oracle_torch.py), all host threads, a bounded sample of the same workload."""
    import numpy as np
    from oracle import rd_oracle as O
    from oracle import rd_oracle_torch as OT
    from oracle.weights import make_params
    p = {k: torch.from_numpy(v) for k, v in make_params(0).items()}
    g = torch.Generator().manual_seed(1)
    x = torch.rand(B, 1, args.height, args.width, generator=g)
    lab = torch.rand(B, 1, generator=g)
    w = torch.zeros(B)
    ts = O.torch_linspace(1, 1e-5, args.num_scales)
    lang = args.corrector == 'langevin'
    n, t_used = 0, 0.0
    with torch.no_grad():
        OT.pc_update(p, x, torch.full((B,), float(ts[0])), lab, w, torch.randn(x.shape, generator=g), args.num_scales)   # warm
        while t_used < 12.0 and n < 200:
            t = torch.full((B,), float(ts[min(n * 5, args.num_scales - 2)]))
            zp = torch.randn(x.shape, generator=g)
            zc = torch.randn(x.shape, generator=g) if lang else None
            t0 = time.perf_counter()
            x = OT.pc_update(p, x, t, lab, w, zp, args.num_scales, z_corr=zc)
            t_used += time.perf_counter() - t0
            n += 1
    per_update = t_used / n
    return {'value': B / (per_update * (args.num_scales - 1)), 'unit': 'trajectories/s', 'cores': torch.get_num_threads(),
            'kind': 'port', 'host_cpus': os.cpu_count(),
            'sample': f'{n} PC updates of the same B={B} CFG batch ({t_used:.1f} s), scaled to {args.num_scales - 1} updates'}


if __name__ == '__main__':
    main()
