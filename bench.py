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
GFLOP_PER_FORWARD_8X9 = 0.207374848    # 103 687 424 MAC at 8x9
TRAFFIC_JSON = os.path.join(ROOT, 'profiles', 'unet_traffic.json')   # written by scripts/pmc_traffic_to_json.py from rocprofv3 --pmc passes


def csrc_sha16():
    """Hash of the kernel sources: a PMC traffic figure is only quoted for the binary it was measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'optimized-diffusion-model_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        with open(os.path.join(d, f), 'rb') as fh:
            h.update(f.encode()); h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(kernel, B, hw):
    """(bytes per launch | None, note): the HBM-side bytes of the dominant kernel come from separate rocprofv3 --pmc passes
    (FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE; they cannot be collected inside this
    process).  The figure is emitted only when it was measured on these kernel sources at this shape."""
    try:
        with open(TRAFFIC_JSON) as f:
            t = json.load(f)
    except Exception as e:                                       # noqa: BLE001
        return None, f'no {os.path.relpath(TRAFFIC_JSON, ROOT)} ({e.__class__.__name__})'
    if t.get('kernel') != kernel or t.get('batch') != B or t.get('pixels') != hw:
        return None, f"{os.path.relpath(TRAFFIC_JSON, ROOT)} holds {t.get('kernel')} at B={t.get('batch')}, {t.get('pixels')} px: not this run's shape"
    if t.get('csrc_sha16') != csrc_sha16():
        return None, f"stale: measured on csrc {t.get('csrc_sha16')} (commit {t.get('commit')}), this run is csrc {csrc_sha16()}"
    return t['bytes_per_launch'], (f"rocprofv3 --pmc passes on commit {t.get('commit')}: FETCH_SIZE {t.get('fetch_size_kb_mean')} KB x2 "
                                   f"(gfx950 correction) + WRITE_SIZE {t.get('write_size_kb_mean')} KB per launch; {t.get('source')}")


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
    ap.add_argument('--no-variants', action='store_true', help='skip the Langevin / [1,8,9] / training-step variant measurements')
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
    ctx0 = model._ctx[(str(dev), args.height, args.width)]
    coop_fallback = ctx0.coop_gave_up()
    if world > 1:                                  # every rank takes the same branch (the re-run below holds collectives)
        flag = torch.tensor([1.0 if coop_fallback else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        coop_fallback = bool(flag.item() > 0)
    if coop_fallback:
        # A co-operative launch gave up an inter-workgroup wait (the groups were not co-resident: a shared or partitioned device):
        # its samples are NaN and the context has switched to the single-sample program.  Time THAT, and say so.
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
    # N > 1: how many ranks really took part (a sum of ones over the RCCL group) and what the one collective of the path costs alone
    ranks_seen, gather_us = 1, None
    if world > 1:
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        shard = x[rank * B:(rank + 1) * B].contiguous()
        full = torch.empty_like(x)
        for _ in range(3):
            dist.all_gather_into_tensor(full, shard)
        sync(); tg = time.perf_counter()
        for _ in range(20):
            dist.all_gather_into_tensor(full, shard)
        torch.cuda.synchronize()
        gather_us = (time.perf_counter() - tg) / 20 * 1e6

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
        'plan': ctx0.path_info(), 'coop_fallback': coop_fallback, 'ranks_seen': ranks_seen, 'all_gather_us': gather_us,
    }
    if rank == 0 and not args.no_roofline:
        out['roofline'] = roofline(ge, model, cfg, sde, shape, labels, dev, args, value / world, fwd_per_traj)
    if rank == 0 and world == 1 and not args.no_variants and not args.no_roofline:
        try:
            out['variants'] = variants(ge, dev, args, labels)
        except Exception as e:                                   # noqa: BLE001  (the headline above is already measured)
            out['variants'] = {'error': f'{e.__class__.__name__}: {e}'[:300]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(args, B)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def profile_pass(model, cfg, shape, labels, dev, corrector):
    """Instrumented pass: HIP events around every launch on the launch stream (csrc ProfScope), 60 updates of the batch."""
    from rdmi import sampling, sde_lib
    N = 61
    sde2 = sde_lib.RVESDE(cfg.sde.sigma_min, cfg.sde.sigma_max, N=N)
    fn = sampling.get_pc_sampler(sde2, shape, sampling.get_predictor('euler_maruyama'),
                                 sampling.get_corrector(corrector), sampling.get_denoiser('none'), cfg.sampling.snr,
                                 cfg.sampling.n_steps_each, 1e-5, dev, seed=7)
    fn(model, weight=0.0, class_labels=labels)                 # warm
    ctx = model._ctx[(str(dev), shape[2], shape[3])]
    ctx.set_profiling(True)
    fn(model, weight=0.0, class_labels=labels)
    torch.cuda.synchronize()
    prof = ctx.get_profile()
    ctx.set_profiling(False)
    return prof, ctx.path_info()


def roofline(ge, model, cfg, sde, shape, labels, dev, args, traj_per_s_per_gpu, fwd_per_traj):
    prof, plan = profile_pass(model, cfg, shape, labels, dev, args.corrector)
    total_ms = sum(p['ms'] for p in prof)
    dom = max(prof, key=lambda p: p['ms'])
    achieved = dom['flops'] / (dom['ms'] * 1e-3) / 1e12 if dom['ms'] > 0 else 0.0
    kernels = {p['kernel']: {'launches': p['launches'], 'avg_us': 1e3 * p['ms'] / max(p['launches'], 1),
                             'share': p['ms'] / total_ms, 'tflops': (p['flops'] / (p['ms'] * 1e-3) / 1e12) if p['flops'] else None}
               for p in sorted(prof, key=lambda p: -p['ms'])}
    hbm_bytes_per_traj = fwd_per_traj * ACT_BYTES_PER_FORWARD + (fwd_per_traj / 2) * WEIGHT_BYTES / shape[0]
    traffic, note = measured_traffic(dom['kernel'], shape[0], shape[2] * shape[3])
    return {'bound': 'mfma', 'kernel': dom['kernel'], 'achieved': achieved, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': achieved / PEAK_FP32_MFMA_TFLOPS, 'traffic': traffic, 'traffic_note': note,
            'avg_launch_us': 1e3 * dom['ms'] / max(dom['launches'], 1), 'launches': dom['launches'], 'plan': plan,
            'whole_path_frac_of_fp32_peak': traj_per_s_per_gpu * fwd_per_traj * GFLOP_PER_FORWARD / 1e3 / PEAK_FP32_MFMA_TFLOPS,
            'whole_path_layer_granular_hbm_frac': traj_per_s_per_gpu * hbm_bytes_per_traj / 1e9 / PEAK_HBM_GBPS,
            'kernels': kernels}


def variants(ge, dev, args, labels):
    """The other configurations BASELINE.md 5 asks for, measured in this same run (rank 0, N=1): C2' Langevin corrector,
    the [1,8,9] shape BASELINE.json names, and C4 the score-matching training step at B=128.  Sampler variants: one warm
    call at a short schedule, then ONE full 1000-scale call timed device-synchronised; `frac` = the dominant kernel's
    algorithmic FLOP over its mean HIP-event launch time, as for the headline."""
    from rdmi import sampling, sde_lib
    B = args.batch
    out = {}
    for name, corr, H, W in (('langevin', 'langevin', 9, 9), ('shape_8x9', 'none', 8, 9)):
        model, cfg, _ = ge.make_model(dev, corrector=corr, num_scales=args.num_scales, image_size=9, image_width=W)
        shape = (B, 1, H, W)
        prof, plan = profile_pass(model, cfg, shape, labels, dev, corr)
        dom = max(prof, key=lambda p: p['ms'])
        sde = sde_lib.RVESDE(cfg.sde.sigma_min, cfg.sde.sigma_max, N=cfg.sde.num_scales)
        fn = sampling.get_sampling_fn(cfg, sde, shape, 1e-5, dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x, nfe = fn(model, weight=0.0, class_labels=labels)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        assert bool(torch.isfinite(x).all()) and float(x.min()) >= 0 and float(x.max()) <= 1
        gf = GFLOP_PER_FORWARD if H * W == 81 else GFLOP_PER_FORWARD_8X9
        out[name] = {'value': B / dt, 'unit': 'trajectories/s', 'ms_per_call': 1e3 * dt, 'nfe': nfe,
                     'kernel': dom['kernel'], 'avg_launch_us': 1e3 * dom['ms'] / max(dom['launches'], 1),
                     'frac': (dom['flops'] / (dom['ms'] * 1e-3) / 1e12) / PEAK_FP32_MFMA_TFLOPS,
                     'whole_path_frac_of_fp32_peak': (B / dt) * 2 * (args.num_scales - 1) * (2 if corr == 'langevin' else 1) * gf / 1e3 / PEAK_FP32_MFMA_TFLOPS}
        del model
    # A/B of the co-operative program (default at this batch): the same workload on the one-sample-per-workgroup program
    os.environ['RDMI_COOP'] = '0'
    try:
        model, cfg, _ = ge.make_model(dev, num_scales=args.num_scales)
        sde = sde_lib.RVESDE(cfg.sde.sigma_min, cfg.sde.sigma_max, N=cfg.sde.num_scales)
        fn = sampling.get_sampling_fn(cfg, sde, (B, 1, 9, 9), 1e-5, dev)
        warm = sampling.get_sampling_fn(cfg, sde_lib.RVESDE(cfg.sde.sigma_min, cfg.sde.sigma_max, N=6), (B, 1, 9, 9), 1e-5, dev)
        warm(model, weight=0.0, class_labels=labels)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x, nfe = fn(model, weight=0.0, class_labels=labels)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out['coop_off'] = {'value': B / dt, 'unit': 'trajectories/s', 'ms_per_call': 1e3 * dt, 'plan': model._ctx[(str(dev), 9, 9)].path_info()}
        del model
    finally:
        os.environ.pop('RDMI_COOP', None)
    # the reference's own operating point is B = 4096-8192 per GPU: from 512 model samples on, the
    # low-resolution half of the U-Net runs for 2 / 4 samples per workgroup (csrc/rdmi.hip: build_fused_program)
    for name, Bv, Nv in (('batch_1024', 1024, 1000), ('batch_4096', 4096, 251)):
        model, cfg, _ = ge.make_model(dev, num_scales=Nv)
        sde = sde_lib.RVESDE(cfg.sde.sigma_min, cfg.sde.sigma_max, N=Nv)
        lab = torch.rand(Bv, 1, generator=torch.Generator().manual_seed(3)).to(dev)
        fn = sampling.get_sampling_fn(cfg, sde, (Bv, 1, 9, 9), 1e-5, dev)
        warm = sampling.get_sampling_fn(cfg, sde_lib.RVESDE(cfg.sde.sigma_min, cfg.sde.sigma_max, N=6), (Bv, 1, 9, 9), 1e-5, dev)
        warm(model, weight=0.0, class_labels=lab)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x, nfe = fn(model, weight=0.0, class_labels=lab)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        assert bool(torch.isfinite(x).all()) and float(x.min()) >= 0 and float(x.max()) <= 1
        per_1000 = dt / (Nv - 1) * 999                          # a shorter schedule is scaled per update (every update costs the same)
        out[name] = {'value': Bv / per_1000, 'unit': 'trajectories/s', 'batch': Bv, 'num_scales_run': Nv, 'ms_per_update': 1e3 * dt / (Nv - 1),
                     'plan': model._ctx[(str(dev), 9, 9)].path_info(),
                     'whole_path_frac_of_fp32_peak': (Bv / per_1000) * 2 * 999 * GFLOP_PER_FORWARD / 1e3 / PEAK_FP32_MFMA_TFLOPS}
        del model
    # the secondary configurations are isolated: a failure there (e.g. memory on a shared box) is reported in place and never costs
    # the headline line
    for name, fn_ in (('cifar_b64', lambda: cifar_variant(ge, dev, 64)), ('cifar_b64_bf16', lambda: cifar_variant(ge, dev, 64, dtype='bf16')),
                      ('train_b128', lambda: train_variant(ge, dev, 128)), ('train_b128_bf16', lambda: train_variant(ge, dev, 128, dtype='bf16')),
                      ('train_b4096_bf16', lambda: train_variant(ge, dev, 4096, dtype='bf16', steps=5))):
        try:
            out[name] = fn_()
        except Exception as e:                                   # noqa: BLE001
            out[name] = {'error': f'{e.__class__.__name__}: {e}'[:300]}
        torch.cuda.empty_cache()
    return out


def cifar_variant(ge, dev, B, N=5, dtype='f32'):
    """BASELINE config #5: the CIFAR-shape NCSN++ (32x32x3, nf=128, ch_mult [1,2,2,2], 8 res blocks, attention at 16x16, 104.7 M
    parameters, 36.9 GFLOP per sample-forward) through the tiled plan, fp32, guidance path on (2B forwards per update).  A
    1000-scale trajectory of this model is ~1 min per batch of 64: N-1 = 4 updates are timed and scaled per update.  dtype 'bf16':
    bf16 MFMA operands with fp32 accumulate (the BASELINE line asks bf16); `frac` is against the peak of the MFMA dtype used."""
    from rdmi import sampling, sde_lib
    GF = 36.912                                            # 18.456 GMAC per sample-forward
    model, cfg, _ = ge.make_cifar_model(dev, num_scales=N, compute_dtype=dtype)
    sde = sde_lib.RVESDE(cfg.sde.sigma_min, cfg.sde.sigma_max, N=N)
    lab = torch.zeros(B, 1, device=dev)
    fn = sampling.get_sampling_fn(cfg, sde, (B, 3, 32, 32), 1e-5, dev)
    fn(model, weight=0.0, class_labels=lab)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x, nfe = fn(model, weight=0.0, class_labels=lab)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (N - 1)
    assert bool(torch.isfinite(x).all()) and float(x.min()) >= 0 and float(x.max()) <= 1
    tf = 2 * B * GF / dt / 1e3
    return {'value': B / (dt * 999), 'unit': 'trajectories/s (scaled to 1000 scales from %d timed updates)' % (N - 1), 'batch': B,
            'ms_per_update': 1e3 * dt, 'dtype': dtype, 'tflops': tf, 'frac': tf / (PEAK_FP32_MFMA_TFLOPS if dtype == 'f32' else 2500.0),
            'peak_tflops': PEAK_FP32_MFMA_TFLOPS if dtype == 'f32' else 2500.0,
            'plan': model._ctx[(str(dev), 32, 32)].path_info()}


def train_variant(ge, dev, B, steps=10, dtype='f32'):
    """BASELINE config #4 shape: losses.get_step_fn(train=True) -- HIP forward (dropout 0.2, label drop 0.5) + loss + HIP
    backward + clip + Adam + EMA -- at batch B.  FLOP = 3 x forward (forward, data gradient, weight gradient)."""
    from rdmi import losses, sde_lib
    from rdmi.models.ema import ExponentialMovingAverage
    model, cfg, _ = ge.make_model(dev)
    model.train_dtype = dtype          # 'bf16': bf16 weight copies + bf16 MFMA operands in fwd / dgrad / wgrad, fp32 accumulate and master weights
    model.train()
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    opt = losses.get_optimizer(cfg, model.parameters())
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    state = dict(optimizer=opt, model=model, ema=ema, step=0, scaler=None)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=False,
                                 likelihood_weighting=False)
    g = torch.Generator().manual_seed(5)
    batch = torch.rand(B, 1, 9, 9, generator=g).to(dev); lab = torch.rand(B, 1, generator=g).to(dev)
    for _ in range(3):
        loss = step_fn(state, batch, class_labels=lab)
    dts = []
    for _ in range(3):                                            # three timed groups of `steps` back-to-back steps; the median group is reported
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            loss = step_fn(state, batch, class_labels=lab)
        torch.cuda.synchronize(); dts.append((time.perf_counter() - t0) / steps)
    dt = sorted(dts)[1]
    assert bool(torch.isfinite(loss.detach()))
    tf = 3 * GFLOP_PER_FORWARD * B / dt / 1e3
    hbm = {}
    try:            # HBM bytes of one step (every kernel; PMC FETCH_SIZE x2 + WRITE_SIZE, scripts/gpu_pmc_train.sh), valid for these kernel sources only
        with open(os.path.join(ROOT, 'profiles', 'train_traffic.json')) as f:
            t = json.load(f)
        if t.get('batch') == B and t.get('dtype') == dtype:
            if t.get('csrc_sha16') == csrc_sha16():
                hbm = {'hbm_bytes_per_step': t['bytes_per_step'], 'hbm_gbps': t['bytes_per_step'] / dt / 1e9}
            else:
                hbm = {'hbm_bytes_per_step': None, 'hbm_note': f"stale: measured on csrc {t.get('csrc_sha16')}"}
    except Exception:                                            # noqa: BLE001
        pass
    return {**hbm, 'ms_per_step': 1e3 * dt, 'samples_per_s': B / dt, 'batch': B, 'dtype': dtype, 'tflops': tf,
            'frac': tf / (PEAK_FP32_MFMA_TFLOPS if dtype == 'f32' else 2500.0), 'peak_tflops': PEAK_FP32_MFMA_TFLOPS if dtype == 'f32' else 2500.0,
            'loss': float(loss.detach()),
            'ms_per_step_groups': [round(1e3 * d, 3) for d in dts],
            'what': 'train-mode forward + score-matching loss + backward + clip_grad_norm_ + Adam + EMA; three groups of steps timed back to back, median group reported'}


def cpu_baseline(args, B):
    """torch-CPU restatement (oracle/rd_oracle_torch.py) on a bounded sample of the same workload.  81-pixel convolutions do
    not scale to every core of a large host (oneDNN over-threading made 128 threads slower than 8 in round 1), so a short
    sweep over thread counts is timed and the BEST is reported with its `cores`."""
    import numpy as np  # noqa: F401
    from oracle import rd_oracle as O
    from oracle import rd_oracle_torch as OT
    from oracle.weights import make_params
    p = {k: torch.from_numpy(v) for k, v in make_params(0).items()}
    g = torch.Generator().manual_seed(1)
    x0 = torch.rand(B, 1, args.height, args.width, generator=g)
    lab = torch.rand(B, 1, generator=g)
    w = torch.zeros(B)
    ts = O.torch_linspace(1, 1e-5, args.num_scales)
    lang = args.corrector == 'langevin'
    ncpu = os.cpu_count() or 1
    sweep = sorted({min(n, ncpu) for n in (8, 16, 32, 64, 128)})
    results = {}
    best = None
    for nthr in sweep:
        torch.set_num_threads(nthr)
        x = x0.clone()
        n, t_used = 0, 0.0
        with torch.no_grad():
            OT.pc_update(p, x, torch.full((B,), float(ts[0])), lab, w, torch.randn(x.shape, generator=g), args.num_scales)   # warm
            while t_used < 4.0 and n < 200:
                t = torch.full((B,), float(ts[min(n * 5, args.num_scales - 2)]))
                zp = torch.randn(x.shape, generator=g)
                zc = torch.randn(x.shape, generator=g) if lang else None
                t0 = time.perf_counter()
                x = OT.pc_update(p, x, t, lab, w, zp, args.num_scales, z_corr=zc)
                t_used += time.perf_counter() - t0
                n += 1
        v = B / ((t_used / n) * (args.num_scales - 1))
        results[str(nthr)] = round(v, 4)
        if best is None or v > best[0]:
            best = (v, nthr, n, t_used)
    # ... and ONE full run of the whole schedule at the best thread count ("run the full 1000 steps once if it fits")
    full_s, full_v = None, None
    if not lang and os.environ.get('RDMI_BENCH_NO_FULL_CPU') is None:
        torch.set_num_threads(best[1])
        x = x0.clone()
        t0 = time.perf_counter()
        with torch.no_grad():
            for i in range(args.num_scales - 1):
                x = OT.pc_update(p, x, torch.full((B,), float(ts[i])), lab, w, torch.randn(x.shape, generator=g), args.num_scales)
        full_s = time.perf_counter() - t0
        full_v = B / full_s
        assert bool(torch.isfinite(x).all()) and float(x.min()) >= 0 and float(x.max()) <= 1
    return {'value': best[0], 'unit': 'trajectories/s', 'cores': best[1], 'kind': 'port', 'host_cpus': ncpu,
            'threads_sweep': results, 'full_run_s': full_s, 'full_run_value': full_v,
            'sample': f'{best[2]} PC updates of the same B={B} CFG batch ({best[3]:.1f} s) per thread count, scaled to '
                      f'{args.num_scales - 1} updates; best of the sweep'}


if __name__ == '__main__':
    main()
