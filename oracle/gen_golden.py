"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

Imports the reference's own modules from /root/reference/Reflected-Diffusion
(SURVEY.md 8c / Appendix A), loads the seeded synthetic weights of
oracle/weights.py into the reference NCSNpp with strict=True, and records
inputs + expected outputs.  Only data is written: no reference source text is
copied.  /root/reference does not exist on the GPU box; tests read the .npz.

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
"""
import hashlib
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/Reflected-Diffusion'
OUT = os.path.join(HERE, '..', 'tests', 'golden')
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

import cube, sde_lib, sampling, losses            # noqa: E402  (reference modules)
from models import utils as mutils               # noqa: E402
from models.ema import ExponentialMovingAverage  # noqa: E402
import weights as W                              # noqa: E402  (oracle/weights.py)

torch.set_num_threads(8)


def make_cfg(image_size=9, image_width=9, dropout=0.2, cond_drop_prob=0.5, corrector='none', num_scales=1000, warmup=10000):
    return NS(
        model=NS(name='ncsnpp', channels=1, image_size=image_size, image_width=image_width, num_classes=1,
                 cond_drop_prob=cond_drop_prob, conditional=True, init_scale=0., ema_rate=0.999, nf=64,
                 ch_mult=[1, 2, 2], num_res_blocks=2, attn_resolutions=[9], resamp_with_conv=True,
                 embedding_type='fourier', fourier_scale=16, resblock_type='ddpm', skip_rescale=True,
                 nonlinearity='swish', fir=False, fir_kernel=[1, 3, 3, 1], dropout=dropout, scale_by_sigma=False),
        sampling=NS(method='pc', n_steps_each=1, noise_removal=True, probability_flow=False, snr=0.01,
                    predictor='euler_maruyama', corrector=corrector, denoiser='none'),
        sde=NS(sigma_min=0.01, sigma_max=5, num_scales=num_scales),
        optim=NS(weight_decay=0, optimizer='Adam', lr=5e-4, beta1=0.9, beta2=0.999, eps=1e-8, warmup=warmup,
                 grad_clip=0.5),
        training=NS(reduce_mean=False, likelihood_weighting=False))


def ref_model(cfg, seed=0):
    model = mutils.create_model(cfg)
    params = W.make_params(seed)
    sd = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    assert list(sd.keys()) == list(model.state_dict().keys()), 'state-dict order/name mismatch'
    model.load_state_dict(sd, strict=True)
    return model.eval(), params


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB, keys={len(arrs)}')


def gen_cube_sde():
    known_in = np.array([-2.3, -1.2, -0.3, 1.2, 2.3, 3.7, 1.0, -1.0, 2.0, 0.0, -2.0, 4.0, 0.5, 1e-9, -1e-9, 1.9999999],
                        np.float32)
    g = torch.Generator().manual_seed(11)
    rnd = (torch.rand(4, 1, 9, 9, generator=g) * 14 - 7)
    out = dict(reflect_known_in=known_in, reflect_known_out=cube.reflect(torch.from_numpy(known_in.copy())).numpy(),
               reflect_rand_in=rnd.numpy(), reflect_rand_out=cube.reflect(rnd.clone()).numpy())
    # score_hk across the t = sigma^2/2 = 1e-2 switch (sigma = 0.14142)
    sig = torch.tensor([0.01, 0.03, 0.08, 0.12, 0.1414, 0.1415, 0.2, 0.5, 1.0, 2.5, 5.0, 0.05], dtype=torch.float32)
    x0 = torch.rand(sig.numel(), 1, 9, 9, generator=g)
    z = torch.randn(sig.numel(), 1, 9, 9, generator=g)
    x = cube.reflect(x0 + sig[:, None, None, None] * z)
    out.update(hk_sigma=sig.numpy(), hk_x=x.numpy(), hk_x0=x0.numpy(), hk_score=cube.score_hk(x, x0, sig).numpy())
    out.update(hk_ef_only=cube._score_hk_ef(x, x0, sig ** 2 / 2).numpy(),
               hk_refl_only=cube._score_hk_refl(x, x0, sig ** 2 / 2, refls=10).numpy())
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    t = torch.tensor([1e-5, 1e-3, 0.1, 0.4263, 0.5, 0.9, 1.0], dtype=torch.float32)
    xz = torch.zeros(t.numel(), 1, 9, 9)
    out.update(sde_t=t.numpy(), sde_sigma=sde.marginal_prob(xz, t)[1].numpy(), sde_g=sde.sde(xz, t)[1].numpy(),
               ts_1000=torch.linspace(sde.T, 1e-5, 1000).numpy(), ts_10=torch.linspace(sde.T, 1e-5, 10).numpy())
    save('cube_sde.npz', **out)


TAPS = ['input_conv', 'down_blocks.0', 'down_attn.0', 'down_blocks.1', 'downsample.0', 'down_blocks.2',
        'down_blocks.3', 'downsample.1', 'down_blocks.5', 'mid_block1', 'mid_block2', 'up_blocks.0', 'up_blocks.2',
        'upsample.0', 'up_blocks.3', 'up_blocks.5', 'upsample.1', 'up_blocks.6', 'up_attn.6', 'up_blocks.8',
        'up_attn.8']


def gen_forward():
    cfg = make_cfg()
    model, params = ref_model(cfg)
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    g = torch.Generator().manual_seed(21)
    B = 8
    x = torch.rand(B, 1, 9, 9, generator=g)
    t = torch.tensor([1e-5, 0.1, 0.4262, 0.4264, 0.7, 0.9, 0.999, 1.0], dtype=torch.float32)
    labels = torch.rand(B, 1, generator=g)
    taps = {}
    mods = dict(model.named_modules())
    hooks = [mods[n].register_forward_hook(lambda m, i, o, n=n: taps.__setitem__(n, o[:2].detach().numpy().copy()))
             for n in TAPS]
    with torch.no_grad():
        score = mutils.get_score_fn(sde, model)(x, t, class_labels=labels)
    for h in hooks:
        h.remove()
    with torch.no_grad():
        temb = model.time_mlp(model.time_embed(torch.log(sde.marginal_prob(x, t)[1]))) + model.label_emb(labels)
        cf0 = mutils.get_cf_score_fn(sde, model, labels, 0.0)(x, t)
        cfn = mutils.get_cf_score_fn(sde, model, labels, None)(x, t)
        wt = torch.linspace(-0.5, 3.0, B)
        cfw = mutils.get_cf_score_fn(sde, model, labels, wt)(x, t)
        model_out = mutils.get_model_fn(model)(x, sde.marginal_prob(x, t)[1], class_labels=labels)
    assert torch.equal(model_out, score)
    save('forward_9x9.npz', x=x.numpy(), t=t.numpy(), labels=labels.numpy(), score=score.numpy(), temb=temb.numpy(),
         cf_w0=cf0.numpy(), cf_none=cfn.numpy(), cf_wt=cfw.numpy(), wt=wt.numpy(),
         params_sha256=np.frombuffer(bytes.fromhex(W.params_sha256(params)), np.uint8),
         **{'tap.' + k: v for k, v in taps.items()})
    # 8x9 variant (BASELINE.json's named shape): image_size stays 9 so attention is kept
    x89 = torch.rand(4, 1, 8, 9, generator=g)
    t89 = torch.tensor([0.05, 0.3, 0.6, 1.0], dtype=torch.float32)
    l89 = torch.rand(4, 1, generator=g)
    with torch.no_grad():
        s89 = mutils.get_score_fn(sde, model)(x89, t89, class_labels=l89)
    save('forward_8x9.npz', x=x89.numpy(), t=t89.numpy(), labels=l89.numpy(), score=s89.numpy())


class Recorder:
    """Records every torch.rand / torch.randn_like the reference sampler draws."""

    def __init__(self):
        self.rand, self.randn = [], []
        self._rand, self._randn_like = torch.rand, torch.randn_like

    def __enter__(self):
        def rand(*a, **k):
            v = self._rand(*a, **k); self.rand.append(v.numpy().copy()); return v

        def randn_like(x, **k):
            v = self._randn_like(x, **k); self.randn.append(v.numpy().copy()); return v
        torch.rand, torch.randn_like = rand, randn_like
        return self

    def __exit__(self, *a):
        torch.rand, torch.randn_like = self._rand, self._randn_like


def gen_sampler():
    out = {}
    B, N = 8, 10
    g = torch.Generator().manual_seed(31)
    labels = torch.rand(B, 1, generator=g)
    wt = torch.linspace(0.0, 2.0, B)
    cases = [('none_w0', 'none', labels, 0.0), ('langevin_w0', 'langevin', labels, 0.0),
             ('none_wt', 'none', labels, wt), ('langevin_none', 'langevin', labels, None)]
    # (class_labels=None is not a runnable case: with conditional=True the reference calls label_emb(None))
    for tag, corr, lab, w in cases:
        cfg = make_cfg(corrector=corr, num_scales=N)
        model, _ = ref_model(cfg)
        sde = sde_lib.RVESDE(0.01, 5, N=N)
        fn = sampling.get_sampling_fn(cfg, sde, (B, 1, 9, 9), 1e-5, 'cpu')
        # trace per-update x by wrapping the predictor's reflect-returning update
        steps = []
        pred_cls = sampling.get_predictor('euler_maruyama')
        orig = pred_cls.update_fn

        def traced(self, x, t, _o=orig):
            r = _o(self, x, t); steps.append(r[0].numpy().copy()); return r
        pred_cls.update_fn = traced
        torch.manual_seed(1234)
        with Recorder() as rec:
            x, nfe = fn(model, weight=w, class_labels=lab)
        pred_cls.update_fn = orig
        assert len(rec.rand) == 2 and len(steps) == N - 1
        out[f'{tag}.prior'] = rec.rand[1]
        out[f'{tag}.noises'] = np.stack(rec.randn)
        out[f'{tag}.steps'] = np.stack(steps)
        out[f'{tag}.x'] = x.numpy()
        out[f'{tag}.nfe'] = np.int64(nfe)
        assert np.array_equal(steps[-1], x.numpy())
    out['labels'] = labels.numpy(); out['wt'] = wt.numpy()
    save('sampler_10step.npz', **out)


def gen_train():
    """One score-matching step with dropout=0 and cond_drop_prob=0 so the loss is a pure function of (t, z)."""
    cfg = make_cfg(dropout=0.0, cond_drop_prob=0.0)
    model, _ = ref_model(cfg)
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    B = 8
    g = torch.Generator().manual_seed(41)
    batch = torch.rand(B, 1, 9, 9, generator=g)
    labels = torch.rand(B, 1, generator=g)
    optimizer = losses.get_optimizer(cfg, model.parameters())
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    state = dict(optimizer=optimizer, model=model, ema=ema, step=0, scaler=None)
    optimize_fn = losses.optimization_manager(cfg)
    train_step = losses.get_step_fn(sde, train=True, optimize_fn=optimize_fn, reduce_mean=False,
                                    likelihood_weighting=False)
    eval_step = losses.get_step_fn(sde, train=False, optimize_fn=optimize_fn, reduce_mean=False,
                                   likelihood_weighting=False)
    out = dict(batch=batch.numpy(), labels=labels.numpy())
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    # make t straddle the score_hk switch (sde-time 0.4263) deterministically
    tvals = torch.tensor([0.02, 0.2, 0.41, 0.43, 0.6, 0.8, 0.95, 0.999], dtype=torch.float32)
    _rand = torch.rand
    torch.rand = lambda *a, **k: ((tvals - 1e-5) / (1 - 1e-5)).clone() if a and a[0] == B else _rand(*a, **k)
    try:
        for step in range(2):
            torch.manual_seed(77 + step)
            with Recorder() as rec:
                # Recorder wraps the patched rand above; randn_like gives z
                loss = train_step(state, batch, class_labels=labels)
            out[f'step{step}.t'] = (rec.rand[0] * (1 - 1e-5) + 1e-5).astype(np.float32)
            out[f'step{step}.z'] = rec.randn[0]
            out[f'step{step}.loss'] = loss.detach().numpy()
            if step == 0:
                gn = np.array([float(p.grad.norm()) for n, p in model.named_parameters() if p.requires_grad], np.float32)
                out['step0.grad_norms'] = gn          # AFTER clip_grad_norm_(0.5) (losses.py:39-40)
                for n in ['out_conv.weight', 'time_mlp.0.bias', 'down_blocks.0.Conv_0.bias', 'up_attn.8.NIN_3.W',
                          'label_emb.weight']:
                    out['step0.grad.' + n] = dict(model.named_parameters())[n].grad.numpy().copy()
        torch.manual_seed(99)
        with Recorder() as rec:
            ev = eval_step(state, batch, class_labels=labels)
        out['eval.t'] = (rec.rand[0] * (1 - 1e-5) + 1e-5).astype(np.float32)
        out['eval.z'] = rec.randn[0]
        out['eval.loss'] = ev.numpy()
    finally:
        torch.rand = _rand
    out['after2.out_conv.bias'] = model.out_conv.bias.detach().numpy().copy()
    out['after2.time_mlp.0.bias'] = model.time_mlp[0].bias.detach().numpy().copy()
    out['after2.ema.out_conv.bias'] = ema.shadow_params[-1].numpy().copy()
    out['param_names'] = np.array(names)
    save('train_step.npz', **out)


def gen_train_w0():
    """Optimizer / EMA fixture.  train_step.npz runs the shipped warmup=10000, whose learning rate is 0 at step 0 and 5e-8 at
    step 1 (RD/losses.py:36-38): the parameter update is below any useful tolerance, so a no-op optimize_fn would pass.
    Here warmup=0 (the ramp is skipped, lr = 5e-4 from the first step) and three steps are taken, so Adam's moments, the
    bias correction, clip_grad_norm_(0.5) and the EMA decay schedule min(0.999, (1+n)/(10+n)) all act at full size.
    Recorded: losses, and for a few parameters the value before, after each step, and the EMA shadow after each step."""
    cfg = make_cfg(dropout=0.0, cond_drop_prob=0.0, warmup=0)
    model, _ = ref_model(cfg)
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    B = 8
    g = torch.Generator().manual_seed(43)
    batch = torch.rand(B, 1, 9, 9, generator=g)
    labels = torch.rand(B, 1, generator=g)
    optimizer = losses.get_optimizer(cfg, model.parameters())
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    state = dict(optimizer=optimizer, model=model, ema=ema, step=0, scaler=None)
    train_step = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=False,
                                    likelihood_weighting=False)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    watch = ['out_conv.bias', 'out_conv.weight', 'time_mlp.0.bias', 'label_emb.weight', 'down_blocks.0.Conv_0.weight',
             'mid_block1.Dense_0.weight', 'up_attn.8.NIN_3.W', 'up_blocks.4.GroupNorm_0.weight', 'downsample.1.Conv_0.bias']
    idx = {n: i for i, (n, _) in enumerate(named)}
    out = dict(batch=batch.numpy(), labels=labels.numpy(), watch=np.array(watch))
    for n in watch:
        out['p0.' + n] = dict(named)[n].detach().numpy().reshape(-1)[:256].copy()      # first 256 values of each watched tensor
    tvals = torch.tensor([0.05, 0.15, 0.3, 0.45, 0.55, 0.7, 0.85, 0.97], dtype=torch.float32)
    _rand = torch.rand
    torch.rand = lambda *a, **k: ((tvals - 1e-5) / (1 - 1e-5)).clone() if a and a[0] == B else _rand(*a, **k)
    try:
        for step in range(3):
            torch.manual_seed(177 + step)
            with Recorder() as rec:
                loss = train_step(state, batch, class_labels=labels)
            out[f'step{step}.t'] = (rec.rand[0] * (1 - 1e-5) + 1e-5).astype(np.float32)
            out[f'step{step}.z'] = rec.randn[0]
            out[f'step{step}.loss'] = loss.detach().numpy()
            for n in watch:
                out[f'p{step + 1}.' + n] = dict(named)[n].detach().numpy().reshape(-1)[:256].copy()
                out[f'ema{step + 1}.' + n] = ema.shadow_params[idx[n]].numpy().reshape(-1)[:256].copy()
    finally:
        torch.rand = _rand
    out['num_updates'] = np.int64(ema.num_updates)
    save('train_step_w0.npz', **out)


def gen_ode():
    """the reference's probability-flow ODE sampler (RD/sampling.py:342-392, scipy RK45) on an injected prior.
    RVESDE(0.01, 0.5): with the shipped sigma_max=5 and these synthetic (untrained) weights the flow is chaotic -- the
    reference itself needs 5390 right-hand sides and leaves the cube -- so no two fp32 implementations could be compared;
    at sigma_max=0.5 (g(1)^2 = 2) it takes ~60 steps.  Recorded: prior, labels, final sample, nfev, every accepted time, and
    the float64 state at a few accepted steps (for one-step parity: restart from (t_k, y_k) with first_step = |t_k+1 - t_k|)."""
    from scipy import integrate as _integ
    cfg = make_cfg()
    model, _ = ref_model(cfg)
    sde = sde_lib.RVESDE(0.01, 0.5, N=1000)
    out = {}
    for tag, B in (('b4', 4), ('b1', 1)):
        g = torch.Generator().manual_seed(21 + B)
        z = (1 - 2e-2) * torch.rand(B, 1, 9, 9, generator=g) + 1e-2
        lab = torch.rand(B, 1, generator=g)
        fn = sampling.get_ode_sampler(sde, (B, 1, 9, 9), eps=1e-3 if B == 4 else 0.9, moll=200, side_eps=1e-2, device='cpu')
        captured = {}
        orig = _integ.solve_ivp

        def spy(*a, _o=orig, **k):
            sol = _o(*a, **k); captured['sol'] = sol; return sol
        sampling.integrate.solve_ivp = spy
        try:
            x, nfe = fn(model, z=z, weight=0.5, class_labels=lab)
        finally:
            sampling.integrate.solve_ivp = orig
        sol = captured['sol']
        ks = sorted({0, 1, min(5, len(sol.t) - 2), (len(sol.t) - 1) // 2, len(sol.t) - 2})
        out.update({f'{tag}.z': z.numpy(), f'{tag}.labels': lab.numpy(), f'{tag}.x': x.numpy(), f'{tag}.nfev': np.int64(nfe),
                    f'{tag}.t': sol.t, f'{tag}.ks': np.array(ks)})
        for k in ks:
            out[f'{tag}.y{k}'] = sol.y[:, k].copy()
            out[f'{tag}.y{k + 1}'] = sol.y[:, k + 1].copy()
    save('ode_rk45.npz', **out)


def make_cifar_cfg():
    """BASELINE config #5: RD/configs/model/ddpmpp.yaml completed with the keys the fork's NCSNpp.__init__ requires and the yaml
    lacks: channels=3, image_size=32, num_classes=1 with zero labels (conditional: True would otherwise call
    label_emb(None))."""
    cfg = make_cfg()
    m = cfg.model
    m.nf, m.ch_mult, m.num_res_blocks, m.attn_resolutions = 128, [1, 2, 2, 2], 8, [16]
    m.channels, m.image_size, m.image_width, m.scale_by_sigma, m.dropout = 3, 32, 32, True, 0.1
    cfg.sde.sigma_max = 50
    return cfg


CIFAR_ARCH = dict(nf=128, ch_mult=(1, 2, 2, 2), num_res_blocks=8, attn_resolutions=(16,), image_size=32, channels=3)


def gen_cifar():
    """Whole-forward golden of the CIFAR-shape NCSN++ (104.7 M parameters, 18.5 GMAC per sample): B=2, zero labels, plus a few
    intermediate activations captured by forward hooks (first / last block of each level, one attention block)."""
    cfg = make_cifar_cfg()
    model = mutils.create_model(cfg)
    params = W.make_params(0, **CIFAR_ARCH)
    sd = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    assert list(sd.keys()) == list(model.state_dict().keys()), 'state-dict order/name mismatch (CIFAR arch)'
    model.load_state_dict(sd, strict=True)
    model.eval()
    sde = sde_lib.RVESDE(0.01, 50, N=1000)
    g = torch.Generator().manual_seed(61)
    B = 2
    x = torch.rand(B, 3, 32, 32, generator=g)
    t = torch.tensor([0.35, 0.8])
    labels = torch.zeros(B, 1)
    taps = {}
    names = {'down_blocks.0': model.down_blocks[0], 'down_blocks.8': model.down_blocks[8], 'down_attn.8': model.down_attn[8],
             'down_blocks.31': model.down_blocks[31], 'mid_block2': model.mid_block2, 'up_blocks.0': model.up_blocks[0],
             'up_blocks.35': model.up_blocks[35]}
    hooks = [m.register_forward_hook(lambda mod, inp, out, k=k: taps.__setitem__(k, out.detach().numpy().copy())) for k, m in names.items()]
    with torch.no_grad():
        score = mutils.get_score_fn(sde, model)(x, t, class_labels=labels)
    for h in hooks:
        h.remove()
    # taps are large (2 x 128 x 32 x 32 ...): keep a strided subsample [:, ::8 channels, ::4, ::4]
    save('forward_cifar.npz', x=x.numpy(), t=t.numpy(), labels=labels.numpy(), score=score.numpy(),
         n_params=np.int64(sum(p.numel() for p in model.parameters())),
         **{'tap.' + k: v[:, ::8, ::4, ::4].copy() for k, v in taps.items()})


def gen_init():
    """Reference init under torch.manual_seed(0): lets the build's parameter shell prove it consumes the torch RNG
    identically (same construction order)."""
    torch.manual_seed(0)
    model = mutils.create_model(make_cfg())
    sd = model.state_dict()
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode()); h.update(v.numpy().tobytes())
    save('init_seed0.npz', sha256=np.frombuffer(h.digest(), np.uint8), names=np.array(list(sd.keys())),
         **{'v.' + k: sd[k].numpy().reshape(-1)[:4].copy() for k in
            ['time_embed.W', 'time_mlp.2.weight', 'label_emb.bias', 'input_conv.weight', 'down_blocks.2.NIN_0.W',
             'down_attn.1.NIN_3.W', 'up_blocks.6.Dense_0.weight', 'upsample.1.Conv_0.weight', 'out_conv.weight']})


def gen_gto():
    """SURVEY 8f N1.  Benchmark/gto_halo_benchmarking.py cannot be imported (omegaconf is absent), and the un-normalisation is
    inline in generate_samples, so only its helper method _convert_to_spherical (:335-361) is run: the method's AST node is
    compiled from the reference file in memory (nothing is copied) and called with a stub `self` holding the two counters."""
    import ast
    path = '/root/reference/Benchmark/gto_halo_benchmarking.py'
    tree = ast.parse(open(path).read())
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == '_convert_to_spherical')
    ns = {'np': np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, 'exec'), ns)
    rng = np.random.RandomState(7)
    u3 = (rng.rand(64, 20, 3).astype(np.float32) * 2.6 - 1.3)
    u3[0, 0] = 0.0                                   # |u| = 0 branch
    u3[0, 1] = [1.0, 0.0, 0.0]; u3[0, 2] = [-1.0, 0.0, 0.0]; u3[0, 3] = [0.0, -1.0, 0.0]; u3[0, 4] = [0.0, 0.0, -1.0]
    u3[0, 5] = [0.6, -0.8, 0.0]; u3[0, 6] = [2.0, 2.0, 2.0]
    stub = NS(total_spherical_clips=0, total_spherical_elements=0)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        a, th, u = ns['_convert_to_spherical'](stub, u3[..., 0].copy(), u3[..., 1].copy(), u3[..., 2].copy())
    # The whole un-normalisation (affine columns included) is an inline block of generate_samples (:254-328) that cannot be
    # imported (omegaconf): its statements are compiled from the file's AST in memory -- the statements from
    # `samples = all_samples.reshape(...)` to `samples = np.column_stack(...)` -- and executed on a synthetic all_samples with
    # a stub `self` that carries the compiled _convert_to_spherical.  Nothing of the reference is written to disk but the arrays.
    gs = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == 'generate_samples')
    body = [st for st in gs.body if 254 <= st.lineno <= 328 and not (isinstance(st, ast.Expr) and isinstance(st.value, ast.Call)
                                                                       and getattr(st.value.func, 'id', '') == 'print')]
    assert isinstance(body[0], ast.Assign) and body[0].targets[0].id == 'samples' and body[-1].targets[0].id == 'samples', 'block moved'
    rng2 = np.random.RandomState(9)
    all_samples = (rng2.rand(48, 1, 9, 9).astype(np.float32) * 1.3 - 0.15)          # sampler output lives in [0,1]; go a bit outside
    stub2 = NS(total_spherical_clips=0, total_spherical_elements=0)
    stub2._convert_to_spherical = lambda ux, uy, uz: ns['_convert_to_spherical'](stub2, ux, uy, uz)
    env2 = {'np': np, 'self': stub2, 'all_samples': all_samples.copy()}
    with contextlib.redirect_stdout(io.StringIO()):
        exec(compile(ast.Module(body=body, type_ignores=[]), path, 'exec'), env2)
    assert env2['samples'].shape == (48, 67)
    save('gto_unnormalize.npz', ux=u3[..., 0], uy=u3[..., 1], uz=u3[..., 2], alpha=a, theta=th, u=u,
         clips=np.int64(stub.total_spherical_clips), full_in=all_samples, full_out=env2['samples'].astype(np.float64),
         full_clips=np.int64(stub2.total_spherical_clips))


def gen_gto_dataset():
    """SURVEY 8f N3.  RD/datasets.py imports torchvision (absent), so only the class GTOHaloImageDataset (:82-98) is compiled
    from the file's AST in memory; its __init__ (a pickle.load of a data file that is not here) is bypassed by setting the
    three attributes it would set."""
    import ast
    path = REF + '/datasets.py'
    tree = ast.parse(open(path).read())
    cls = next(n for n in ast.walk(tree) if isinstance(n, ast.ClassDef) and n.name == 'GTOHaloImageDataset')
    ns = {'np': np, 'torch': torch, 'Dataset': object}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), path, 'exec'), ns)
    ds = object.__new__(ns['GTOHaloImageDataset'])
    rng = np.random.RandomState(11)
    ds.data = rng.rand(37, 67).astype(np.float32)
    ds.mean, ds.std = 0.4652, 0.1811
    items = [ds[i] for i in range(len(ds))]
    save('gto_dataset.npz', data=ds.data, images=np.stack([im.numpy() for im, _ in items]), labels=np.stack([lb.numpy() for _, lb in items]))


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ['cifar']:
        gen_cifar()
        sys.exit(0)
    if sys.argv[1:] == ['ode']:
        gen_ode()
        sys.exit(0)
    if sys.argv[1:] == ['train_w0']:
        gen_train_w0()
        sys.exit(0)
    if sys.argv[1:] == ['gto']:
        gen_gto()
        gen_gto_dataset()
        sys.exit(0)
    gen_gto_dataset()
    gen_gto()
    gen_cube_sde()
    gen_forward()
    gen_sampler()
    gen_train()
    gen_train_w0()
    gen_ode()
    gen_cifar()
    gen_init()
