"""CPU parity of the REAL kernels + launch plans, executed by the execution-model emulator (tests/emu): the
unmodified csrc sources compiled by g++ with work-items as fibers and wave64 MFMA/shuffle semantics emulated.
Covers both execution plans (workgroup-resident fused U-Net and the layer-by-layer plan), the CFG adapter, the
cube helpers and the fused PC sampler, against reference-recorded fixtures and the oracle.  Sized for CPU."""
import os

import numpy as np
import pytest
import torch

from oracle import rd_oracle as O


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope='module')
def env(emu):
    import __graft_entry__ as ge
    model, cfg, params = ge.make_model('cpu')
    return dict(ge=ge, model=model, params=params)


def _model_with_path(ge, path, taps=False):
    os.environ['RDMI_PATH'] = path
    if taps:
        os.environ['RDMI_DEBUG_TAPS'] = '1'
    try:
        model, _, _ = ge.make_model('cpu')
        model.native_context(2, 9, 9, 'cpu')          # plan is chosen at context creation
    finally:
        os.environ.pop('RDMI_PATH', None)
        os.environ.pop('RDMI_DEBUG_TAPS', None)
    return model


def test_fused_plan_is_selected_and_fits_lds(env):
    ctx = env['model'].native_context(2, 9, 9, 'cpu')
    info = ctx.path_info()
    assert info.startswith('fused'), info
    ctx89 = env['model'].native_context(2, 8, 9, 'cpu')
    assert ctx89.path_info().startswith('fused')


@pytest.mark.parametrize('path', ['fused', 'layers'])
def test_forward_golden_both_plans(env, golden, path):
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    g = golden('forward_9x9.npz')
    model = _model_with_path(env['ge'], path)
    assert model._ctx[('cpu', 9, 9)].path_info().startswith(path)
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    idx = [0, 3, 7]
    with torch.no_grad():
        s = mutils.get_score_fn(sde, model)(T(g['x'][idx]), T(g['t'][idx]), class_labels=T(g['labels'][idx]))
    np.testing.assert_allclose(s.numpy(), g['score'][idx], rtol=0, atol=5e-5)


def test_multi_sample_programs(env, golden):
    """Large-batch programs (S = 2 / 4 samples per workgroup: the low-resolution half of the network runs for all S samples at
    once) against the reference's recorded forward.  RDMI_S_MIN_WG=1 lowers the batch threshold so that 2 samples select S=2
    and 5 samples select S=4 with a ragged last workgroup (one real sample, three clamped slots)."""
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    g = golden('forward_9x9.npz')
    os.environ['RDMI_S_MIN_WG'] = '1'
    try:
        model, _, _ = env['ge'].make_model('cpu')
        sde = sde_lib.RVESDE(0.01, 5, N=1000)
        for idx in ([6, 2], [0, 1, 2, 3, 7]):
            with torch.no_grad():
                s = mutils.get_score_fn(sde, model)(T(g['x'][idx]), T(g['t'][idx]), class_labels=T(g['labels'][idx]))
            np.testing.assert_allclose(s.numpy(), g['score'][idx], rtol=0, atol=5e-5)
        info = model._ctx[('cpu', 9, 9)].path_info()
        assert 'S=2 samples/workgroup from batch 2' in info and 'S=4 samples/workgroup from batch 4' in info, info
    finally:
        os.environ.pop('RDMI_S_MIN_WG', None)


def test_cooperative_program(env, golden):
    """The co-operative program on the emulator (four workgroups = four OS threads exchanging through the granule slots; consecutive
    workgroup ids form a group there): recorded forward, ragged batches (5 and 1 samples: clamped members recompute the last
    sample), the 8x9 shape, and agreement with the single-sample program (RDMI_COOP=0)."""
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    g = golden('forward_9x9.npz')
    g8 = golden('forward_8x9.npz')
    sde = sde_lib.RVESDE(0.01, 5, N=1000)

    def run(envvars):
        os.environ.update(envvars)
        try:
            model, _, _ = env['ge'].make_model('cpu')
            out = []
            for idx in ([0, 1, 2, 3, 4], [6], list(range(8))):
                with torch.no_grad():
                    out.append(mutils.get_score_fn(sde, model)(T(g['x'][idx]), T(g['t'][idx]), class_labels=T(g['labels'][idx])).numpy())
            with torch.no_grad():
                s89 = mutils.get_score_fn(sde, model)(T(g8['x'][:3]), T(g8['t'][:3]), class_labels=T(g8['labels'][:3])).numpy()
            ctx = model._ctx[('cpu', 9, 9)]
            assert not ctx.coop_gave_up()
            return out, s89, ctx.path_info()
        finally:
            for k in envvars:
                os.environ.pop(k, None)
    (a5, a1, a8), a89, info = run({})
    assert 'co-operative groups of 4 workgroups' in info, info
    (c5, c1, c8), c89, info_c = run({'RDMI_COOP': '0'})
    assert 'co-operative' not in info_c, info_c
    np.testing.assert_allclose(a8, g['score'], rtol=0, atol=5e-5)
    np.testing.assert_allclose(a5, g['score'][:5], rtol=0, atol=5e-5)
    np.testing.assert_allclose(a1, g['score'][[6]], rtol=0, atol=5e-5)
    np.testing.assert_allclose(a89, g8['score'][:3], rtol=0, atol=5e-5)
    np.testing.assert_allclose(a8, c8, rtol=0, atol=3e-5)
    assert np.array_equal(a5, a8[:5]) and np.array_equal(a1, a8[[6]])          # a sample does not depend on its group's composition


def test_layer_plan_intermediate_activations(env, golden):
    """Every recorded reference activation (21 taps) against the layer plan's tensors."""
    from rdmi import sde_lib
    g = golden('forward_9x9.npz')
    model = _model_with_path(env['ge'], 'layers', taps=True)
    os.environ['RDMI_DEBUG_TAPS'] = '1'
    try:
        model._ctx.clear()
        x, t, lab = T(g['x'][:2]), T(g['t'][:2]), T(g['labels'][:2])
        sigma = sde_lib.RVESDE(0.01, 5, N=1000).marginal_prob(x, t)[1]
        with torch.no_grad():
            model(x, sigma, lab)
    finally:
        os.environ.pop('RDMI_DEBUG_TAPS', None)
    np.testing.assert_allclose(model.get_tap('temb', x, 2).numpy().reshape(2, -1), g['temb'][:2], rtol=0, atol=5e-5)
    for k in g.files:
        if k.startswith('tap.'):
            a = model.get_tap(k[4:], x, 2).numpy()
            np.testing.assert_allclose(a, g[k], rtol=0, atol=5e-5, err_msg=k)


def test_8x9_and_cfg(env, golden):
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    model = env['model']
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    g8 = golden('forward_8x9.npz')
    with torch.no_grad():
        s = mutils.get_score_fn(sde, model)(T(g8['x'][:2]), T(g8['t'][:2]), class_labels=T(g8['labels'][:2]))
    np.testing.assert_allclose(s.numpy(), g8['score'][:2], rtol=0, atol=5e-5)
    g = golden('forward_9x9.npz')
    idx = [1, 6]
    with torch.no_grad():
        cf = mutils.get_cf_score_fn(sde, model, T(g['labels'][idx]), T(g['wt'][idx]))(T(g['x'][idx]), T(g['t'][idx]))
    np.testing.assert_allclose(cf.numpy(), g['cf_wt'][idx], rtol=0, atol=2e-4)


def test_cube_helpers(emu, golden):
    from rdmi import cube
    g = golden('cube_sde.npz')
    assert np.array_equal(cube.reflect(T(g['reflect_known_in'])).numpy(), g['reflect_known_out'])
    assert np.array_equal(cube.reflect(T(g['reflect_rand_in'])).numpy(), g['reflect_rand_out'])
    hk = cube.score_hk(T(g['hk_x']), T(g['hk_x0']), T(g['hk_sigma'])).numpy()
    np.testing.assert_allclose(hk, g['hk_score'], rtol=2e-5, atol=1e-4)
    assert bool(cube.inside(torch.rand(3, 1, 9, 9)).all())


def test_gto_dataset_batches(emu, golden):
    """device-resident table -> batches, bit-exact against items recorded from the reference dataset class."""
    from rdmi import datasets
    g = golden('gto_dataset.npz')
    ds = datasets.GTOHaloImageDataset(g['data'], 'cpu')
    assert len(ds) == 37
    img, lab = ds.batch(None)
    assert np.array_equal(img.numpy(), g['images']) and np.array_equal(lab.numpy(), g['labels'])
    idx = torch.tensor([36, 0, 5, 5, 17], dtype=torch.int64)
    img, lab = ds.batch(idx)
    assert np.array_equal(img.numpy(), g['images'][idx.numpy()]) and np.array_equal(lab.numpy(), g['labels'][idx.numpy()])
    im0, lb0 = ds[3]
    assert im0.shape == (1, 9, 9) and np.array_equal(im0.numpy(), g['images'][3]) and np.array_equal(lb0.numpy(), g['labels'][3])
    seen = torch.cat([lab for _, lab in ds.epoch(6, generator=torch.Generator().manual_seed(0), drop_last=False)])
    assert sorted(seen.view(-1).tolist()) == sorted(g['labels'].reshape(-1).tolist())          # one epoch = every item once
    with pytest.raises(ValueError):
        datasets.GTOHaloImageDataset(np.zeros((2, 82), np.float32), 'cpu')


def test_gto_unnormalize(emu, golden):
    """HIP un-normalisation kernel vs the oracle (spherical part pinned by reference-recorded vectors).
    Tolerance: angles from asinf/atan2f of this libm vs numpy's, 2e-6 rad; affine parts 1e-6 relative (fma contraction)."""
    from rdmi import harness
    g = golden('gto_unnormalize.npz')
    N = g['ux'].shape[0]
    ctrl = np.stack([g['ux'], g['uy'], g['uz']], -1).reshape(N, 60)
    s = np.random.RandomState(3).rand(N, 81).astype(np.float32) * 1.4 - 0.2
    s[:, 4:64] = ((ctrl + 1) / 2 - 0.4652) / 0.1811                    # so that the kernel sees (about) the recorded triplets
    out, clips = harness.unnormalize_gto(T(s).reshape(N, 1, 9, 9))
    ref, rclips = O.gto_unnormalize(s)
    big = np.abs(ref[:, 4:64].reshape(N, 20, 3)[..., :2] - out.numpy()[:, 4:64].reshape(N, 20, 3)[..., :2])
    assert np.all(np.minimum(big, np.abs(big - 2 * np.pi)) < 5e-5)      # angles (a wrap at 0 / 2pi is the same direction)
    np.testing.assert_allclose(out.numpy()[:, 6:64:3], ref[:, 6:64:3], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(out.numpy()[:, :4], ref[:, :4], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(out.numpy()[:, 64:], ref[:, 64:], rtol=2e-6, atol=1e-6)
    assert abs(int(clips.item()) - rclips) <= 2                          # |u| within an ulp of 1 may flip
    # all 67 columns against the reference's own inline block (fixture full_in/full_out, recorded by exec-ing its AST)
    fo = harness.unnormalize_gto(T(g['full_in']))[0].numpy()
    _check_gto_full(fo, g['full_out'])


def _check_gto_full(out, ref):
    """out [N,67] (HIP) vs the reference block's float64 output: affine columns 2e-6 relative (fp32 vs float64-of-fp32 inputs),
    angles 5e-5 rad (libm asinf/atan2f), modulo the 0 / 2pi wrap."""
    ang = np.zeros(67, bool)
    for t in range(20):
        ang[4 + 3 * t] = ang[5 + 3 * t] = True
    np.testing.assert_allclose(out[:, ~ang], ref[:, ~ang], rtol=2e-6, atol=1e-6)
    d = np.abs(out[:, ang] - ref[:, ang])
    assert np.all(np.minimum(d, np.abs(d - 2 * np.pi)) < 5e-5)


@pytest.mark.parametrize('corr', ['none', 'langevin'])
def test_fused_sampler_vs_oracle(env, corr):
    """3 reflected PC updates (N=4), B=2, CFG on, injected noise: the C loop (rdmi_pc_sample) against the oracle."""
    from rdmi import sampling, sde_lib
    model, params = env['model'], env['params']
    B, N = 2, 4
    g = torch.Generator().manual_seed(17)
    lab = torch.rand(B, 1, generator=g)
    per = 2 if corr == 'langevin' else 1
    noise = torch.randn((N - 1) * per, B, 81, generator=g)
    prior = torch.rand(B, 1, 9, 9, generator=g)
    trace = torch.zeros(N - 1, B, 81)
    sde = sde_lib.RVESDE(0.01, 5, N=N)
    fn = sampling.get_pc_sampler(sde, (B, 1, 9, 9), sampling.get_predictor('euler_maruyama'), sampling.get_corrector(corr),
                                 sampling.get_denoiser('none'), 0.01, 1, 1e-5, 'cpu', noise=noise, trace=trace)
    _rand = torch.rand
    torch.rand = lambda *a, **k: prior.clone()
    try:
        x, nfe = fn(model, weight=0.5, class_labels=lab)
    finally:
        torch.rand = _rand
    ref_trace = []
    xr, nfe_r = O.pc_sampler(params, O.RVESDE(0.01, 5, N=N), prior.numpy(), list(noise.reshape(-1, B, 1, 9, 9).numpy()),
                             lab.numpy(), 0.5, eps=1e-5, snr=0.01, n_steps=1, corrector=corr, trace=ref_trace)
    assert nfe == nfe_r == 8
    # first update is exact to fp32 amplification (g^2/N = 77 at t=1, N=4); later ones inherit that chaos
    np.testing.assert_allclose(trace[0].numpy().reshape(B, 1, 9, 9), ref_trace[0], rtol=0, atol=5e-3)
    assert float(x.min()) >= 0 and float(x.max()) <= 1
    assert np.median(np.abs(x.numpy() - xr)) < 5e-2


def test_param_rebinding_is_seen(env):
    """EMA copy_to / restore write through parameter storage: the next call must see the new values (repack per call)."""
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    from rdmi.models.ema import ExponentialMovingAverage
    model = env['ge'].make_model('cpu')[0]
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    x, t, lab = torch.rand(1, 1, 9, 9), torch.tensor([0.3]), torch.rand(1, 1)
    fn = mutils.get_score_fn(sde, model)
    with torch.no_grad():
        a = fn(x, t, class_labels=lab)
        ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
        for p in model.parameters():
            if p.requires_grad:
                p.mul_(1.01)
        b = fn(x, t, class_labels=lab)
        ema.store(model.parameters()); ema.copy_to(model.parameters())
        c = fn(x, t, class_labels=lab)
        ema.restore(model.parameters())
        d = fn(x, t, class_labels=lab)
    assert not torch.allclose(a, b) and torch.equal(a, c) and torch.equal(b, d)


def _eval_loss(ge, dev, golden_npz, model):
    from rdmi import losses, sde_lib
    from rdmi.models.ema import ExponentialMovingAverage
    g = golden_npz
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    state = dict(optimizer=None, model=model, ema=ExponentialMovingAverage(model.parameters(), 0.999), step=0, scaler=None)
    ev = losses.get_step_fn(sde, train=False, optimize_fn=None, reduce_mean=False, likelihood_weighting=False)
    tv, zv = torch.from_numpy(g['step0.t']).to(dev), torch.from_numpy(g['step0.z']).to(dev)
    _r, _n = torch.rand, torch.randn_like
    torch.rand = lambda *a, **k: ((tv - 1e-5) / (1 - 1e-5)).clone()
    torch.randn_like = lambda x, **k: zv.clone()
    try:
        return float(ev(state, torch.from_numpy(g['batch']).to(dev), class_labels=torch.from_numpy(g['labels']).to(dev)))
    finally:
        torch.rand, torch.randn_like = _r, _n


def test_eval_loss_step_matches_reference(env, golden):
    """losses.get_step_fn(train=False): perturb + reflect + score_hk target + weighted SSE (HIP kernels) with the
    reference's recorded (t, z): the reference's own loss value for the same weights (dropout and label drop off)."""
    g = golden('train_step.npz')
    loss = _eval_loss(env['ge'], 'cpu', g, env['model'])
    np.testing.assert_allclose(loss, float(g['step0.loss']), rtol=2e-5)


def _train_two_steps(ge, dev, g):
    """Two reference-shaped training steps (losses.get_step_fn(train=True)) with the recorded (t, z); dropout and label
    drop off (the fixture was recorded that way: the loss is then a pure function of t, z)."""
    from rdmi import losses, sde_lib
    from rdmi.models.ema import ExponentialMovingAverage
    model, cfg, _ = ge.make_model(dev)
    model.dropout, model.cond_drop_prob = 0.0, 0.0
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    optimizer = losses.get_optimizer(cfg, model.parameters())
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    state = dict(optimizer=optimizer, model=model, ema=ema, step=0, scaler=None)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=False,
                                 likelihood_weighting=False)
    batch, labels = torch.from_numpy(g['batch']).to(dev), torch.from_numpy(g['labels']).to(dev)
    out = {}
    _r, _n = torch.rand, torch.randn_like
    try:
        for k in range(2):
            tv, zv = torch.from_numpy(g[f'step{k}.t']).to(dev), torch.from_numpy(g[f'step{k}.z']).to(dev)
            torch.rand = lambda *a, tv=tv, **kw: ((tv - 1e-5) / (1 - 1e-5)).clone()
            torch.randn_like = lambda x, zv=zv, **kw: zv.clone()
            loss = step_fn(state, batch, class_labels=labels)
            out[f'loss{k}'] = float(loss.detach())
            if k == 0:
                out['grads'] = {n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters() if p.requires_grad}
    finally:
        torch.rand, torch.randn_like = _r, _n
    out['model'], out['ema'], out['state'] = model, ema, state
    return out


def check_train_against_reference(out, g, rtol_norm):
    names = list(g['param_names'])
    np.testing.assert_allclose(out['loss0'], float(g['step0.loss']), rtol=2e-5)
    np.testing.assert_allclose(out['loss1'], float(g['step1.loss']), rtol=2e-5)
    # the fixture holds gradients AFTER clip_grad_norm_(0.5) (RD/losses.py:39-40); ours are read after the same call
    gn = np.array([np.sqrt((out['grads'][n].astype(np.float64) ** 2).sum()) for n in names])
    ref = g['step0.grad_norms'].astype(np.float64)
    big = ref > 1e-7 * ref.max()             # NIN_1.b (key bias) gradients are analytically zero: softmax shift invariance
    assert big.sum() >= 250
    np.testing.assert_allclose(gn[big], ref[big], rtol=rtol_norm)
    assert gn[~big].max() < 1e-6 * ref.max()
    for n in ['out_conv.weight', 'time_mlp.0.bias', 'down_blocks.0.Conv_0.bias', 'up_attn.8.NIN_3.W', 'label_emb.weight']:
        r = g['step0.grad.' + n]
        np.testing.assert_allclose(out['grads'][n], r, rtol=0, atol=rtol_norm * np.abs(r).max())
    m = out['model']
    np.testing.assert_allclose(m.out_conv.bias.detach().cpu().numpy(), g['after2.out_conv.bias'], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(m.time_mlp[0].bias.detach().cpu().numpy(), g['after2.time_mlp.0.bias'], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(out['ema'].shadow_params[-1].cpu().numpy(), g['after2.ema.out_conv.bias'], rtol=1e-5, atol=1e-9)
    assert out['state']['step'] == 2 and out['ema'].num_updates == 2


def test_training_steps_match_reference(env, golden):
    """Loss, all 260 parameter gradients (norms) + 5 full gradient tensors, and the parameters / EMA after two Adam
    steps (warm-up, clipping) against the reference's recorded run."""
    g = golden('train_step.npz')
    out = _train_two_steps(env['ge'], 'cpu', g)
    check_train_against_reference(out, g, rtol_norm=2e-4)


def _train_steps_w0(ge, dev, g, nsteps):
    """Reference-shaped training steps with warmup=0 (full learning rate from step 0) and the recorded (t, z)."""
    from rdmi import losses, sde_lib
    from rdmi.models.ema import ExponentialMovingAverage
    model, cfg, _ = ge.make_model(dev)
    model.dropout, model.cond_drop_prob = 0.0, 0.0
    cfg.optim.warmup = 0
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    optimizer = losses.get_optimizer(cfg, model.parameters())
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    state = dict(optimizer=optimizer, model=model, ema=ema, step=0, scaler=None)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=False,
                                 likelihood_weighting=False)
    batch, labels = torch.from_numpy(g['batch']).to(dev), torch.from_numpy(g['labels']).to(dev)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    idx = {n: i for i, (n, _) in enumerate(named)}
    watch = [str(n) for n in g['watch']]
    out = {'p0': {n: dict(named)[n].detach().cpu().numpy().reshape(-1)[:256].copy() for n in watch}}
    _r, _n = torch.rand, torch.randn_like
    try:
        for k in range(nsteps):
            tv, zv = torch.from_numpy(g[f'step{k}.t']).to(dev), torch.from_numpy(g[f'step{k}.z']).to(dev)
            torch.rand = lambda *a, tv=tv, **kw: ((tv - 1e-5) / (1 - 1e-5)).clone()
            torch.randn_like = lambda x, zv=zv, **kw: zv.clone()
            out[f'loss{k}'] = float(step_fn(state, batch, class_labels=labels).detach())
            out[f'p{k + 1}'] = {n: dict(named)[n].detach().cpu().numpy().reshape(-1)[:256].copy() for n in watch}
            out[f'ema{k + 1}'] = {n: ema.shadow_params[idx[n]].cpu().numpy().reshape(-1)[:256].copy() for n in watch}
    finally:
        torch.rand, torch.randn_like = _r, _n
    out['state'], out['ema'] = state, ema
    return out


def check_train_w0_against_reference(out, g, nsteps, rtol_loss=2e-3):
    """Adam (bias-corrected moments, eps), clip_grad_norm_(0.5) and the EMA schedule against the reference's recorded run,
    compared as UPDATES (p_k - p_0): a no-op optimizer or EMA fails by 100 %.  Tolerance: step 1 moves every weight by
    lr * g/(|g| + 1e-8) = +-5e-4 exactly unless |g| ~ 1e-8; later steps divide by sqrt(v) of fp32 gradients that agree to
    ~1e-4 relative, so 1 % of the largest update per tensor (elements whose gradient changes sign move by < that).  An element
    whose gradient is within a few 1e-7 of zero has g / (|g| + 1e-8) of order one with an error of percents from ANY 1e-5
    difference in the forward's rounding (round 3: the training forward is the fused kernel, whose GroupNorm variance is
    single-pass): up to 1 % of a tensor's elements may miss the 1 % bound, none may miss 5 %."""
    for k in range(nsteps):
        np.testing.assert_allclose(out[f'loss{k}'], float(g[f'step{k}.loss']), rtol=2e-5 if k == 0 else rtol_loss)
    for n in [str(x) for x in g['watch']]:
        p0 = g['p0.' + n]
        np.testing.assert_array_equal(out['p0'][n], p0)
        for k in range(1, nsteps + 1):
            du, dr = out[f'p{k}'][n] - p0, g[f'p{k}.' + n] - p0
            assert np.abs(dr).max() > 1e-5, n                       # the reference really moved
            dev = np.abs(du - dr) / np.abs(dr).max()
            assert dev.max() <= 0.05 and (dev > 0.01).mean() <= 0.01, (n, k, float(dev.max()), float((dev > 0.01).mean()))
            eu, er = out[f'ema{k}'][n] - p0, g[f'ema{k}.' + n] - p0
            assert np.abs(er).max() > 1e-6, n
            dev = np.abs(eu - er) / np.abs(er).max()              # the shadow follows the parameter: same outlier rule
            assert dev.max() <= 0.05 and (dev > 0.01).mean() <= 0.01, ('EMA', n, k, float(dev.max()), float((dev > 0.01).mean()))
    assert out['state']['step'] == nsteps and out['ema'].num_updates == nsteps


def test_optimizer_and_ema_updates_match_reference(env, golden):
    """The optimizer / EMA half of losses.get_step_fn at full learning rate (fixture train_step_w0.npz, warmup=0): one step on
    the emulator -- clipping, Adam's bias-corrected first update and the EMA shadow all move at full size -- while the GPU test
    runs all three recorded steps (second moments, later bias corrections)."""
    g = golden('train_step_w0.npz')
    out = _train_steps_w0(env['ge'], 'cpu', g, 1)
    check_train_w0_against_reference(out, g, 1)


def test_fused_optimizer_step_equals_torch(emu):
    check_fused_optimizer_step_equals_torch('cpu')


def check_fused_optimizer_step_equals_torch(dev):
    """rdmi_opt_step (clip + Adam/AdamW + EMA in three launches) against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam(W)
    .step() + the reference-shaped EMA loop on the same tensors (ragged sizes, a tensor larger than one chunk, weight decay,
    several steps so the moments and bias corrections matter); state_dict round-trips with torch.optim.Adam."""
    from rdmi import losses
    from rdmi.models.ema import ExponentialMovingAverage
    from types import SimpleNamespace as NS
    for name, wd in (('Adam', 0.0), ('AdamW', 0.01), ('Adam', 0.02)):
        g = torch.Generator().manual_seed(5)
        shapes = [(3,), (17, 5), (4100,), (64, 9, 9), (1,)]
        pa = [torch.nn.Parameter(torch.randn(*s, generator=g).to(dev)) for s in shapes]
        pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
        cfg = NS(optim=NS(optimizer=name, lr=3e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=wd, warmup=4, grad_clip=0.7))
        oa = losses.get_optimizer(cfg, pa)
        assert isinstance(oa, losses._FusedStep) and isinstance(oa, torch.optim.Adam if name == 'Adam' else torch.optim.AdamW)
        ob = (torch.optim.Adam if name == 'Adam' else torch.optim.AdamW)(pb, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
        ea, eb = ExponentialMovingAverage(pa, 0.999), ExponentialMovingAverage(pb, 0.999)
        fn = losses.optimization_manager(cfg)
        for step in range(5):
            grads = [(torch.randn(*s, generator=g) * (3.0 if step % 2 else 0.01)).to(dev) for s in shapes]     # clipped and unclipped steps
            for p, q, gr in zip(pa, pb, grads):
                p.grad, q.grad = gr.clone(), gr.clone()
            assert fn(oa, pa, step=step, ema=ea) is True
            for grp in ob.param_groups:
                grp['lr'] = 3e-3 * min(step / 4, 1.0)
            torch.nn.utils.clip_grad_norm_(pb, max_norm=0.7)
            ob.step()
            eb.update(pb)
            for p, q in zip(pa, pb):
                np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.cpu().numpy(), rtol=2e-6, atol=0)        # the clipped gradient is left in .grad
                np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=3e-7)
            for s, t in zip(ea.shadow_params, eb.shadow_params):
                np.testing.assert_allclose(s.cpu().numpy(), t.cpu().numpy(), rtol=2e-6, atol=3e-7)
        assert ea.num_updates == eb.num_updates == 5
        sd = oa.state_dict()
        assert int(sd['state'][0]['step']) == 5 and set(sd['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'}
        import copy                                              # (load_state_dict keeps same-dtype tensors by reference: copy, as a checkpoint file would)
        ob.load_state_dict(copy.deepcopy(sd))                    # a torch.optim.Adam accepts the fused optimizer's checkpoint
        oa.load_state_dict(copy.deepcopy(ob.state_dict()))       # and the other way round; the next fused step re-binds the pointers
        for p, q in zip(pa, pb):
            p.grad, q.grad = torch.ones_like(p), torch.ones_like(q)
        fn(oa, pa, step=5, ema=None)
        torch.nn.utils.clip_grad_norm_(pb, max_norm=0.7); ob.step()
        for p, q in zip(pa, pb):
            np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=3e-7)
        # a non-finite gradient poisons the WHOLE step, as torch's clamp of the clip coefficient does: not only the
        # NaN element's own parameter
        for p, q in zip(pa, pb):
            p.grad, q.grad = torch.ones_like(p), torch.ones_like(q)
        pa[1].grad[0, 0] = float('nan'); pb[1].grad[0, 0] = float('nan')
        fn(oa, pa, step=6, ema=None)
        torch.nn.utils.clip_grad_norm_(pb, max_norm=0.7); ob.step()
        for p, q in zip(pa, pb):
            assert torch.isnan(q).all() and torch.isnan(p).all()
    # a step that leaves a shadowed parameter without gradient does NOT claim the EMA (the caller's ema.update relaxes every shadow),
    # and an optimize_fn without the `ema` keyword (the reference's signature) is called exactly once
    pa = [torch.nn.Parameter(torch.randn(5).to(dev)), torch.nn.Parameter(torch.randn(7).to(dev))]
    cfg = NS(optim=NS(optimizer='Adam', lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, warmup=0, grad_clip=-1.0))
    oa = losses.get_optimizer(cfg, pa)
    ea = ExponentialMovingAverage(pa, 0.5)
    pa[0].grad = torch.ones_like(pa[0])                        # pa[1] has no gradient this step
    assert losses.optimization_manager(cfg)(oa, pa, step=0, ema=ea) is False
    calls = []

    def ref_style(optimizer, params, step, lr=1e-2, warmup=0, grad_clip=-1.0, scaler=None):
        calls.append(step)
        raise TypeError('raised inside optimize_fn')
    from rdmi import sde_lib
    stepper = losses.get_step_fn(sde_lib.RVESDE(0.01, 5, N=10), train=True, optimize_fn=ref_style)
    import inspect
    assert 'takes_ema' in inspect.getclosurevars(stepper).nonlocals and inspect.getclosurevars(stepper).nonlocals['takes_ema'] is False


def test_bf16_training_step_within_bf16_tolerance(env, golden):
    """train_dtype='bf16' (BASELINE config #4): bf16 weight copies and bf16 MFMA operands in the forward, data-gradient and
    weight-gradient contractions, fp32 accumulate / master weights / optimizer.  One reference-shaped step against the fp32
    reference fixture.  Stated tolerance: loss 1 %, every gradient norm 4 % (bf16 keeps 8 significant bits: 2^-9 per operand
    through ~45 convolutions each way; measured on the GPU: loss 0.26 %, norms 1.1 % max / 0.3 % median); the fp32 step
    (train_dtype='f32') keeps its own tolerance (tests above)."""
    g = golden('train_step.npz')
    ge = env['ge']
    _mk = ge.make_model

    def mk(device, **kw):
        m, cfg, p = _mk(device, **kw); m.train_dtype = 'bf16'; return m, cfg, p
    ge.make_model = mk
    try:
        out = _train_steps_generic(ge, 'cpu', g, 1)
    finally:
        ge.make_model = _mk
    check_bf16_train(out, g)


def _train_steps_generic(ge, dev, g, nsteps):
    from rdmi import losses, sde_lib
    from rdmi.models.ema import ExponentialMovingAverage
    model, cfg, _ = ge.make_model(dev)
    model.dropout, model.cond_drop_prob = 0.0, 0.0
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    optimizer = losses.get_optimizer(cfg, model.parameters())
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    state = dict(optimizer=optimizer, model=model, ema=ema, step=0, scaler=None)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=False, likelihood_weighting=False)
    batch, labels = torch.from_numpy(g['batch']).to(dev), torch.from_numpy(g['labels']).to(dev)
    out = {}
    _r, _n = torch.rand, torch.randn_like
    try:
        for k in range(nsteps):
            tv, zv = torch.from_numpy(g[f'step{k}.t']).to(dev), torch.from_numpy(g[f'step{k}.z']).to(dev)
            torch.rand = lambda *a, tv=tv, **kw: ((tv - 1e-5) / (1 - 1e-5)).clone()
            torch.randn_like = lambda x, zv=zv, **kw: zv.clone()
            out[f'loss{k}'] = float(step_fn(state, batch, class_labels=labels).detach())
            if k == 0:
                out['grads'] = {n: p.grad.detach().cpu().numpy().copy() for n, p in model.named_parameters() if p.requires_grad}
    finally:
        torch.rand, torch.randn_like = _r, _n
    tctx = model._ctx[('train', str(torch.device(dev)), 9, 9)]
    out['dtype'] = tctx.train_dtype
    out['path'] = tctx.path_info()
    assert ('training forward: fused program' in out['path']) == (os.environ.get('RDMI_TRAIN_FUSED', '1') != '0'), out['path']
    return out


def test_training_forward_fused_equals_layer_plan(env, golden):
    """The training forward as one workgroup-resident launch (stash of every layer output + Dropout_0 in the GroupNorm_1 epilogues)
    against the layer plan's forward (RDMI_TRAIN_FUSED=0), dropout ON with the same step seed: same loss, same gradients (the
    backward reads the stashed activations and regenerates the same masks)."""
    from rdmi import losses, sde_lib
    g = golden('train_step.npz')
    batch, labels = torch.from_numpy(g['batch'][:4]), torch.from_numpy(g['labels'][:4])

    def run(envvars):
        os.environ.update(envvars)
        try:
            model, cfg, _ = env['ge'].make_model('cpu')
            model.train(); model.cond_drop_prob = 0.0
            loss_fn = losses.get_sde_loss_fn(sde_lib.RVESDE(0.01, 5, N=1000), train=True, reduce_mean=False, likelihood_weighting=False)
            torch.manual_seed(21)
            loss = loss_fn(model, batch, class_labels=labels)
            loss.backward()
            info = model._ctx[('train', 'cpu', 9, 9)].path_info()
            return float(loss.detach()), {n: p.grad.detach().numpy().copy() for n, p in model.named_parameters() if p.requires_grad}, info
        finally:
            for k in envvars:
                os.environ.pop(k, None)
    la, ga, ia = run({})
    lb, gb, ib = run({'RDMI_TRAIN_FUSED': '0'})
    assert 'training forward: fused program' in ia and 'training forward' not in ib
    assert abs(la / lb - 1) < 2e-5, (la, lb)
    for n in ga:
        ref = np.abs(gb[n]).max()
        assert np.abs(ga[n] - gb[n]).max() <= 2e-4 * ref + 1e-7, n


def check_bf16_train(out, g):
    assert out['dtype'] == 'bf16'
    names = list(g['param_names'])
    assert abs(out['loss0'] / float(g['step0.loss']) - 1) < 1e-2
    gn = np.array([np.sqrt((out['grads'][n].astype(np.float64) ** 2).sum()) for n in names])
    ref = g['step0.grad_norms'].astype(np.float64)
    big = ref > 1e-5 * ref.max()
    assert big.sum() >= 240
    # not a silent fp32 run: the gradients carry bf16 rounding (the loss itself may equal the fp32 one: at batches of one wave of
    # workgroups the forward contraction runs on the fused exact-fp32 kernel, bf16 enters with the stashed activations and the backward)
    assert np.abs(gn[big] / ref[big] - 1).max() > 1e-4
    assert np.abs(gn[big] / ref[big] - 1).max() < 4e-2, float(np.abs(gn[big] / ref[big] - 1).max())
    assert np.median(np.abs(gn[big] / ref[big] - 1)) < 1e-2


def _small_rgb_model(ge, compute_dtype):
    """A 16x16 RGB NCSN++ (nf=64, ch_mult [1,2,2], one block per level, attention at 8x8 with C=128, scale_by_sigma: the bf16 plan's copy-staged
    3x3 and 1x1 convs and fused attention core are reached, and the 192-channel concat has 6-channel GroupNorm groups): small enough for the
    emulator, and -- more than one image channel -- planned by the spatially TILED plan like BASELINE config #5."""
    from oracle.weights import make_params
    from rdmi.models import utils as mutils
    cfg = ge.demo_config(image_size=16, image_width=16)
    m = cfg.model
    m.nf, m.ch_mult, m.num_res_blocks, m.attn_resolutions = 64, [1, 2, 2], 1, [8]
    m.channels, m.scale_by_sigma, m.compute_dtype = 3, True, compute_dtype
    cfg.sde.sigma_max = 50
    params = make_params(3, nf=64, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(8,), image_size=16, channels=3)
    model = mutils.create_model(cfg)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()}, strict=True)
    return model.eval(), params


@pytest.mark.parametrize('dtype,tol,iconv', [('f32', 2e-5, None), ('bf16', 3e-2, '0'), ('bf16', 3e-2, '1'), ('bf16', 3e-2, 'wide')])
def test_tiled_plan_small_rgb_model(emu, dtype, tol, iconv, monkeypatch):
    """The tiled plan (csrc/tiled_kernels.h: tconv, channel-sum GroupNorm statistics, batched-GEMM attention, and for bf16 the
    pre-activated gn_act + tconv_pre pair) on the emulator against the torch oracle: classifier-free-guidance score of two samples.
    Tolerance relative to each sample's largest |score|: fp32 2e-5; bf16 operands 3e-2 (stated bf16 tolerance, as on the GPU).
    iconv '1': the implicit-GEMM conv (iconv_kernel: LDS-DMA staged 128 x 128 tiles) is forced at this small batch (the 8x8 level's
    128-column convs and the attention projections take it; four forwards = two row tiles), '0': switched off."""
    import __graft_entry__ as ge
    if iconv == 'wide':         # the workgroup widths of a sampling batch (NCT = 2 / 4 column tiles per wave): the epilogue's vector form
        monkeypatch.setenv('RDMI_TILED_MIN_WGS', '1')
    elif iconv == '1':
        monkeypatch.setenv('RDMI_ICONV_MIN_WGS', '1')
    elif iconv == '0':
        monkeypatch.setenv('RDMI_ICONV', '0')
    from oracle import rd_oracle_torch as OT
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    model, params = _small_rgb_model(ge, dtype)
    pt = {k: torch.from_numpy(v) for k, v in params.items()}
    sde = sde_lib.RVESDE(0.01, 50, N=1000)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(2, 3, 16, 16, generator=g); lab = torch.zeros(2, 1); w = torch.tensor([0.0, 0.6]); t = torch.tensor([0.7, 0.2])
    with torch.no_grad():
        s = mutils.get_cf_score_fn(sde, model, lab, w)(x, t)
        ref = OT.cf_score(pt, x, t, lab, w, smax=50.0, ch_mult=(1, 2, 2), nrb=1, attn_levels=(False, True, False), scale_by_sigma=True)
    info = model._ctx[('cpu', 16, 16)].path_info()
    assert info.startswith('tiled'), info
    for n in range(2):
        assert float((s[n] - ref[n]).abs().max()) <= tol * float(ref[n].abs().max()), (n, dtype)
    if iconv in ('0', '1'):         # the forced run really took the implicit-GEMM kernel (and the other one did not)
        ctx = model._ctx[('cpu', 16, 16)]
        ctx.set_profiling(True)
        with torch.no_grad():
            mutils.get_cf_score_fn(sde, model, lab, w)(x, t)
        names = {p['kernel']: p['launches'] for p in ctx.get_profile()}
        assert (names.get('iconv_kernel<bf16>', 0) >= 4) == (iconv == '1'), names


def test_tiled_plan_groupnorm_statistics_with_large_group_means(emu):
    """Round-2 advisor finding: GroupNorm statistics formed from the producing convs' per-tile channel records must survive groups
    whose mean is large against their spread (trained checkpoints; the synthetic weights above have near-zero means).  Every conv bias
    of the small RGB model is shifted by +40 (conv outputs have unit-order spread: |mean| / std ~ 40, so E[x^2] - mean^2 in fp32 would
    lose ~3 digits of the variance), and the tiled plan (per-tile sums + squared deviations about the tile mean, merged with Chan's
    formula) is compared with the torch oracle (F.group_norm) and with the exact two-pass statistics path (RDMI_TILED_STATS_PASS=1)."""
    import __graft_entry__ as ge
    from oracle import rd_oracle_torch as OT
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    sde = sde_lib.RVESDE(0.01, 50, N=1000)
    g = torch.Generator().manual_seed(12)
    x = torch.rand(2, 3, 16, 16, generator=g); lab = torch.zeros(2, 1); t = torch.tensor([0.6, 0.3])

    def run(envvars):
        os.environ.update(envvars)
        try:
            model, params = _small_rgb_model(ge, 'f32')
            sd = model.state_dict()
            shifted = {}
            for k, v in sd.items():
                if k.endswith('Conv_0.bias') or k.endswith('Conv_1.bias') or k == 'input_conv.bias':
                    shifted[k] = v + 40.0
            model.load_state_dict({**sd, **shifted})
            with torch.no_grad():
                s = mutils.get_score_fn(sde, model)(x, t, class_labels=lab)
            return s, {k: v.clone() for k, v in model.state_dict().items()}
        finally:
            for k in envvars:
                os.environ.pop(k, None)
    s, pt = run({})
    s2, _ = run({'RDMI_TILED_STATS_PASS': '1'})
    with torch.no_grad():
        ref = OT.ncsnpp_forward(pt, x, OT.sigma_of(t, smax=50.0), lab, ch_mult=(1, 2, 2), nrb=1, attn_levels=(False, True, False), scale_by_sigma=True)
    for n in range(2):
        amp = float(ref[n].abs().max())
        assert float((s[n] - ref[n]).abs().max()) <= 1e-4 * amp, (n, float((s[n] - ref[n]).abs().max()) / amp)
        assert float((s[n] - s2[n]).abs().max()) <= 1e-4 * amp, n
