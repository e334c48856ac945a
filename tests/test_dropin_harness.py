"""Row H of SURVEY 8a and the advertised drop-in route: the flat modules under
optimized-diffusion-model_amd/dropin (`import sampling, sde_lib, losses, cube; from models import utils`) and
rdmi.harness.generate_samples -- the counterpart of GTOHaloBenchmarker.generate_samples
(Benchmark/gto_halo_benchmarking.py:212-257) and of the run_train snapshot block (RD/run_train.py:272-282) -- must
produce the same (x, nfe) as the direct rdmi API: EMA store / copy_to / restore around the call, labels, the
(N,1,9,9) -> (N,81)[:, :67] flattening.  Also covers the two surfaces that had no test: sampling.get_ode_sampler
(RD/sampling.py:342-392) against the oracle's probability-flow drift, and cube.sample_hk (RD/cube.py:52-70).

The same checks run on the CPU emulator build (small sizes) and, marked gpu, through librdmi.so on the MI355X.
"""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROPIN = os.path.join(ROOT, 'optimized-diffusion-model_amd', 'dropin')
_FLAT = ('sampling', 'sde_lib', 'losses', 'cube', 'models', 'models.utils', 'models.ema', 'models.ncsnpp')


class _DropinPath:
    """`sys.path.insert(0, dropin)` the way INTEGRATION.md tells a reference user to, undone afterwards (the flat
    names are generic; they must not leak into the other test modules)."""

    def __enter__(self):
        self.saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k in _FLAT}
        sys.path.insert(0, DROPIN)
        importlib.invalidate_caches()
        return self

    def __exit__(self, *a):
        sys.path.remove(DROPIN)
        for k in list(sys.modules):
            if k in _FLAT:
                del sys.modules[k]
        sys.modules.update(self.saved)


def _check_dropin_and_harness(ge, dev, B, N, corrector):
    from rdmi import harness
    from rdmi import sampling as r_sampling, sde_lib as r_sde_lib, losses as r_losses, cube as r_cube
    from rdmi.models import utils as r_mutils, ema as r_ema, ncsnpp as r_ncsnpp
    with _DropinPath():
        import sampling, sde_lib, losses, cube                      # noqa: E401  (the reference's flat module names)
        from models import utils as mutils
        from models import ncsnpp
        from models.ema import ExponentialMovingAverage
        # same objects, not copies: registries must be shared so a user's @register_predictor is seen by get_sampling_fn
        assert sampling.get_sampling_fn is r_sampling.get_sampling_fn and sampling._PREDICTORS is r_sampling._PREDICTORS
        assert sde_lib.RVESDE is r_sde_lib.RVESDE and losses.get_step_fn is r_losses.get_step_fn
        assert cube.reflect is r_cube.reflect and mutils.get_score_fn is r_mutils.get_score_fn
        assert ncsnpp.NCSNpp is r_ncsnpp.NCSNpp and ExponentialMovingAverage is r_ema.ExponentialMovingAverage
        assert mutils.get_model('ncsnpp') is ncsnpp.NCSNpp
        with pytest.raises(ValueError):
            mutils.register_model(ncsnpp.NCSNpp, name='ncsnpp')

        model, cfg, _ = ge.make_model(dev, num_scales=N, corrector=corrector)
        ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
        with torch.no_grad():                                         # make the live weights differ from the EMA copy
            for p in model.parameters():
                if p.requires_grad:
                    p.mul_(1.03)
        live = [p.detach().clone() for p in model.parameters()]
        g = torch.Generator().manual_seed(77)
        labels = torch.rand(B, 1, generator=g)

        # ---- the harness (EMA swap, labels, flatten) ...
        torch.manual_seed(5)
        out, times = harness.generate_samples(model, ema, cfg, num_samples=B, batch_size=B, device=dev, guidance_weight=0.25,
                                              labels=labels)
        assert out.shape == (B, 67) and out.device.type == 'cpu' and len(times) == 1
        for p, q in zip(model.parameters(), live):                    # restore() put the live weights back
            assert torch.equal(p.detach(), q)

        # ---- ... equals the reference-shaped call sequence through the flat modules (Benchmark :212-257)
        sde = sde_lib.RVESDE(sigma_min=cfg.sde.sigma_min, sigma_max=cfg.sde.sigma_max, N=cfg.sde.num_scales)
        sampling_fn = sampling.get_sampling_fn(cfg, sde, (B, 1, 9, 9), 1e-5, dev)
        torch.manual_seed(5)
        ema.store(model.parameters())
        ema.copy_to(model.parameters())
        sample, nfe = sampling_fn(model, weight=0.25, class_labels=labels.to(dev))
        ema.restore(model.parameters())
        assert nfe == N * 2 and sample.shape == (B, 1, 9, 9)
        flat = sample.cpu().reshape(B, -1)[:, :67]
        assert torch.equal(out, flat)
        assert float(out.min()) >= 0 and float(out.max()) <= 1

        # ---- and the EMA swap matters: the live weights give a different sample
        torch.manual_seed(5)
        other, _ = sampling_fn(model, weight=0.25, class_labels=labels.to(dev))
        assert not torch.equal(other.cpu().reshape(B, -1)[:, :67], out)

        # ---- run_train's snapshot variant: zero labels, weight None, several batches with a ragged tail
        torch.manual_seed(6)
        out2, times2 = harness.generate_samples(model, ema, cfg, num_samples=B + 1, batch_size=B, device=dev,
                                                guidance_weight=None, labels='zeros')
        assert out2.shape == (B + 1, 67) and len(times2) == 2 and torch.isfinite(out2).all()
    assert 'sampling' not in sys.modules or getattr(sys.modules['sampling'], '__file__', '').find('dropin') < 0


def _check_ode_sampler(ge, dev, params, B, span):
    """get_ode_sampler integrates the probability-flow drift x bump mollifier with scipy RK45 (RD/sampling.py:342-392);
    the right-hand side is a HIP score call.  Checked here: (1) the RHS at the prior against the oracle's
    -1/2 g(t)^2 cf_score(x, t) * bump(x); (2) a short integration T -> T - span against the same integrator driven by the
    oracle RHS (tolerance: rtol/atol 1e-5 of both integrations plus the score tolerance amplified by g^2 * span)."""
    from scipy import integrate
    from oracle import rd_oracle as O
    from rdmi import sampling, sde_lib
    from rdmi.models import utils as mutils
    model, cfg, _ = ge.make_model(dev)
    # sigma_max = 0.5: g(1)^2 = 2 (5 would give 310 and a drift Lipschitz constant that amplifies the 1e-4 score tolerance
    # chaotically within a few 1e-3 of time); the network is conditioned on log(sigma) either way
    sde = sde_lib.RVESDE(0.01, 0.5, N=1000)
    osde = O.RVESDE(0.01, 0.5, N=1000)
    g = torch.Generator().manual_seed(21)
    z = (1 - 2e-2) * torch.rand(B, 1, 9, 9, generator=g) + 1e-2
    lab = torch.rand(B, 1, generator=g)
    moll = 200

    def bump(v):
        return np.exp((-1 / (0.5 ** 2 - (0.5 - v) ** 2) + 4) / moll)

    def rhs_oracle(t, xv):
        x = xv.reshape(B, 1, 9, 9).astype(np.float32)
        tt = np.full((B,), t, np.float32)
        s = O.cf_score_fn(params, osde, x, tt, lab.numpy(), 0.5)
        return (-0.5 * (osde.g(tt) ** 2)[:, None, None, None] * s * bump(x)).reshape(-1).astype(np.float64)

    # (1) right-hand side
    score_fn = mutils.get_cf_score_fn(sde, model, lab.to(dev), 0.5)
    with torch.no_grad():
        tv = torch.full((B,), 0.8, device=dev)
        drift = sde.reverse(score_fn, probability_flow=True).sde(z.to(dev), tv)[0]
    ours = (drift.cpu().numpy() * bump(z.numpy())).reshape(-1)
    ref = rhs_oracle(0.8, z.numpy().reshape(-1))
    np.testing.assert_allclose(ours, ref, rtol=0, atol=max(1e-5, 2e-4 * 0.5 * float(osde.g(np.array([0.8], np.float32))[0]) ** 2 * 2))
    if span <= 0:
        return
    # (2) short integration through the sampler closure
    fn = sampling.get_ode_sampler(sde, (B, 1, 9, 9), eps=1.0 - span, moll=moll, side_eps=1e-2, device=dev)
    x, nfev = fn(model, z=z.to(dev), weight=0.5, class_labels=lab.to(dev))
    sol = integrate.solve_ivp(rhs_oracle, (1.0, 1.0 - span), z.numpy().reshape(-1).astype(np.float64), rtol=1e-5, atol=1e-5, method='RK45')
    assert nfev >= 8 and x.shape == (B, 1, 9, 9)
    np.testing.assert_allclose(x.cpu().numpy().reshape(-1), sol.y[:, -1], rtol=0, atol=5e-3)
    # dispatch through the config switch (method: ode) builds the same closure
    cfg.sampling.method = 'ode'; cfg.sampling.moll = moll; cfg.sampling.side_eps = 1e-2
    assert callable(sampling.get_sampling_fn(cfg, sde, (B, 1, 9, 9), 1e-3, dev))


def _check_ode_device_vs_reference(ge, dev, g, tag, ks, full):
    """The on-device RK45 (rdmi_ode_sample) against the REFERENCE's recorded scipy run (fixture ode_rk45.npz, made by
    oracle/gen_golden.py from RD/sampling.py:342-392 on an injected prior; rtol = atol = 1e-5 as the reference sets them).
    One-step parity: restart from the recorded float64 state y_k at t_k with first_step = |t_k+1 - t_k| and max_steps = 1: the
    controller must accept that very step (t reached == t_k+1) and land on y_k+1.  Tolerance 5e-6: the state moves by
    h * 1/2 g^2 |score| <= 0.01 per step, of which the 2e-4 score tolerance is 2e-6; the start state is passed as fp32.
    Full run (GPU): nfev within two step attempts of the reference's and the final sample within 2e-3 -- a free run, so NOT the
    contract: per-step differences of ~1e-6 are amplified by the flow (Lipschitz constant 1/2 g^2 |ds/dx| integrated over t) and a
    borderline accept/reject flips the step sequence; measured 7.9e-4 at the worst of 324 elements."""
    from rdmi import sampling, sde_lib
    model, cfg, _ = ge.make_model(dev)
    sde = sde_lib.RVESDE(0.01, 0.5, N=1000)
    lab = torch.from_numpy(g[f'{tag}.labels']).to(dev)
    B = lab.shape[0]
    ts = g[f'{tag}.t']
    for k in ks:
        yk = torch.from_numpy(g[f'{tag}.y{k}'].astype(np.float32)).reshape(B, 1, 9, 9).to(dev)
        class _From(sde_lib.RVESDE):                           # integrate from t_k (the sampler starts at sde.T)
            T = property(lambda self, _t=float(ts[k]): _t)
        sub = _From(0.01, 0.5, N=1000)
        fn = sampling.get_ode_sampler(sub, (B, 1, 9, 9), eps=float(ts[-1]), moll=200, side_eps=1e-2, device=dev,
                                      first_step=abs(float(ts[k + 1] - ts[k])), max_steps=1)
        x, nfev = fn(model, z=yk, weight=0.5, class_labels=lab)
        assert nfev == 7                                      # f(t_k, y_k) + 6 stages, no rejected attempt
        assert abs(fn.last['t'] - float(ts[k + 1])) < 1e-12, (k, fn.last, ts[k + 1])
        np.testing.assert_allclose(x.cpu().numpy().reshape(-1), g[f'{tag}.y{k + 1}'], rtol=0, atol=5e-6, err_msg=f'step {k}')
        if k + 2 < len(ts) - 1:       # the controller's proposal bounds the step the reference accepted next (equal unless the
            #                           reference first tried it and rejected, or this step itself followed a rejection: factor <= 1)
            assert abs(float(ts[k + 2] - ts[k + 1])) <= fn.last['h_next'] * (1 + 2e-2)
    if full:
        fn = sampling.get_ode_sampler(sde, (B, 1, 9, 9), eps=float(ts[-1]), moll=200, side_eps=1e-2, device=dev)
        x, nfev = fn(model, z=torch.from_numpy(g[f'{tag}.z']).to(dev), weight=0.5, class_labels=lab)
        assert abs(nfev - int(g[f'{tag}.nfev'])) <= 12, (nfev, int(g[f'{tag}.nfev']))
        np.testing.assert_allclose(x.cpu().numpy(), g[f'{tag}.x'], rtol=0, atol=2e-3)
        # the generic scipy route over the HIP score function (fused=False) is the same sampler
        fn2 = sampling.get_ode_sampler(sde, (B, 1, 9, 9), eps=float(ts[-1]), moll=200, side_eps=1e-2, device=dev, fused=False)
        x2, nfev2 = fn2(model, z=torch.from_numpy(g[f'{tag}.z']).to(dev), weight=0.5, class_labels=lab)
        assert abs(nfev2 - nfev) <= 12
        np.testing.assert_allclose(x2.cpu().numpy(), x.cpu().numpy(), rtol=0, atol=2e-3)


def _check_sample_hk(dev):
    """cube.sample_hk (RD/cube.py:52-70) = reflect(x + sigma z): inside the cube, exact against reflect() of the same draw,
    and for a large sigma the reflected heat kernel is (nearly) uniform: mean 1/2, variance 1/12."""
    from rdmi import cube
    x = torch.rand(64, 1, 9, 9, device=dev)
    torch.manual_seed(3)
    a = cube.sample_hk(x, 0.3)
    torch.manual_seed(3)
    zz = torch.randn_like(x)
    assert torch.equal(a, cube.reflect(zz * 0.3 + x))
    assert bool(cube.inside(a).all())
    sig = torch.full((64,), 4.0, device=dev)
    b = cube.sample_hk(x, sig)
    assert abs(float(b.mean()) - 0.5) < 0.02 and abs(float(b.var()) - 1 / 12) < 0.01
    small = cube.sample_hk(x, 1e-4)
    assert float((small - x).abs().max()) < 1e-3
    assert not bool(cube.inside(x + 1.5).any())


# ---- CPU: emulator build of the same kernels (one emulated sample-forward costs ~4 s: sizes are minimal) ---------------
def test_dropin_and_harness_emulator(emu):
    import __graft_entry__ as ge
    _check_dropin_and_harness(ge, 'cpu', B=1, N=2, corrector='none')


def test_ode_device_rk45_one_step_emulator(emu, golden):
    """One adaptive RK45 step of the on-device ODE sampler (7 right-hand sides at B=1, CFG) against the reference's recorded
    scipy step; the full integration runs on the GPU (test_ode_device_rk45_gpu)."""
    import __graft_entry__ as ge
    _check_ode_device_vs_reference(ge, 'cpu', golden('ode_rk45.npz'), 'b1', [1], full=False)


def test_sample_hk_emulator(emu):
    _check_sample_hk('cpu')


def test_ode_sampler_host_logic():
    """get_ode_sampler's host side (RD/sampling.py:342-392) with a user-registered analytic score model (the generic
    route: any callable model keeps working): prior scaling by side_eps, bump mollifier, float64 scipy state <-> float32
    device tensor round trip, nfev, and agreement with an independent scipy integration of the same closed-form drift."""
    from scipy import integrate
    from rdmi import sampling, sde_lib

    class Analytic(torch.nn.Module):                   # score(x, sigma) = (0.5 - x) / (1 + sigma^2): pulls towards the centre
        def forward(self, x, time_cond, class_labels=None):
            return (0.5 - x) / (1 + time_cond[:, None, None, None] ** 2)

    sde = sde_lib.RVESDE(0.01, 2.0, N=1000)
    shape = (3, 1, 4, 5)
    fn = sampling.get_ode_sampler(sde, shape, eps=1e-3, moll=200, side_eps=1e-2, device='cpu')
    torch.manual_seed(0)
    x, nfev = fn(Analytic())
    torch.manual_seed(0)
    z = (1 - 2e-2) * torch.rand(shape) + 1e-2
    gc = np.sqrt(2 * (np.log(2.0) - np.log(0.01)))

    def rhs(t, xv):
        sig = 0.01 * (2.0 / 0.01) ** t
        score = (0.5 - xv) / (1 + sig ** 2)
        bump = np.exp((-1 / (0.25 - (0.5 - xv) ** 2) + 4) / 200)
        return -0.5 * (sig * gc) ** 2 * score * bump
    sol = integrate.solve_ivp(rhs, (1.0, 1e-3), z.numpy().reshape(-1).astype(np.float64), rtol=1e-5, atol=1e-5, method='RK45')
    assert x.shape == shape and x.dtype == torch.float32 and nfev >= 20
    np.testing.assert_allclose(x.numpy().reshape(-1), sol.y[:, -1], rtol=0, atol=2e-4)
    assert float(x.min()) > 0 and float(x.max()) < 1
    # class-conditional route: get_cf_score_fn doubles the batch for any model
    x2, _ = fn(Analytic(), z=z.clone(), weight=0.5, class_labels=torch.rand(3, 1))
    np.testing.assert_allclose(x2.numpy(), x.numpy(), rtol=0, atol=2e-4)     # this score ignores labels: (1+w)s - ws = s


# ---- GPU -----------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('corrector', ['none', 'langevin'])
def test_dropin_and_harness_gpu(corrector):
    import __graft_entry__ as ge
    ge.build()
    _check_dropin_and_harness(ge, torch.device('cuda:0'), B=16, N=20, corrector=corrector)


@pytest.mark.gpu
def test_ode_sampler_gpu(params0):
    import __graft_entry__ as ge
    ge.build()
    _check_ode_sampler(ge, torch.device('cuda:0'), params0, B=4, span=0.2)


@pytest.mark.gpu
def test_ode_device_rk45_gpu(golden):
    import __graft_entry__ as ge
    ge.build()
    g = golden('ode_rk45.npz')
    _check_ode_device_vs_reference(ge, torch.device('cuda:0'), g, 'b4', [int(k) for k in g['b4.ks']], full=True)
    _check_ode_device_vs_reference(ge, torch.device('cuda:0'), g, 'b1', [int(k) for k in g['b1.ks']], full=True)


@pytest.mark.gpu
def test_sample_hk_gpu():
    import __graft_entry__ as ge
    ge.build()
    _check_sample_hk(torch.device('cuda:0'))
