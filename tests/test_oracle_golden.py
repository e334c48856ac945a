"""Pins the numpy oracle (oracle/rd_oracle.py) against fixtures recorded from the REFERENCE itself
(oracle/gen_golden.py, run in the build container).  CPU only."""
import numpy as np
import pytest

from oracle import rd_oracle as O
from oracle.weights import make_params, param_specs, params_sha256


def close(a, b, rtol=2e-5, atol=2e-5):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_reflect_known_answers(golden):
    g = golden('cube_sde.npz')
    # SURVEY 8a-A1 table
    close(O.reflect(np.array([-2.3, -1.2, -0.3, 1.2, 2.3, 3.7, 1, -1, 2, 0], np.float32)),
          np.array([0.3, 0.8, 0.3, 0.8, 0.3, 0.3, 1, 1, 0, 0], np.float32), atol=1e-6)
    assert np.array_equal(O.reflect(g['reflect_known_in']), g['reflect_known_out'])
    assert np.array_equal(O.reflect(g['reflect_rand_in']), g['reflect_rand_out'])


def test_score_hk(golden):
    g = golden('cube_sde.npz')
    t = g['hk_sigma'] ** 2 / 2
    close(O.score_hk_ef(g['hk_x'], g['hk_x0'], t), g['hk_ef_only'], rtol=2e-4, atol=1e-4)
    close(O.score_hk_refl(g['hk_x'], g['hk_x0'], t), g['hk_refl_only'], rtol=2e-4, atol=1e-3)
    close(O.score_hk(g['hk_x'], g['hk_x0'], g['hk_sigma']), g['hk_score'], rtol=2e-5, atol=1e-4)   # measured: 3.1e-5 abs, 2.1e-6 rel


def test_sde_schedule(golden):
    g = golden('cube_sde.npz')
    sde = O.RVESDE(0.01, 5, N=1000)
    close(sde.sigma(g['sde_t']), g['sde_sigma'], rtol=1e-6, atol=0)
    close(sde.g(g['sde_t']), g['sde_g'], rtol=1e-6, atol=0)
    assert np.array_equal(O.torch_linspace(1, 1e-5, 1000), g['ts_1000'])
    assert np.array_equal(O.torch_linspace(1, 1e-5, 10), g['ts_10'])


def test_param_recipe(golden, params0):
    g = golden('forward_9x9.npz')
    assert bytes(g['params_sha256']).hex() == params_sha256(params0)
    assert len(param_specs()) == 261
    assert sum(int(np.prod(s)) for _, s in param_specs()) == 6254913
    names = list(golden('init_seed0.npz')['names'])
    assert [n for n, _ in param_specs()] == names


def test_forward_9x9(golden, params0):
    g = golden('forward_9x9.npz')
    sde = O.RVESDE(0.01, 5, N=1000)
    taps = {}
    out = O.ncsnpp_forward(params0, g['x'], sde.sigma(g['t']), g['labels'], taps=taps)
    close(taps['temb'], g['temb'], rtol=1e-4, atol=1e-4)
    for k in g.files:
        if k.startswith('tap.'):
            close(taps[k[4:]][:2], g[k], rtol=1e-4, atol=1e-4)
    close(out, g['score'], rtol=1e-4, atol=1e-4)
    close(O.cf_score_fn(params0, sde, g['x'], g['t'], g['labels'], 0.0), g['cf_w0'], rtol=1e-4, atol=1e-4)
    close(O.cf_score_fn(params0, sde, g['x'], g['t'], g['labels'], None), g['cf_none'], rtol=1e-4, atol=1e-4)
    close(O.cf_score_fn(params0, sde, g['x'], g['t'], g['labels'], g['wt']), g['cf_wt'], rtol=1e-4, atol=2e-4)


def test_forward_8x9(golden, params0):
    g = golden('forward_8x9.npz')
    sde = O.RVESDE(0.01, 5, N=1000)
    close(O.score_fn(params0, sde, g['x'], g['t'], g['labels']), g['score'], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('tag,corr,wkey', [('none_w0', 'none', 0.0), ('langevin_w0', 'langevin', 0.0),
                                           ('none_wt', 'none', 'wt'), ('langevin_none', 'langevin', None)])
def test_sampler_10step(golden, params0, tag, corr, wkey):
    g = golden('sampler_10step.npz')
    sde = O.RVESDE(0.01, 5, N=10)
    w = g['wt'] if wkey == 'wt' else wkey
    steps = g[f'{tag}.steps']
    ts = O.torch_linspace(1, 1e-5, 10)
    amp = np.maximum(1.0, sde.g(ts[:9]) ** 2 / 10)          # per-update amplification of a score error
    trace = []
    x, nfe = O.pc_sampler(params0, sde, g[f'{tag}.prior'], list(g[f'{tag}.noises']), g['labels'], w, eps=1e-5,
                          snr=0.01, n_steps=1, corrector=corr, trace=trace, teacher=steps)
    assert nfe == int(g[f'{tag}.nfe']) == 20
    assert len(trace) == len(steps) == 9
    for i, (a, b) in enumerate(zip(trace, steps)):
        close(a, b, rtol=0, atol=2e-5 * amp[i])              # teacher-forced, per update
    # free run: N=10 makes the first updates chaotic (x += 31*score), so only a loose (median) end-to-end bound
    x, _ = O.pc_sampler(params0, sde, g[f'{tag}.prior'], list(g[f'{tag}.noises']), g['labels'], w, eps=1e-5,
                        snr=0.01, n_steps=1, corrector=corr)
    assert np.median(np.abs(x - g[f'{tag}.x'])) < 3e-2
    assert x.min() >= 0 and x.max() <= 1


def test_train_loss(golden, params0):
    g = golden('train_step.npz')
    sde = O.RVESDE(0.01, 5, N=1000)
    loss, *_ = O.sde_loss(params0, sde, g['batch'], g['labels'], g['step0.t'], g['step0.z'])
    close(loss, g['step0.loss'], rtol=2e-4, atol=0)


def test_torch_flavour_oracle(golden, params0):
    """oracle/rd_oracle_torch.py (the cpu_baseline leg of bench.py) against the reference fixtures."""
    import torch
    from oracle import rd_oracle_torch as OT
    p = {k: torch.from_numpy(v) for k, v in params0.items()}
    g = golden('forward_9x9.npz')
    x, t, lab = (torch.from_numpy(g[k]) for k in ('x', 't', 'labels'))
    with torch.no_grad():
        close(OT.ncsnpp_forward(p, x, OT.sigma_of(t), lab).numpy(), g['score'], rtol=1e-4, atol=1e-4)
        close(OT.cf_score(p, x, t, lab, torch.from_numpy(g['wt'])).numpy(), g['cf_wt'], rtol=1e-4, atol=2e-4)
    gs = golden('sampler_10step.npz')
    ts = O.torch_linspace(1, 1e-5, 10)
    steps, nz = gs['langevin_w0.steps'], gs['langevin_w0.noises']
    xcur = torch.from_numpy(gs['langevin_w0.prior'])
    amp = np.maximum(1.0, O.RVESDE(0.01, 5, N=10).g(ts[:9]) ** 2 / 10)
    with torch.no_grad():
        for i in range(9):
            tt = torch.full((8,), float(ts[i]))
            xn = OT.pc_update(p, xcur, tt, torch.from_numpy(gs['labels']), torch.zeros(8), torch.from_numpy(nz[2 * i + 1]), 10,
                              z_corr=torch.from_numpy(nz[2 * i]))
            close(xn.numpy(), steps[i], rtol=0, atol=2e-5 * amp[i])
            xcur = torch.from_numpy(steps[i])


def test_gto_spherical_matches_reference_helper(golden):
    """vectors recorded from the reference's own _convert_to_spherical (Benchmark/gto_halo_benchmarking.py:335-361),
    including |u| = 0, axis-aligned directions (alpha / theta wrap to [0, 2pi)) and magnitudes > 1 (clipped, counted)."""
    g = golden('gto_unnormalize.npz')
    a, th, u, clips = O.convert_to_spherical(g['ux'], g['uy'], g['uz'])
    assert np.array_equal(a, g['alpha']) and np.array_equal(th, g['theta']) and np.array_equal(u, g['u'])
    assert clips == int(g['clips'])


def test_gto_unnormalize_known_answers():
    """The affine part of :255-333 has no importable reference (omegaconf) -> hand-computed known answers ("parity unpinned")."""
    s = np.full((2, 81), 0.5, np.float32)
    s[1, 0] = 1.0
    out, clips = O.gto_unnormalize(s)
    assert out.shape == (2, 67) and clips == 0
    m = 0.5 * 0.1811 + 0.4652
    np.testing.assert_allclose(out[0, 0], 0.5 * 0.087 + 0.008, rtol=1e-6)
    np.testing.assert_allclose(out[1, 0], 0.095, rtol=1e-6)
    np.testing.assert_allclose(out[0, 1:4], [m * 40, m * 15, m * 15], rtol=1e-6)
    c = m * 2 - 1
    np.testing.assert_allclose(out[0, 4:64].reshape(20, 3), np.tile([np.pi / 4, np.arcsin(1 / np.sqrt(3)), c * np.sqrt(3)], (20, 1)), rtol=2e-6)
    np.testing.assert_allclose(out[0, 64:], [m * 62 + 408, m, m * 6 + 5], rtol=1e-6)


def test_gto_dataset_item_matches_reference_class(golden):
    """items recorded from the reference's GTOHaloImageDataset.__getitem__ (RD/datasets.py:88-98)."""
    g = golden('gto_dataset.npz')
    for i in range(len(g['data'])):
        img, lab = O.gto_image_item(g['data'][i])
        assert np.array_equal(img.reshape(1, 9, 9), g['images'][i]) and np.array_equal(lab, g['labels'][i])



def test_gto_unnormalize_all_columns_against_reference_block(golden):
    """N1: the oracle's un-normalisation against the output of the reference's own inline block
    (Benchmark/gto_halo_benchmarking.py:254-328, executed from its AST by oracle/gen_golden.py) -- all 67 columns, bit for bit."""
    from oracle import rd_oracle as O
    g = golden('gto_unnormalize.npz')
    ref, clips = O.gto_unnormalize(g['full_in'].reshape(g['full_in'].shape[0], -1))
    assert np.array_equal(ref, g['full_out']) and clips == int(g['full_clips'])
