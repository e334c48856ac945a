"""BASELINE config #5 on the GPU: the CIFAR-shape NCSN++ (RD/configs/model/ddpmpp.yaml completed per SURVEY F9: nf=128,
ch_mult [1,2,2,2], 8 res blocks per level, attention at 16x16 over L=256 positions with C=256, scale_by_sigma, 32x32x3 input,
zero labels; 104.7 M parameters, 18.5 GMAC per sample-forward) through the spatially tiled plan (csrc/tiled_kernels.h), fp32.

Pinned by tests/golden/forward_cifar.npz, recorded by oracle/gen_golden.py from the IMPORTED reference model with the seeded
synthetic weights (strict=True load): whole forward of two samples and seven intermediate activations (strided subsamples).
Tolerance: 2e-5 relative to each sample's largest |score| (fp32 through ~75 convs / 17 attention blocks; measured 5e-6)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def env():
    import __graft_entry__ as ge
    ge.build()
    dev = torch.device('cuda:0')
    model, cfg, params = ge.make_cifar_model(dev)
    return dict(ge=ge, dev=dev, model=model, cfg=cfg, params=params)


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_cifar_forward_golden_and_taps(env, golden):
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    g = golden('forward_cifar.npz')
    dev, model = env['dev'], env['model']
    assert sum(p.numel() for p in model.parameters()) == int(g['n_params']) == 104701571
    sde = sde_lib.RVESDE(0.01, 50, N=1000)
    x, t, lab = T(g['x'], dev), T(g['t'], dev), T(g['labels'], dev)
    with torch.no_grad():
        s = mutils.get_score_fn(sde, model)(x, t, class_labels=lab)
        s_model = mutils.get_model_fn(model)(x, sde.marginal_prob(x, t)[1], class_labels=lab)
    ctx = model._ctx[(str(dev), 32, 32)]
    assert ctx.path_info().startswith('tiled'), ctx.path_info()
    ref = g['score']
    for out in (s.cpu().numpy(), s_model.cpu().numpy()):
        assert out.shape == (2, 3, 32, 32)
        for n in range(2):
            assert np.abs(out[n] - ref[n]).max() <= 2e-5 * np.abs(ref[n]).max(), n
    for k in g.files:
        if k.startswith('tap.'):
            a = ctx.get_tap(k[4:], x, 2).cpu().numpy()[:, ::8, ::4, ::4]
            # intermediate activations: fp32 rounding accumulated through up to 32 residual blocks (|activation| ~ 1-2)
            np.testing.assert_allclose(a, g[k], rtol=0, atol=1e-4 * max(1.0, float(np.abs(g[k]).max())), err_msg=k)


def test_cifar_cfg_score_and_pc_update_vs_oracle(env):
    """Classifier-free-guidance score (2B forward) and one reflected PC update of the CIFAR-shape model against the torch oracle
    (oracle/rd_oracle_torch.py with the CIFAR architecture keys; the oracle's architecture-generic forward is the one pinned to the
    reference by the 9x9 fixtures, and this model's forward is pinned by forward_cifar.npz above)."""
    from oracle import rd_oracle as O
    from oracle import rd_oracle_torch as OT
    from rdmi import sampling, sde_lib
    from rdmi.models import utils as mutils
    dev, model, params = env['dev'], env['model'], env['params']
    pt = {k: torch.from_numpy(v) for k, v in params.items()}
    B, N = 2, 1000
    sde = sde_lib.RVESDE(0.01, 50, N=N)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(B, 3, 32, 32, generator=g); lab = torch.zeros(B, 1); w = torch.tensor([0.0, 0.7])
    t = torch.tensor([0.6, 0.25])
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        s = mutils.get_cf_score_fn(sde, model, lab.to(dev), w.to(dev))(x.to(dev), t.to(dev)).cpu()
        ref = OT.cf_score(pt, x, t, lab, w, smax=50.0, **OT.CIFAR_ARCH)
    for n in range(B):
        assert float((s[n] - ref[n]).abs().max()) <= 5e-5 * float(ref[n].abs().max()), n
    # one PC update of the fused C loop (teacher forcing off, injected noise) at the start and near the end of the schedule
    ts = O.torch_linspace(1, 1e-5, N)
    noise = torch.randn(N - 1, B, 3 * 32 * 32, generator=g)
    teacher = torch.rand(N - 1, B, 3 * 32 * 32, generator=g)
    # a full 999-update run of this model is minutes of GPU time: run the first 3 updates of a 1000-scale schedule instead, by
    # checking trace rows 0..2 of a run whose remaining updates are cut with a short schedule of the SAME time grid start
    Ns = 4
    sde4 = sde_lib.RVESDE(0.01, 50, N=Ns)
    ts4 = O.torch_linspace(1, 1e-5, Ns)
    trace = torch.zeros(Ns - 1, B, 3 * 32 * 32, device=dev)
    fn = sampling.get_pc_sampler(sde4, (B, 3, 32, 32), sampling.get_predictor('euler_maruyama'), sampling.get_corrector('none'),
                                 sampling.get_denoiser('none'), 0.01, 1, 1e-5, dev, noise=noise[:Ns - 1].to(dev), trace=trace,
                                 teacher=teacher[:Ns - 1].to(dev))
    _rand = torch.rand
    torch.rand = lambda *a, **k: x.clone()
    try:
        xs, nfe = fn(model, weight=w.to(dev), class_labels=lab.to(dev))
    finally:
        torch.rand = _rand
    assert nfe == Ns * 2
    trace = trace.cpu()
    for i in range(Ns - 1):
        x_prev = x if i == 0 else teacher[i - 1].reshape(B, 3, 32, 32)
        with torch.no_grad():
            r = OT.pc_update(pt, x_prev, torch.full((B,), float(ts4[i])), lab, w, noise[i].reshape(B, 3, 32, 32), Ns, smax=50.0, **OT.CIFAR_ARCH)
        # x' = reflect(x + g^2/N * score + ...): the score tolerance (5e-5 of |score|, with |score| ~ 1/sigma under scale_by_sigma)
        # is amplified by g(t)^2 / N
        gg = float(OT.g_of(torch.tensor([float(ts4[i])]), smax=50.0)[0]) ** 2 / Ns
        sc = float(OT.cf_score(pt, x_prev, torch.full((B,), float(ts4[i])), lab, w, smax=50.0, **OT.CIFAR_ARCH).abs().max())
        err = float((trace[i].reshape(B, 3, 32, 32) - r).abs().max())
        assert err <= 2e-5 + 1e-4 * gg * sc, (i, err, gg, sc)
    assert float(xs.min()) >= 0 and float(xs.max()) <= 1


def test_cifar_bf16_forward_within_bf16_tolerance(env, golden):
    """compute_dtype='bf16' (BASELINE config #5 as written: bf16 MFMA operands, fp32 accumulate / GroupNorm / softmax / tensors)
    against the same reference-recorded forward.  Stated tolerance: 3e-2 of each sample's largest |score| at the maximum and 6e-3
    in the root-mean-square (bf16 keeps 8 significant bits: 2^-9 relative rounding per operand through ~75 convolutions and 17
    attention blocks; measured 1.4e-2 / 2.5e-3), and within that the fp32 plan (compute_dtype='f32') stays the parity reference:
    its own tolerance is unchanged (test above)."""
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    g = golden('forward_cifar.npz')
    dev, ge = env['dev'], env['ge']
    model, cfg, _ = ge.make_cifar_model(dev, compute_dtype='bf16')
    sde = sde_lib.RVESDE(0.01, 50, N=1000)
    with torch.no_grad():
        s = mutils.get_score_fn(sde, model)(T(g['x'], dev), T(g['t'], dev), class_labels=T(g['labels'], dev)).cpu().numpy()
    assert 'bf16' in model._ctx[(str(dev), 32, 32)].path_info()
    ref = g['score']
    for n in range(2):
        d = s[n] - ref[n]
        assert np.abs(d).max() <= 3e-2 * np.abs(ref[n]).max(), (n, np.abs(d).max() / np.abs(ref[n]).max())
        assert np.sqrt((d ** 2).mean()) <= 6e-3 * np.abs(ref[n]).max(), (n, np.sqrt((d ** 2).mean()) / np.abs(ref[n]).max())
    # not a silent fp32 run: the result differs from the fp32 plan's
    with torch.no_grad():
        s32 = mutils.get_score_fn(sde, env['model'])(T(g['x'], dev), T(g['t'], dev), class_labels=T(g['labels'], dev)).cpu().numpy()
    assert np.abs(s - s32).max() > 1e-4 * np.abs(ref).max()


def test_cifar_bf16_cfg_score_and_pc_updates_batch8(env):
    """The bf16 plan at a batch that exercises the tiling (B = 8 with guidance: 16 forwards per launch, flash_attn_bf16_kernel
    over 16 x 256 positions) -- classifier-free-guidance score with per-sample weights and reflected PC updates of the REAL
    1000-scale schedule (teacher forcing, injected noise; updates 400, 700 and 998) against the fp32 torch oracle, within the
    stated bf16 tolerance: score 3e-2 of each sample's largest |score| (max) / 6e-3 (rms); an update 2e-5 + 3e-2 * g(t)^2/N *
    max|score| (the reverse SDE multiplies the score error by g^2/N)."""
    from oracle import rd_oracle as O
    from oracle import rd_oracle_torch as OT
    from rdmi import sampling, sde_lib
    from rdmi.models import utils as mutils
    dev, ge, params = env['dev'], env['ge'], env['params']
    model, cfg, _ = ge.make_cifar_model(dev, compute_dtype='bf16')
    pt = {k: torch.from_numpy(v) for k, v in params.items()}
    B, N = 8, 1000
    sde = sde_lib.RVESDE(0.01, 50, N=N)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, 3, 32, 32, generator=g); lab = torch.zeros(B, 1); w = torch.rand(B, generator=g)
    t = torch.rand(B, generator=g) * 0.9 + 0.05
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        s = mutils.get_cf_score_fn(sde, model, lab.to(dev), w.to(dev))(x.to(dev), t.to(dev)).cpu()
        ref = OT.cf_score(pt, x, t, lab, w, smax=50.0, **OT.CIFAR_ARCH)
    assert 'bf16' in model._ctx[(str(dev), 32, 32)].path_info()
    for n in range(B):
        d = s[n] - ref[n]
        amp = float(ref[n].abs().max()) * (1 + 2 * float(w[n]))          # (1+w) s_c - w s_u: both halves carry bf16 rounding
        assert float(d.abs().max()) <= 3e-2 * amp, (n, float(d.abs().max()) / amp)
        assert float((d ** 2).mean().sqrt()) <= 6e-3 * amp, n
    E = 3 * 32 * 32
    noise = torch.randn(N - 1, B, E, generator=g)
    teacher = torch.rand(N - 1, B, E, generator=g)
    trace = torch.zeros(N - 1, B, E, device=dev)
    fn = sampling.get_pc_sampler(sde, (B, 3, 32, 32), sampling.get_predictor('euler_maruyama'), sampling.get_corrector('none'),
                                 sampling.get_denoiser('none'), 0.01, 1, 1e-5, dev, noise=noise.to(dev), trace=trace, teacher=teacher.to(dev))
    _rand = torch.rand
    torch.rand = lambda *a, **k: x.clone()
    try:
        xs, nfe = fn(model, weight=w.to(dev), class_labels=lab.to(dev))
    finally:
        torch.rand = _rand
    assert nfe == 2 * N
    trace = trace.cpu()
    assert torch.isfinite(trace).all() and float(trace.min()) >= 0 and float(trace.max()) <= 1
    ts = O.torch_linspace(1, 1e-5, N)
    for i in (400, 700, 998):
        x_prev = teacher[i - 1].reshape(B, 3, 32, 32)
        tv = torch.full((B,), float(ts[i]))
        with torch.no_grad():
            r = OT.pc_update(pt, x_prev, tv, lab, w, noise[i].reshape(B, 3, 32, 32), N, smax=50.0, **OT.CIFAR_ARCH)
            sc = float((OT.cf_score(pt, x_prev, tv, lab, torch.zeros(B), smax=50.0, **OT.CIFAR_ARCH).abs().max()) * 3)
        gg = float(OT.g_of(tv[:1], smax=50.0)[0]) ** 2 / N
        d = (trace[i].reshape(B, 3, 32, 32) - r).abs()
        d = torch.minimum(d, 1 - d)                                        # a value reflected at the other face of the cube is the same point
        assert float(d.max()) <= 2e-5 + 3e-2 * gg * sc, (i, float(d.max()), gg, sc)


def test_cifar_bf16_implicit_gemm_conv_equals_window_conv(env, golden, monkeypatch):
    """iconv_kernel (implicit GEMM on 128 x 128 tiles, operands staged by the LDS-DMA; an alternative conv form that measured 8 %
    slower than the window form on MI355X and is therefore off unless RDMI_ICONV=1) against tconv_pre_kernel (64-pixel LDS window)
    on the same bf16 operands: the two differ only in the fp32
    summation order, so the scores agree far inside the bf16 tolerance (stated: 2e-3 of each sample's largest |score|; measured
    ~3e-4), the forced run meets the reference-recorded forward within the stated bf16 tolerance, and its profile shows the
    kernel on the 32x32 / 16x16 / 8x8 levels (B = 2 and a ragged B = 3: a last row tile with one valid half)."""
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    g = golden('forward_cifar.npz')
    dev, ge = env['dev'], env['ge']
    sde = sde_lib.RVESDE(0.01, 50, N=1000)
    x3 = torch.cat([T(g['x'], dev), T(g['x'], dev)[:1].flip(-1)]); t3 = torch.tensor([float(g['t'][0]), float(g['t'][1]), 0.37], device=dev)
    lab3 = torch.cat([T(g['labels'], dev), T(g['labels'], dev)[:1]])

    def run(envvars):
        for k, v in envvars.items():
            monkeypatch.setenv(k, v)
        model, cfg, _ = ge.make_cifar_model(dev, compute_dtype='bf16')
        with torch.no_grad():
            s2 = mutils.get_score_fn(sde, model)(T(g['x'], dev), T(g['t'], dev), class_labels=T(g['labels'], dev)).cpu().numpy()
            ctx = model._ctx[(str(dev), 32, 32)]
            ctx.set_profiling(True)
            s3 = mutils.get_score_fn(sde, model)(x3, t3, class_labels=lab3).cpu().numpy()
        names = {p['kernel']: p['launches'] for p in ctx.get_profile()}
        for k in envvars:
            monkeypatch.delenv(k)
        return s2, s3, names
    a2, a3, na = run({'RDMI_ICONV_MIN_WGS': '1'})
    b2, b3, nb = run({'RDMI_ICONV': '0'})
    assert na.get('iconv_kernel<bf16>', 0) >= 100 and 'iconv_kernel<bf16>' not in nb, (na, nb)
    ref = g['score']
    for n in range(2):
        amp = np.abs(ref[n]).max()
        assert np.abs(a2[n] - ref[n]).max() <= 3e-2 * amp and np.sqrt(((a2[n] - ref[n]) ** 2).mean()) <= 6e-3 * amp, n
        assert np.abs(a2[n] - b2[n]).max() <= 2e-3 * amp, (n, np.abs(a2[n] - b2[n]).max() / amp)
    for n in range(3):
        assert np.abs(a3[n] - b3[n]).max() <= 2e-3 * np.abs(b3[n]).max(), (n, np.abs(a3[n] - b3[n]).max() / np.abs(b3[n]).max())
    assert np.abs(a3[:2] - a2).max() <= 2e-3 * np.abs(a2).max()        # a sample's score does not depend on the batch around it


def test_cifar_bf16_vector_epilogue_is_bit_identical(env, golden, monkeypatch):
    """The bf16 packs interleave their columns so that, at the workgroup widths of a sampling batch (NCT = 2 / 4 column tiles per wave,
    forced here with RDMI_TILED_MIN_WGS=1), a lane's accumulators of one row are adjacent output columns and the conv epilogue moves
    8- / 16-byte vectors.  Neither the interleave nor the width changes any element's arithmetic: the forward is bit-identical with
    the plain column order (RDMI_NO_COL_IL=1) and with the narrow workgroups a small batch gets by default.  Likewise the bf16 storage of
    the attention blocks' q | k | v projection (the fused core rounds them to bf16 anyway) against fp32 storage (RDMI_NO_QKV16=1)."""
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    g = golden('forward_cifar.npz')
    dev, ge = env['dev'], env['ge']
    sde = sde_lib.RVESDE(0.01, 50, N=1000)

    def run(envvars):
        for k, v in envvars.items():
            monkeypatch.setenv(k, v)
        model, cfg, _ = ge.make_cifar_model(dev, compute_dtype='bf16')
        with torch.no_grad():
            s = mutils.get_score_fn(sde, model)(T(g['x'], dev), T(g['t'], dev), class_labels=T(g['labels'], dev)).cpu().numpy()
        for k in envvars:
            monkeypatch.delenv(k)
        return s
    wide = run({'RDMI_TILED_MIN_WGS': '1'})
    wide_plain = run({'RDMI_TILED_MIN_WGS': '1', 'RDMI_NO_COL_IL': '1'})
    narrow = run({})
    qkv32 = run({'RDMI_NO_QKV16': '1'})
    assert np.isfinite(wide).all()
    assert np.array_equal(wide, wide_plain)
    assert np.array_equal(wide, narrow)
    # q | k | v stored as bf16 by the projection (default) or as fp32 and rounded by the attention core: the same bits reach the MFMAs
    # (1 / sqrt(256) is a power of two, so scaling q commutes with the rounding)
    assert np.array_equal(narrow, qkv32)
    ref = g['score']
    for n in range(2):
        assert np.abs(wide[n] - ref[n]).max() <= 3e-2 * np.abs(ref[n]).max()


def test_cifar_bf16_bench_batch_against_fp32_plan(env):
    """The bf16 plan at the batch `bench.py` times (`variants.cifar_b64_bf16`: B = 64 with guidance = 128 forwards per launch, where the
    convs run their wide workgroups -- 2 / 4 column tiles per wave, the vector epilogue over interleaved weight columns -- the activation
    pass folds the GroupNorm statistics and the attention core reads bf16 q | k | v) against the fp32 plan on the same inputs.  The fp32
    plan is the parity reference (pinned to the reference-recorded forward at 2e-5 above); stated bf16 tolerance as everywhere: 3e-2 of
    each sample's largest |score| at the maximum, 6e-3 in the root-mean-square, both scaled by (1 + 2 w) for the guidance combination.
    At the time of update 700 of the real 1000-scale schedule the score difference times g(t)^2/N (what a reflected PC update adds to x)
    stays within 2e-5 + 3e-2 g(t)^2/N max|score|."""
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    dev, ge = env['dev'], env['ge']
    m16, _, _ = ge.make_cifar_model(dev, compute_dtype='bf16')
    m32 = env['model']
    B, N = 64, 1000
    sde = sde_lib.RVESDE(0.01, 50, N=N)
    g = torch.Generator().manual_seed(23)
    x = torch.rand(B, 3, 32, 32, generator=g).to(dev); lab = torch.zeros(B, 1, device=dev); w = torch.rand(B, generator=g).to(dev)
    t = (torch.rand(B, generator=g) * 0.9 + 0.05).to(dev)
    with torch.no_grad():
        s16 = mutils.get_cf_score_fn(sde, m16, lab, w)(x, t).cpu()
        s32 = mutils.get_cf_score_fn(sde, m32, lab, w)(x, t).cpu()
    info = m16._ctx[(str(dev), 32, 32)].path_info()
    assert 'bf16' in info and 'tiled' in info, info
    assert torch.isfinite(s16).all()
    worst = 0.0
    for n in range(B):
        d = s16[n] - s32[n]
        amp = float(s32[n].abs().max()) * (1 + 2 * float(w[n]))
        assert float(d.abs().max()) <= 3e-2 * amp, (n, float(d.abs().max()) / amp)
        assert float((d ** 2).mean().sqrt()) <= 6e-3 * amp, n
        worst = max(worst, float(d.abs().max()) / amp)
    assert worst > 1e-5          # not a silent fp32 run
    # the same at ONE time of the real schedule (update 700), as a PC update sees it: x' = reflect(x + g^2/N * score + g/sqrt(N) z), so the
    # two plans' updates from one state and one noise draw differ by g^2/N times their score difference
    i = 700
    ts = torch.linspace(1, 1e-5, N)
    tv = torch.full((B,), float(ts[i]), device=dev)
    with torch.no_grad():
        sc16 = mutils.get_cf_score_fn(sde, m16, lab, w)(x, tv).cpu()
        sc32 = mutils.get_cf_score_fn(sde, m32, lab, w)(x, tv).cpu()
    from oracle import rd_oracle_torch as OT
    gg = float(OT.g_of(tv[:1].cpu(), smax=50.0)[0]) ** 2 / N
    amp = float(sc32.abs().max()) * 3
    assert float((sc16 - sc32).abs().max()) * gg <= 2e-5 + 3e-2 * gg * amp


def test_bf16_is_refused_where_it_is_not_built(env):
    """The 9x9 GTO-Halo plans are fp32 only: asking bf16 there fails loudly instead of silently computing in fp32."""
    ge, dev = env['ge'], env['dev']
    from rdmi import sde_lib
    from rdmi.models import utils as mutils
    model, cfg, _ = ge.make_model(dev)
    model.compute_dtype = 'bf16'
    model._ctx.clear()
    with torch.no_grad(), pytest.raises(RuntimeError, match='bf16'):
        mutils.get_score_fn(sde_lib.RVESDE(0.01, 5, N=1000), model)(torch.rand(2, 1, 9, 9, device=dev), torch.rand(2, device=dev), class_labels=torch.rand(2, 1, device=dev))
