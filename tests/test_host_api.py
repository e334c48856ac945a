"""Host-side mirror of the reference interface: registries, error behaviour, parameter layout/initialisation,
EMA arithmetic, SDE schedule -- no kernels involved (CPU)."""
import hashlib

import numpy as np
import pytest
import torch


def test_model_registry_semantics():
    from rdmi.models import utils as mutils
    assert mutils.get_model('ncsnpp').__name__ == 'NCSNpp'
    with pytest.raises(ValueError, match='Already registered'):
        @mutils.register_model(name='ncsnpp')
        class Dup:                                   # noqa
            pass

    @mutils.register_model
    class _TmpModelForTest:                          # bare decorator form registers under the class name
        pass
    assert mutils.get_model('_TmpModelForTest') is _TmpModelForTest
    with pytest.raises(KeyError):
        mutils.get_model('nope')


def test_sampler_registries_and_dispatch():
    from types import SimpleNamespace as NS
    from rdmi import sampling, sde_lib
    assert sampling.get_predictor('euler_maruyama') is sampling.ReflectedEulerMaruyamaPredictor
    assert sampling.get_corrector('langevin') is sampling.ReflectedLangevinCorrector
    assert sampling.get_corrector('none') is sampling.NoneCorrector
    assert {k: sampling.get_denoiser(k).__name__ for k in ('network', 'mean', 'none')} == \
        {'network': 'TrainedDenoiser', 'mean': 'MeanDenoiser', 'none': 'NoneDenoiser'}
    with pytest.raises(KeyError):
        sampling.get_predictor('ancestral')
    with pytest.raises(ValueError, match='Already registered'):
        sampling.register_corrector(name='langevin')(type('X', (), {}))
    cfg = NS(sampling=NS(method='bogus'))
    with pytest.raises(ValueError, match='unknown'):
        sampling.get_sampling_fn(cfg, sde_lib.RVESDE(0.01, 5, N=10), (2, 1, 9, 9), 1e-5, 'cpu')
    cfg = NS(sampling=NS(method='pc', predictor='euler_maruyama', corrector='none', denoiser='none', snr=0.01, n_steps_each=1))
    fn = sampling.get_sampling_fn(cfg, sde_lib.RVESDE(0.01, 5, N=10), (2, 1, 9, 9), 1e-5, 'cpu')
    assert callable(fn)
    x = torch.rand(2, 1, 9, 9)
    assert sampling.NoneCorrector(None, None, 0.1, 1).update_fn(x, None)[0] is x
    assert sampling.MeanDenoiser(None).update_fn(x, x + 1, None).equal(x + 1)


def test_state_dict_layout_and_reference_init(golden):
    """261 names in the reference's registration order; seeded init consumes the torch RNG exactly like the reference
    (bit-identical parameters under torch.manual_seed(0)): fixture recorded from the reference."""
    import __graft_entry__ as ge
    from rdmi.models import utils as mutils
    g = golden('init_seed0.npz')
    torch.manual_seed(0)
    model = mutils.create_model(ge.demo_config())
    sd = model.state_dict()
    assert list(sd.keys()) == list(g['names'])
    assert sum(v.numel() for v in sd.values()) == 6254913
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode()); h.update(v.numpy().tobytes())
    assert h.digest() == bytes(g['sha256'])
    assert [n for n, p in model.named_parameters() if not p.requires_grad] == ['time_embed.W']


def test_unsupported_configs_fail_loudly():
    import __graft_entry__ as ge
    from rdmi.models import utils as mutils
    cfg = ge.demo_config(); cfg.model.nonlinearity = 'elu'
    with pytest.raises(NotImplementedError):
        mutils.create_model(cfg)
    cfg = ge.demo_config(); cfg.model.fir = True
    with pytest.raises(NotImplementedError):
        mutils.create_model(cfg)
    cfg = ge.demo_config(); cfg.model.embedding_type = 'positional'
    with pytest.raises(NotImplementedError, match='fourier'):
        mutils.create_model(cfg)


def test_ema_matches_reference_formula():
    from rdmi.models.ema import ExponentialMovingAverage
    torch.manual_seed(1)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2), requires_grad=False)]
    ema = ExponentialMovingAverage(ps, decay=0.999)
    assert len(ema.shadow_params) == 2
    shadow = [p.detach().clone() for p in ps[:2]]
    for step in range(1, 6):
        with torch.no_grad():
            for p in ps[:2]:
                p.add_(0.1 * torch.randn_like(p))
        ema.update(ps)
        d = min(0.999, (1 + step) / (10 + step))
        for s, p in zip(shadow, ps[:2]):
            s.sub_((1.0 - d) * (s - p.detach()))
        for a, b in zip(ema.shadow_params, shadow):
            assert torch.equal(a, b)
    before = [p.detach().clone() for p in ps]
    ema.store(ps); ema.copy_to(ps)
    assert torch.equal(ps[0].detach(), ema.shadow_params[0]) and torch.equal(ps[2].detach(), before[2])
    ema.restore(ps)
    assert all(torch.equal(a.detach(), b) for a, b in zip(ps, before))
    sd = ema.state_dict()
    assert set(sd) == {'decay', 'num_updates', 'shadow_params'} and sd['num_updates'] == 5
    with pytest.raises(ValueError):
        ExponentialMovingAverage(ps, decay=1.5)


def test_rvesde_schedule(golden):
    from rdmi import sde_lib
    g = golden('cube_sde.npz')
    sde = sde_lib.RVESDE(0.01, 5, N=1000)
    t = torch.from_numpy(g['sde_t'])
    x = torch.zeros(t.numel(), 1, 9, 9)
    assert np.array_equal(sde.marginal_prob(x, t)[1].numpy(), g['sde_sigma'])
    drift, diff = sde.sde(x, t)
    assert np.array_equal(diff.numpy(), g['sde_g']) and float(drift.abs().max()) == 0
    assert sde.T == 1 and sde.prior_sampling((2, 3)).shape == (2, 3) and float(sde.prior_logp(x).abs().max()) == 0
    rs = sde.reverse(lambda xx, tt: torch.ones_like(xx), probability_flow=True)
    d, gdiff = rs.sde(x, t)
    np.testing.assert_allclose(d[:, 0, 0, 0].numpy(), -0.5 * g['sde_g'] ** 2, rtol=1e-6)
    assert float(gdiff.abs().max()) == 0
    f, G = sde.discretize(x, t)
    assert G.shape == t.shape and float(f.abs().max()) == 0


def test_flatten_helpers_and_sigmas():
    from types import SimpleNamespace as NS
    from rdmi.models import utils as mutils
    x = torch.arange(12.).reshape(3, 4)
    assert mutils.from_flattened_numpy(mutils.to_flattened_numpy(x), (3, 4)).equal(x)
    s = mutils.get_sigmas(NS(sde=NS(sigma_max=5, sigma_min=0.01, num_scales=7)))
    assert s.shape == (7,) and np.isclose(s[0], 5) and np.isclose(s[-1], 0.01)


class _NotPlain:                                        # stands for the omegaconf object the reference pickles
    pass


def test_checkpoint_layout_round_trip(tmp_path, golden):
    """the reference's checkpoint layout (RD/utils.py:48-86; EMA list layout RD/models/ema.py:92-99) written and
    read back with the safe loader only; a reference-style file that carries a non-plain `config` object is refused loudly."""
    import __graft_entry__ as ge
    from rdmi import losses, utils
    from rdmi.models import utils as mutils
    from rdmi.models.ema import ExponentialMovingAverage
    cfg = ge.demo_config()
    torch.manual_seed(0)
    model = mutils.create_model(cfg)
    opt = losses.get_optimizer(cfg, model.parameters())
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    for p in model.parameters():                       # one fake optimisation step so optimizer/EMA state is non-trivial
        if p.requires_grad:
            p.grad = torch.full_like(p, 1e-3)
    opt.step(); ema.update(model.parameters())
    state = dict(optimizer=opt, model=model, ema=ema, step=7, scaler=None, config=cfg)
    path = str(tmp_path / 'checkpoint_1.pth')
    utils.save_checkpoint(path, state)
    raw = torch.load(path, weights_only=True)
    assert set(raw) == {'step', 'model', 'optimizer', 'ema', 'scaler', 'config'}
    assert list(raw['model'].keys()) == list(golden('init_seed0.npz')['names'])       # the reference's own state_dict names
    assert set(raw['ema']) == {'decay', 'num_updates', 'shadow_params'} and len(raw['ema']['shadow_params']) == 260
    assert raw['config']['model']['nf'] == cfg.model.nf

    torch.manual_seed(1)
    model2 = mutils.create_model(cfg)
    opt2 = losses.get_optimizer(cfg, model2.parameters())
    ema2 = ExponentialMovingAverage(model2.parameters(), decay=cfg.model.ema_rate)
    ptrs = [p.data_ptr() for p in model2.parameters()]
    st2 = utils.restore_checkpoint(path, dict(optimizer=opt2, model=model2, ema=ema2, step=0, scaler=None), 'cpu')
    assert st2['step'] == 7 and ema2.num_updates == ema.num_updates
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), model2.state_dict().values()))
    assert all(torch.equal(a, b) for a, b in zip(ema.shadow_params, ema2.shadow_params))
    assert [p.data_ptr() for p in model2.parameters()] == ptrs        # loaded in place: borrowed device pointers stay valid
    s1, s2 = opt.state_dict()['state'], opt2.state_dict()['state']
    assert all(torch.equal(s1[k]['exp_avg'], s2[k]['exp_avg']) for k in s1)

    model3 = utils.load_denoising_model(path, mutils.create_model(cfg))
    assert torch.equal(model3.state_dict()['out_conv.weight'], model.state_dict()['out_conv.weight'])
    with pytest.raises(ValueError, match='No checkpoint'):
        utils.load_denoising_model(str(tmp_path / 'missing.pth'), model3)
    # untouched state + directory creation when the file is absent (RD/utils.py:49-53)
    st = dict(optimizer=opt, model=model, ema=ema, step=3, scaler=None)
    assert utils.restore_checkpoint(str(tmp_path / 'new' / 'checkpoint.pth'), st, 'cpu') is st and (tmp_path / 'new').is_dir()

    torch.save({'model': model.state_dict(), 'config': _NotPlain()}, str(tmp_path / 'ref_style.pth'))
    with pytest.raises(RuntimeError, match='weights_only'):
        utils.load_denoising_model(str(tmp_path / 'ref_style.pth'), model3)



def _write_reference_layout_checkpoint(path, model, opt, ema, cfg_dict):
    """A checkpoint file in EXACTLY the layout RD/utils.py:78-86 writes, including a pickled omegaconf-style `config` object:
    classes named omegaconf.dictconfig.DictConfig / omegaconf.listconfig.ListConfig / omegaconf.nodes.AnyNode /
    omegaconf.base.{ContainerMetadata,Metadata} with omegaconf's state layout (_content / _val / _parent / _metadata, typing.Any and
    builtins.dict references).  omegaconf itself is not installed here, so the classes are created under those module names only
    while the file is written and removed again: the loader under test never sees them."""
    import sys
    import types
    import typing
    made = []

    def mk(modname, clsname):
        if modname not in sys.modules:
            sys.modules[modname] = types.ModuleType(modname); made.append(modname)
        cls = type(clsname, (), {'__module__': modname})
        setattr(sys.modules[modname], clsname, cls)
        return cls
    for mn in ('omegaconf',):
        if mn not in sys.modules:
            sys.modules[mn] = types.ModuleType(mn); made.append(mn)
    CM, MD = mk('omegaconf.base', 'ContainerMetadata'), mk('omegaconf.base', 'Metadata')
    DC, LC, AN = mk('omegaconf.dictconfig', 'DictConfig'), mk('omegaconf.listconfig', 'ListConfig'), mk('omegaconf.nodes', 'AnyNode')

    def meta(cls, obj_type, key):
        m = cls()
        m.__dict__.update(ref_type=typing.Any, object_type=obj_type, optional=True, key=key, flags=None)
        if cls is CM:
            m.__dict__.update(key_type=typing.Any, element_type=typing.Any)
        return m

    def wrap(v, parent, key):
        if isinstance(v, dict):
            o = DC()
            o.__dict__.update(_metadata=meta(CM, dict, key), _parent=parent, _flags_cache=None, _content={})
            for k, x in v.items():
                o._content[k] = wrap(x, o, k)
            return o
        if isinstance(v, (list, tuple)):
            o = LC()
            o.__dict__.update(_metadata=meta(CM, list, key), _parent=parent, _flags_cache=None, _content=[])
            for i, x in enumerate(v):
                o._content.append(wrap(x, o, i))
            return o
        o = AN()
        o.__dict__.update(_val=v, _parent=parent, _metadata=meta(MD, None, key))
        return o
    try:
        ck = {'step': 8400, 'model': model.state_dict(), 'optimizer': opt.state_dict(), 'ema': ema.state_dict(), 'scaler': None,
              'config': wrap(cfg_dict, None, None)}
        torch.save(ck, path)
    finally:
        for mn in made:
            sys.modules.pop(mn, None)
        for mn in list(sys.modules):
            if mn.startswith('omegaconf'):
                sys.modules.pop(mn, None)


def test_reference_layout_checkpoint_with_omegaconf_config_loads(tmp_path):
    """SURVEY 8f N2, the reference's own files: RD/utils.py:85 stores the omegaconf config object.  It loads with
    torch.load(weights_only=True) and inert stand-ins for the omegaconf classes (rdmi/utils.py) -- omegaconf absent, nothing of the
    file executed -- into model / optimizer / EMA in place, and the config comes back as plain dicts; a pickle that refers to any
    other foreign class is still refused."""
    import sys
    import __graft_entry__ as ge
    from rdmi import losses, utils
    from rdmi.models import utils as mutils
    from rdmi.models.ema import ExponentialMovingAverage
    cfg = ge.demo_config()
    torch.manual_seed(3)
    model = mutils.create_model(cfg)
    opt = losses.get_optimizer(cfg, model.parameters())
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    for p in model.parameters():
        p.grad = torch.full_like(p, 2e-3)
    opt.step(); ema.update(model.parameters())
    cfg_dict = utils._plain(cfg)
    path = str(tmp_path / 'checkpoint_8400.pth')
    _write_reference_layout_checkpoint(path, model, opt, ema, cfg_dict)
    assert not any(m.startswith('omegaconf') for m in sys.modules)
    names = torch.serialization.get_unsafe_globals_in_checkpoint(path)
    assert 'omegaconf.dictconfig.DictConfig' in names and 'omegaconf.nodes.AnyNode' in names
    with pytest.raises(Exception):
        torch.load(path, weights_only=True)                        # the plain safe loader refuses the file ...
    torch.manual_seed(4)
    model2 = mutils.create_model(cfg)
    opt2 = losses.get_optimizer(cfg, model2.parameters())
    ema2 = ExponentialMovingAverage(model2.parameters(), decay=cfg.model.ema_rate)
    st = utils.restore_checkpoint(path, dict(optimizer=opt2, model=model2, ema=ema2, step=0, scaler=None), 'cpu')      # ... rdmi loads it
    assert st['step'] == 8400
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), model2.state_dict().values()))
    assert all(torch.equal(a, b) for a, b in zip(ema.shadow_params, ema2.shadow_params)) and ema2.num_updates == ema.num_updates
    loaded = utils._load(path, 'cpu')
    assert loaded['config'] == cfg_dict and loaded['config']['model']['ch_mult'] == [1, 2, 2]
    assert not any(m.startswith('omegaconf') for m in sys.modules)          # nothing was imported to do it
    m3 = utils.load_denoising_model(path, mutils.create_model(cfg))
    assert torch.equal(m3.state_dict()['input_conv.weight'], model.state_dict()['input_conv.weight'])
    torch.save({'model': model.state_dict(), 'config': _NotPlain()}, str(tmp_path / 'other.pth'))
    with pytest.raises(RuntimeError, match='only omegaconf'):
        utils.load_denoising_model(str(tmp_path / 'other.pth'), m3)
