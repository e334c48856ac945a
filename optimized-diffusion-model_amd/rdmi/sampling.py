"""Sampling: predictor / corrector / denoiser registries, PC sampler, ODE sampler.

Host mirror of Reflected-Diffusion/sampling.py ("RD/sampling.py"): same registries and decorators, the same
abstract classes and `update_fn` contracts, `get_sampling_fn(config, sde, shape, eps, device)` returning
`sampling_fn(model, z=None, noise_removal_model=None, weight=0, class_labels=None) -> (x, nfe)`.

Two execution routes, both on librdmi (HIP):
  * FUSED: when the registered predictor/corrector classes are the built-in reflected Euler-Maruyama and
    Langevin/None and the model is the native NCSNpp, the whole N-1 update loop is one C call
    (rdmi_pc_sample): no per-step Python, no per-step allocation, in-kernel Philox noise (or injected noise).
  * GENERIC: any user-registered Predictor/Corrector runs through the reference's Python loop; each
    update_fn still lands in HIP kernels (score via rdmi_cf_score / rdmi_score, updates via rdmi_em_update,
    rdmi_langevin_update, rdmi_reflect).
"""
import abc

import numpy as np
import torch

from . import _native, cube
from .models import utils as mutils
from .models.utils import from_flattened_numpy, to_flattened_numpy, get_score_fn  # noqa: F401 (reference re-exports)

_CORRECTORS = {}
_PREDICTORS = {}
_DENOISERS = {}


def _make_register(table):
    def register(cls=None, *, name=None):
        def _register(klass):
            key = klass.__name__ if name is None else name
            if key in table:
                raise ValueError(f'Already registered model with name: {key}')
            table[key] = klass
            return klass

        return _register if cls is None else _register(cls)

    return register


register_predictor = _make_register(_PREDICTORS)    # RD/sampling.py:18-35
register_corrector = _make_register(_CORRECTORS)    # RD/sampling.py:38-55
register_denoiser = _make_register(_DENOISERS)      # RD/sampling.py:57-73


def get_predictor(name):
    return _PREDICTORS[name]


def get_corrector(name):
    return _CORRECTORS[name]


def get_denoiser(name):
    return _DENOISERS[name]


def get_sampling_fn(config, sde, shape, eps, device):
    """RD/sampling.py:87-130: dispatch on config.sampling.method ('pc' | 'ode')."""
    sampler_name = config.sampling.method
    if sampler_name.lower() == 'ode':
        return get_ode_sampler(sde=sde, shape=shape, eps=eps, moll=config.sampling.moll,
                               side_eps=config.sampling.side_eps, device=device)
    if sampler_name.lower() == 'pc':
        return get_pc_sampler(sde=sde, shape=shape,
                              predictor=get_predictor(config.sampling.predictor.lower()),
                              corrector=get_corrector(config.sampling.corrector.lower()),
                              denoiser=get_denoiser(config.sampling.denoiser.lower()),
                              snr=config.sampling.snr, n_steps=config.sampling.n_steps_each, eps=eps, device=device)
    raise ValueError(f'Sampler name {sampler_name} unknown.')


class Predictor(abc.ABC):
    """RD/sampling.py:133-155."""

    def __init__(self, sde, score_fn, probability_flow=False):
        super().__init__()
        self.sde = sde
        self.rsde = sde.reverse(score_fn, probability_flow)
        self.score_fn = score_fn

    @abc.abstractmethod
    def update_fn(self, x, t):
        """-> (x, x_mean)"""


class Corrector(abc.ABC):
    """RD/sampling.py:158-180."""

    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__()
        self.sde, self.score_fn, self.snr, self.n_steps = sde, score_fn, snr, n_steps

    @abc.abstractmethod
    def update_fn(self, x, t):
        """-> (x, x_mean)"""


class Denoiser(abc.ABC):
    """RD/sampling.py:182-190."""

    def __init__(self, denoiser):
        super().__init__()
        self.denoiser = denoiser

    @abc.abstractmethod
    def update_fn(self, x, x_mean, t):
        pass


@register_predictor(name='euler_maruyama')
class ReflectedEulerMaruyamaPredictor(Predictor):
    """RD/sampling.py:193-207.  x_mean = x - g(t)^2 score dt, x' = x_mean + g sqrt(-dt) z, dt = -1/N; both
    reflected.  The elementwise update is one HIP kernel (rdmi_em_update)."""

    def update_fn(self, x, t):
        z = torch.randn_like(x)
        score = self.score_fn(x, t)
        if hasattr(self.sde, 'sigma_min') and not self.rsde.probability_flow:
            return _native.em_update(x, score, z, t, self.rsde.N, float(self.sde.sigma_min), float(self.sde.sigma_max))
        # any other SDE: the reference's arithmetic on tensors, reflection in HIP
        dt = -1. / self.rsde.N
        drift, diffusion = self.sde.sde(x, t)
        drift = drift - diffusion[:, None, None, None] ** 2 * score * (0.5 if self.rsde.probability_flow else 1.)
        diffusion = torch.zeros_like(diffusion) if self.rsde.probability_flow else diffusion
        x_mean = x + drift * dt
        x = x_mean + diffusion[:, None, None, None] * np.sqrt(-dt) * z
        return cube.reflect(x), cube.reflect(x_mean)


@register_corrector(name='langevin')
class ReflectedLangevinCorrector(Corrector):
    """RD/sampling.py:210-233.  step = 2 (snr * mean_b||z_b|| / mean_b||s_b||)^2 -- one scalar for the
    batch it is given; x_mean = x + step s, x' = x_mean + sqrt(2 step) z; both reflected (rdmi_langevin_update)."""

    def update_fn(self, x, t):
        x_mean = x
        for _ in range(self.n_steps):
            grad = self.score_fn(x, t)
            noise = torch.randn_like(x)
            x, x_mean = _native.langevin_update(x, grad, noise, float(self.snr))
        return x, x_mean


@register_corrector(name='none')
class NoneCorrector(Corrector):
    """RD/sampling.py:236-241."""

    def update_fn(self, x, t):
        return x, x


@register_denoiser(name='network')
class TrainedDenoiser(Denoiser):
    """RD/sampling.py:244-248."""

    def update_fn(self, x, x_mean, t):
        return (x - self.denoiser(x, t)).clamp(min=0, max=1)


@register_denoiser(name='mean')
class MeanDenoiser(Denoiser):
    def update_fn(self, x, x_mean, t):
        return x_mean


@register_denoiser(name='none')
class NoneDenoiser(Denoiser):
    def update_fn(self, x, x_mean, t):
        return x


def _fusable(sde, model, predictor, corrector):
    native, _ = mutils._is_native(model)
    return (native and predictor is ReflectedEulerMaruyamaPredictor
            and corrector in (ReflectedLangevinCorrector, NoneCorrector) and hasattr(sde, 'sigma_min'))


def get_pc_sampler(sde, shape, predictor, corrector, denoiser, snr, n_steps=1, eps=1e-3, device='cuda',
                   noise=None, seed=None, seq_offset=0, trace=None, teacher=None, fused=True, shard=None):
    """RD/sampling.py:292-339.  Extra keyword-only knobs (not in the reference) for parity testing and sharding:
    noise: 'torch' (per-update torch.randn_like on `device`, the reference's RNG consumption), a tensor
    [(N-1)*(n_corr+1), B, H*W] of injected draws, or None = in-kernel Philox keyed by (seed, seq_offset);
    trace / teacher: per-update recording / teacher forcing buffers [N-1, B, H*W]; fused=False forces the generic loop;
    shard=(rank, world): `shape` is this rank's slice of a global batch of world*shape[0]: the prior is drawn for the
    global batch and sliced and the Philox stream is offset, so the shards reproduce the unsharded run."""

    def draw_prior():
        if shard is None:
            return torch.rand(shape)
        r, w = shard
        return torch.rand((w * shape[0],) + tuple(shape[1:]))[r * shape[0]:(r + 1) * shape[0]].contiguous()

    off = seq_offset if shard is None else seq_offset + shard[0] * shape[0]

    def pc_sampler(model, z=None, noise_removal_model=None, weight=0, class_labels=None):
        # F5: the reference draws the prior, builds the score fn, then draws the prior AGAIN inside
        # no_grad and ignores z; the denoiser's result is discarded and the noisy x is returned.
        x = draw_prior().to(device) if z is None else z
        if class_labels is None:
            score_fn = mutils.get_score_fn(sde, model, train=False)
        else:
            score_fn = mutils.get_cf_score_fn(sde, model, class_labels, weight)
        pred = predictor(sde, score_fn)
        corr = corrector(sde, score_fn, snr, n_steps)
        deno = denoiser(noise_removal_model)

        with torch.no_grad():
            x = draw_prior().to(device)
            if fused and _fusable(sde, model, predictor, corrector) and not isinstance(noise, str):
                x = _fused_pc(sde, model, x, class_labels, weight, corrector is ReflectedLangevinCorrector, snr, n_steps,
                              eps, noise, seed, off, trace, teacher)
            else:
                timesteps = torch.linspace(sde.T, eps, sde.N, device=device)
                x_mean = x
                for i in range(sde.N):
                    t = timesteps[i]
                    vec_t = torch.ones(shape[0], device=t.device) * t
                    if i < sde.N - 1:
                        x, _ = corr.update_fn(x, vec_t)
                        x, x_mean = pred.update_fn(x, vec_t)
                vec_t = torch.ones(shape[0], device=x.device) * eps
                deno.update_fn(x, x_mean, vec_t)
            return x, sde.N * (n_steps + 1)

    return pc_sampler


def _fused_pc(sde, model, x, class_labels, weight, langevin, snr, n_steps, eps, noise, seed, seq_offset, trace, teacher):
    _, inner = mutils._is_native(model)
    model.eval()
    x = inner._prep(x).clone()
    B = x.shape[0]
    use_cfg = class_labels is not None
    ctx = inner.native_context(2 * B if use_cfg else B, x.shape[2], x.shape[3], x.device)
    o = _native.PcOpts()
    o.N, o.eps = int(sde.N), float(eps)
    o.sigma_min, o.sigma_max = float(sde.sigma_min), float(sde.sigma_max)
    o.snr, o.n_steps_each = float(snr), int(n_steps)
    o.corrector, o.use_cfg = int(bool(langevin)), int(use_cfg)
    if seed is None:
        # default: derive the Philox key from the torch generator so torch.manual_seed() controls the run
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    o.seed, o.seq_offset = int(seed), int(seq_offset)
    lab = class_labels.to(x.device).contiguous().float() if use_cfg else None
    w = mutils._weight_tensor(weight, B, x.device).reshape(-1).contiguous().float() if use_cfg else None
    if noise is not None:
        per = (n_steps if langevin else 0) + 1
        assert noise.numel() == (sde.N - 1) * per * x.numel(), 'noise must hold (N-1)*(n_corr+1) draws of x.shape'
        noise = noise.to(x.device).contiguous().float()
    ctx.pc_sample(x, lab, w, noise, trace, teacher, o)
    return x


def get_ode_sampler(sde, shape, rtol=1e-5, atol=1e-5, method='RK45', eps=1e-3, moll=200, side_eps=1e-2, device='cuda',
                    fused=True, first_step=0.0, max_steps=0):
    """Probability-flow ODE sampler (RD/sampling.py:342-392): solve_ivp(method='RK45') over drift_fn * bump from sde.T to eps.

    Two routes:
      * DEVICE (native NCSNpp, RVESDE, method RK45): rdmi_ode_sample -- scipy's adaptive Dormand-Prince 5(4) with the float64
        state, the stage combinations, the error norm and the score network all on the GPU; the host keeps the scalar step
        controller.  Same (x, nfev) contract as the reference.
      * GENERIC (any other model / SDE / method, or fused=False): the reference's scipy loop, each right-hand side one score call.
    Extra keywords (not in the reference) are parity-test hooks: first_step (solve_ivp's first_step), max_steps."""
    from scipy import integrate

    def drift_fn(score_fn, x, t):
        rsde = sde.reverse(score_fn, probability_flow=True)
        return rsde.sde(x, t)[0]

    def ode_sampler(model, z=None, noise_removal_model=None, weight=0, class_labels=None):
        with torch.no_grad():
            x = (1 - 2 * side_eps) * torch.rand(shape).to(device) + side_eps if z is None else z
            native, inner = mutils._is_native(model)
            if fused and native and hasattr(sde, 'sigma_min') and method == 'RK45':
                model.eval()
                x = inner._prep(x).clone()
                B = x.shape[0]
                use_cfg = class_labels is not None
                ctx = inner.native_context(2 * B if use_cfg else B, x.shape[2], x.shape[3], x.device)
                o = _native.OdeOpts()
                o.T, o.eps, o.rtol, o.atol = float(sde.T), float(eps), float(rtol), float(atol)
                o.sigma_min, o.sigma_max, o.moll = float(sde.sigma_min), float(sde.sigma_max), float(moll)
                o.use_cfg, o.first_step, o.max_steps = int(use_cfg), float(first_step), int(max_steps)
                lab = class_labels.to(x.device).contiguous().float() if use_cfg else None
                w = mutils._weight_tensor(weight, B, x.device).reshape(-1).contiguous().float() if use_cfg else None
                nfev, t_end, h_next = ctx.ode_sample(x, lab, w, o)
                ode_sampler.last = dict(t=t_end, h_next=h_next)          # where the controller stopped (max_steps hook)
                return x, nfev
            if class_labels is None:
                score_fn = mutils.get_score_fn(sde, model, train=False)
            else:
                score_fn = mutils.get_cf_score_fn(sde, model, class_labels, weight)

            def bump(v):
                return ((-1 / (0.5 ** 2 - (0.5 - v).pow(2)) + 4) / moll).exp() if moll > 0 else v

            def ode_func(t, xv):
                xt = from_flattened_numpy(xv, shape).to(device).type(torch.float32)
                vec_t = torch.ones(shape[0], device=xt.device) * t
                return to_flattened_numpy(drift_fn(score_fn, xt, vec_t) * bump(xt))

            sol = integrate.solve_ivp(ode_func, (sde.T, eps), to_flattened_numpy(x), rtol=rtol, atol=atol, method=method)
            x = torch.tensor(sol.y[:, -1]).reshape(shape).to(device).type(torch.float32)
            return x, sol.nfev

    return ode_sampler
