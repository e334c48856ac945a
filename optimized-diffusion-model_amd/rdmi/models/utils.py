"""Model registry and score-function adapters.

Host mirror of Reflected-Diffusion/models/utils.py ("RD/models/utils.py"): same names, arguments,
return values and error behaviour; the arithmetic is done by librdmi (HIP) when the model is the native
NCSNpp, and by calling the model object otherwise (user-registered models keep working).
"""
import numpy as np
import torch

_MODELS = {}


def register_model(cls=None, *, name=None):
    """Class decorator `@register_model(name='x')` / `@register_model` (RD/models/utils.py:11-28).
    Registering a name twice raises ValueError, like the reference."""

    def _register(klass):
        key = klass.__name__ if name is None else name
        if key in _MODELS:
            raise ValueError(f'Already registered model with name: {key}')
        _MODELS[key] = klass
        return klass

    return _register if cls is None else _register(cls)


def get_model(name):
    """RD/models/utils.py:31-32 (KeyError on unknown names)."""
    return _MODELS[name]


def get_sigmas(config):
    """Noise levels for SMLD, RD/models/utils.py:35-45."""
    return np.exp(np.linspace(np.log(config.sde.sigma_max), np.log(config.sde.sigma_min), config.sde.num_scales))


def create_model(config):
    """RD/models/utils.py:48-52."""
    return get_model(config.model.name)(config)


def _is_native(model):
    inner = model.module if hasattr(model, 'module') else model     # DDP unwrap, as RD/utils.py:58 probes
    return getattr(inner, 'rdmi_native', False), inner


def get_model_fn(model, train=False):
    """RD/models/utils.py:55-84: returns model_fn(x, time_cond, class_labels=None); sets train()/eval()
    on every call as a side effect."""

    def model_fn(x, time_cond, class_labels=None):
        if model.training != bool(train):          # (Module.train() walks every submodule: only when the mode really changes)
            model.train() if train else model.eval()
        return model(x, time_cond, class_labels=class_labels)

    return model_fn


def get_score_fn(sde, model, train=False):
    """RD/models/utils.py:87-105: score_fn(x, t, class_labels=None) = model(x, sigma(t), labels).
    For the native NCSNpp in eval mode sigma(t) is evaluated inside the HIP embedding kernel
    (rdmi_score), so no `zeros_like(x)` / marginal_prob temporaries are created."""
    native, inner = _is_native(model)
    model_fn = get_model_fn(model, train=train)

    def score_fn(x, t, class_labels=None):
        if native and not train and hasattr(sde, 'sigma_min') and not torch.is_grad_enabled():
            model.eval()
            return inner.native_score(x, t, class_labels, sde.sigma_min, sde.sigma_max)
        time_cond = sde.marginal_prob(torch.zeros_like(x), t)[1]
        return model_fn(x, time_cond, class_labels=class_labels)

    return score_fn


def _weight_tensor(weight, B, device):
    """None -> zeros, python scalar -> full, tensor -> as is (RD/models/utils.py:130-136)."""
    if weight is None:
        return torch.zeros(B, device=device)
    if isinstance(weight, (float, int)):
        return torch.full((B,), float(weight), device=device)
    return weight


def get_cf_score_fn(sde, model, class_labels, weight):
    """Classifier-free-guidance score, RD/models/utils.py:108-140:
    one forward at 2B on [x;x], [t;t], [labels;0], then (1+w)*s_cond - w*s_uncond."""
    native, inner = _is_native(model)
    score_fn = get_score_fn(sde, model, train=False)

    def weighted_score_fn(x, t):
        B = x.shape[0]
        w = _weight_tensor(weight, B, x.device)
        if native and hasattr(sde, 'sigma_min') and not torch.is_grad_enabled():
            model.eval()
            return inner.native_cf_score(x, t, class_labels, w.reshape(-1), sde.sigma_min, sde.sigma_max)
        xx = x.repeat(2, 1, 1, 1)
        tt = t.repeat(2)
        ll = torch.cat([class_labels, torch.zeros_like(class_labels)], dim=0)
        s = score_fn(xx, tt, ll)
        w = w.view(-1, 1, 1, 1)
        return (1 + w) * s[:B] - w * s[B:]

    return weighted_score_fn


def to_flattened_numpy(x):
    """RD/models/utils.py:143-145."""
    return x.detach().cpu().numpy().reshape((-1,))


def from_flattened_numpy(x, shape):
    """RD/models/utils.py:148-150."""
    return torch.from_numpy(x.reshape(shape))
