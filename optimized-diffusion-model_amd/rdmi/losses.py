"""Loss, optimizer and step functions.

Host mirror of Reflected-Diffusion/losses.py ("RD/losses.py"): get_optimizer, optimization_manager,
get_sde_loss_fn, get_step_fn with the reference's signatures and `state` dict contract
(state = dict(optimizer, model, ema, step, scaler), RD/run_train.py:224-225).

What runs where (round 1):
  * the loss arithmetic -- perturbation + reflection, the reflected-heat-kernel target (cube.score_hk) and the
    weighted squared error reduction -- is two HIP kernels (rdmi_perturb, rdmi_sm_loss);
  * evaluation step: the fused HIP U-Net forward under no_grad with the EMA weights swapped in;
  * training step: train-mode forward (Dropout_0 by in-kernel Philox, label drop) and the NCSN++ backward are HIP
    kernels on the layer plan (csrc/train_plan.h, bwd_kernels.h) behind a torch.autograd.Function; Adam, gradient
    clipping and the EMA update are the reference's own torch calls on the parameter tensors.
The reference's per-call NaN hooks (RD/losses.py:95-104) are deliberately not reproduced: they leak one hook per
parameter per call and slow training from 0.5 s to 38 s per step.
"""
import numpy as np
import torch
import torch.optim as optim

from . import _native
from .models import utils as mutils


_OPTIMIZERS = {'Adam': optim.Adam, 'AdamW': optim.AdamW}


def get_optimizer(config, params):
    """RD/losses.py:12-23: Adam / AdamW from config.optim.{lr, beta1, beta2, eps, weight_decay}."""
    o = config.optim
    cls = _OPTIMIZERS.get(o.optimizer)
    if cls is None:
        raise NotImplementedError(f'Optimizer {o.optimizer} not supported yet!')
    return cls(params, lr=o.lr, betas=(o.beta1, o.beta2), eps=o.eps, weight_decay=o.weight_decay)


def optimization_manager(config):
    """RD/losses.py:26-49.  Returns optimize_fn(optimizer, params, step, lr, warmup, grad_clip, scaler): linear learning-rate
    warm-up over `warmup` steps, global-norm clipping at `grad_clip` (negative: off), then the optimizer step -- through the
    GradScaler when one is given (its unscale_ comes first so the clip sees true gradients)."""
    oc = config.optim

    def optimize_fn(optimizer, params, step, lr=oc.lr, warmup=oc.warmup, grad_clip=oc.grad_clip, scaler=None):
        scaled = scaler is not None
        if scaled:
            scaler.unscale_(optimizer)
        if warmup > 0:
            ramp = lr * np.minimum(step / warmup, 1.0)
            for group in optimizer.param_groups:
                group['lr'] = ramp
        if grad_clip >= 0:
            torch.nn.utils.clip_grad_norm_(params, max_norm=grad_clip)
        if scaled:
            scaler.step(optimizer)
            scaler.update()
        else:
            optimizer.step()

    return optimize_fn


def get_sde_loss_fn(sde, train, reduce_mean=True, likelihood_weighting=True, eps=1e-5):
    """RD/losses.py:52-107.  loss_fn(model, batch, class_labels=None) -> scalar tensor."""

    def loss_fn(model, batch, class_labels=None):
        score_fn = mutils.get_score_fn(sde, model, train=train)
        t = torch.rand(batch.shape[0], device=batch.device) * (sde.T - eps) + eps
        z = torch.randn_like(batch)
        smin, smax = float(sde.sigma_min), float(sde.sigma_max)
        perturbed = _native.perturb(batch.float(), z, t, smin, smax)                  # reflect(mean + std z)
        score = score_fn(perturbed, t, class_labels=class_labels)
        if score.requires_grad:
            from . import autograd_fn
            per = autograd_fn.sm_loss(score, perturbed, batch.float(), t, smin, smax, likelihood_weighting, reduce_mean)
        else:
            per = _native.sm_loss(score, perturbed, batch.float(), t, smin, smax, likelihood_weighting, reduce_mean)
        return torch.mean(per)

    return loss_fn


def get_step_fn(sde, train, optimize_fn=None, reduce_mean=False, likelihood_weighting=False):
    """RD/losses.py:110-160.  step_fn(state, batch, class_labels=None) -> loss."""
    loss_fn = get_sde_loss_fn(sde, train, reduce_mean=reduce_mean, likelihood_weighting=likelihood_weighting)

    def step_fn(state, batch, class_labels=None):
        model = state['model']
        if train:
            optimizer = state['optimizer']
            optimizer.zero_grad()
            loss = loss_fn(model, batch, class_labels=class_labels)
            if state['scaler'] is None:
                loss.backward()
            else:                                   # RD/losses.py:143-146 (the HIP backward is linear in the incoming gradient)
                state['scaler'].scale(loss).backward()
            optimize_fn(optimizer, model.parameters(), step=state['step'], scaler=state['scaler'])
            state['step'] += 1
            state['ema'].update(model.parameters())
            return loss
        with torch.no_grad():
            ema = state['ema']
            ema.store(model.parameters())
            ema.copy_to(model.parameters())
            loss = loss_fn(model, batch, class_labels=class_labels)
            ema.restore(model.parameters())
        return loss

    return step_fn
