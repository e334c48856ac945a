"""Loss, optimizer and step functions.

Host mirror of Reflected-Diffusion/losses.py ("RD/losses.py"): get_optimizer, optimization_manager,
get_sde_loss_fn, get_step_fn with the reference's signatures and `state` dict contract
(state = dict(optimizer, model, ema, step, scaler), RD/run_train.py:224-225).

What runs where (round 1):
  * the loss arithmetic -- perturbation + reflection, the reflected-heat-kernel target (cube.score_hk) and the
    weighted squared error reduction -- is two HIP kernels (rdmi_perturb, rdmi_sm_loss);
  * evaluation step: the fused HIP U-Net forward under no_grad with the EMA weights swapped in;
  * training step: train-mode forward (Dropout_0 by in-kernel Philox, label drop) and the NCSN++ backward are HIP
    kernels on the layer plan (csrc/train_plan.h, bwd_kernels.h) behind a torch.autograd.Function; gradient clipping,
    Adam / AdamW and the EMA update run as ONE multi-tensor HIP step over all 260 tensors (rdmi_opt_step: three
    launches instead of ~1000 torch launches) when get_optimizer's optimizer is driven by optimization_manager's
    optimize_fn; the optimizer object stays a torch.optim.Adam subclass with the standard state_dict layout.
The reference's per-call NaN hooks (RD/losses.py:95-104) are deliberately not reproduced: they leak one hook per
parameter per call and slow training from 0.5 s to 38 s per step.
"""
import inspect

import numpy as np
import torch
import torch.optim as optim

from . import _native
from .models import utils as mutils


class _FusedStep:
    """Mixin for torch.optim.Adam / AdamW: `fused_step(grad_clip, ema)` = clip_grad_norm_ + step() [+ ema.update()] as one
    multi-tensor HIP step (csrc/opt_kernels.h).  State lives where torch keeps it (state[p] = {step, exp_avg, exp_avg_sq}),
    so state_dict() / load_state_dict() and checkpoints are those of torch.optim.Adam; plain .step() still works (torch's)."""
    _decoupled = False

    def _fusable(self):
        for g in self.param_groups:
            if g.get('amsgrad') or g.get('maximize') or g.get('capturable') or g.get('differentiable'):
                return False
            for p in g['params']:
                if p.dtype != torch.float32 or not (p.is_cuda or _native.is_emulator()) or (p.grad is not None and p.grad.is_sparse):
                    return False
        return len(self.param_groups) == 1

    # ---- steady-state bookkeeping.  After one full (slow) pass the lists below are kept: a step then costs one sweep of data_ptr()
    #      calls to prove nothing moved, instead of rebuilding five lists of 260 tensors and a device table.
    _fast = None
    _pending_steps = 0            # per-parameter state['step'] tensors are advanced lazily (state_dict / load_state_dict / step flush them)

    def _flush_steps(self):
        if self._pending_steps and self._fast is not None:
            torch._foreach_add_([self.state[p]['step'] for p in self._fast['params']], float(self._pending_steps))
        self._pending_steps = 0

    def state_dict(self, *a, **k):
        self._flush_steps()
        return super().state_dict(*a, **k)

    def load_state_dict(self, *a, **k):
        self._flush_steps()
        self._fast = None
        return super().load_state_dict(*a, **k)

    def step(self, *a, **k):
        self._flush_steps()
        self._fast = None
        return super().step(*a, **k)

    def add_param_group(self, *a, **k):
        self._fast = None
        return super().add_param_group(*a, **k)

    def fused_step(self, grad_clip=-1.0, ema=None):
        """Returns True when the EMA update was done here too (the caller then skips ema.update)."""
        group = self.param_groups[0]
        f = self._fast
        if f is not None and f['group_params'] is group['params'] and (ema is f['ema'] or f['ema'] is None and ema is None):
            params = f['params']
            grads = [p.grad for p in params]
            # (`None not in grads` would call Tensor.__eq__ 260 times: 2.6 ms)
            if (all(g is not None for g in grads) and [p.data_ptr() for p in params] == f['pptr']
                    and (f['ema'] is None or (ema.shadow_params is f['shadow_list'] and ema.shadow_params[0].data_ptr() == f['e0']))
                    and self.state[params[0]]['exp_avg'].data_ptr() == f['m0']
                    and sum(1 for q in group['params'] if q.grad is not None) == f['n']):
                gptr = [g.data_ptr() for g in grads]
                if gptr != f['gptr']:
                    if not all(g.is_contiguous() and g.dtype == torch.float32 for g in grads):
                        f = None
                    else:
                        f['plan'].update_grads(grads)
                        f['gptr'] = gptr
                if f is not None:
                    return self._launch(f['plan'], group, f['t0'] + self._pending_steps + 1, grad_clip, ema if f['ema'] is not None else None, counted=True)
        self._flush_steps()
        self._fast = None
        params = [p for p in group['params'] if p.grad is not None]
        if not params:
            return False
        req = [q for q in group['params'] if q.requires_grad]
        if ema is not None:                                   # shadow list is over requires_grad parameters, in order
            shadow = {id(p): s for p, s in zip(req, ema.shadow_params)}
            # the EMA is folded in only when this step touches EVERY shadowed parameter (ema.update relaxes all of them,
            # RD/models/ema.py:32-52); otherwise the caller runs ema.update itself
            if len(ema.shadow_params) != len(req) or len(params) != len(req) or any(id(p) not in shadow for p in params):
                ema = None
        for p in params:                                       # lazy state init, as torch.optim.Adam._init_group
            st = self.state[p]
            if len(st) == 0:
                st['step'] = torch.tensor(0.0, dtype=torch.float32)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in params]
        for p, g in zip(params, grads):
            if g is not p.grad:
                p.grad = g
        ms = [self.state[p]['exp_avg'] for p in params]
        vs = [self.state[p]['exp_avg_sq'] for p in params]
        es = [shadow[id(p)] for p in params] if ema is not None else None
        ptrs = tuple(t.data_ptr() for ts in ([p.data for p in params], grads, ms, vs, es or ()) for t in ts)
        plan = getattr(self, '_plan', None)
        if plan is None or plan.ptrs != ptrs:
            if plan is not None:
                plan.close()
            plan = self._plan = _native.OptPlan([p.data for p in params], grads, ms, vs, es, params[0].device)
        t0 = int(self.state[params[0]]['step'])
        if any(int(self.state[p]['step']) != t0 for p in params):
            raise RuntimeError('fused_step: parameters are at different optimizer steps (one bias correction serves all tensors); '
                               'use optimizer.step() for this state')
        self._fast = dict(group_params=group['params'], params=params, n=len(params), nreq_grad=len(params), ema=ema,
                          shadow_list=ema.shadow_params if ema is not None else None, e0=es[0].data_ptr() if es else 0,
                          pptr=[p.data_ptr() for p in params], gptr=[g.data_ptr() for g in grads], m0=ms[0].data_ptr(), plan=plan, t0=t0)
        return self._launch(plan, group, t0 + 1, grad_clip, ema, counted=True)

    def _launch(self, plan, group, t, grad_clip, ema, counted):
        h = _native.OptHyper()
        b1, b2 = group['betas']
        h.lr, h.beta1, h.beta2, h.eps, h.weight_decay = group['lr'], b1, b2, group['eps'], group['weight_decay']
        h.lr_d, h.beta1_d, h.beta2_d = float(group['lr']), float(b1), float(b2)
        h.decoupled_wd, h.step, h.max_norm, h.write_back_grad = int(self._decoupled), t, float(grad_clip), 1
        if ema is not None:
            h.ema_decay_d = float(ema.next_decay())
        plan.step(h)
        # every parameter keeps its OWN step tensor (torch's layout: a checkpoint preserves aliasing, and a torch.optim.Adam that loaded
        # shared step tensors would advance them once per parameter); they are advanced lazily, see _flush_steps
        self._pending_steps += 1
        return ema is not None


class FusedAdam(_FusedStep, optim.Adam):
    pass


class FusedAdamW(_FusedStep, optim.AdamW):
    _decoupled = True


_OPTIMIZERS = {'Adam': FusedAdam, 'AdamW': FusedAdamW}


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

    def optimize_fn(optimizer, params, step, lr=oc.lr, warmup=oc.warmup, grad_clip=oc.grad_clip, scaler=None, ema=None):
        """`ema` (not in the reference's signature): when given and the whole update can run as the fused HIP step, the EMA
        update is folded into it and True is returned -- the caller then skips ema.update()."""
        scaled = scaler is not None
        if not scaled and isinstance(optimizer, _FusedStep) and optimizer._fusable():
            if warmup > 0:
                ramp = lr * np.minimum(step / warmup, 1.0)
                for group in optimizer.param_groups:
                    group['lr'] = ramp
            return optimizer.fused_step(grad_clip=grad_clip, ema=ema)
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
    takes_ema = False                               # decided ONCE from the signature: an exception raised inside optimize_fn is never retried
    if optimize_fn is not None:
        try:
            ps = inspect.signature(optimize_fn).parameters
            takes_ema = 'ema' in ps or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in ps.values())
        except (TypeError, ValueError):
            takes_ema = False

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
            if takes_ema:
                ema_done = optimize_fn(optimizer, model.parameters(), step=state['step'], scaler=state['scaler'], ema=state['ema'])
            else:                                   # a user-supplied optimize_fn with the reference's exact signature
                ema_done = optimize_fn(optimizer, model.parameters(), step=state['step'], scaler=state['scaler'])
            state['step'] += 1
            if not ema_done:
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
