"""torch.autograd glue for the HIP training step: NCSNpp forward/backward and the score-matching loss.

No arithmetic here: forward = rdmi_train_forward (layer plan, activations kept, Dropout_0 by in-kernel Philox),
backward = rdmi_backward (every parameter gradient into one flat buffer in the reference's parameter order, handed
to autograd as views).  The dropout seed of a step is drawn from the torch generator, so torch.manual_seed controls it.
"""
import torch

from . import _native


class _NCSNppFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, sigma, labels, *params):
        object.__setattr__(model, '_plist_now', list(params))          # this call's parameter walk, reused by the binding check
        try:
            tctx = model.train_context(x.shape[0], x.shape[2], x.shape[3], x.device)
        finally:
            object.__setattr__(model, '_plist_now', None)
        ctx.plist = params
        out = torch.empty_like(x)
        p = float(model.dropout) if model.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0
        tctx.train_forward(x, sigma, labels, out, p, seed)
        ctx.model, ctx.tctx = model, tctx
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, gout):
        (x,) = ctx.saved_tensors
        model = ctx.model
        plist = ctx.plist
        total = getattr(model, '_n_param_elems', None)
        if total is None:
            total = model._n_param_elems = sum(p.numel() for p in plist)
        flat = torch.empty(total, dtype=torch.float32, device=x.device)
        ctx.tctx.backward(gout.contiguous().float(), flat, x)
        # one C++ call makes the 260 views (fresh tensor objects: autograd adopts them as .grad without a copy)
        views = torch._utils._unflatten_dense_tensors(flat, plist)
        grads = [v if p.requires_grad else None for v, p in zip(views, plist)]
        return (None, None, None, None, *grads)


def ncsnpp_apply(model, x, time_cond, class_labels):
    x = model._prep(x)
    if model.conditional and class_labels is None:
        raise RuntimeError('class_labels is required: the model is conditional (label_emb)')
    lab = None if class_labels is None else class_labels.contiguous().float()
    return _NCSNppFn.apply(model, x, time_cond.contiguous().float(), lab, *model.parameters())


class _SmLossFn(torch.autograd.Function):
    """per_sample = reduce(w * (score - score_hk)^2) with d per_sample / d score saved by the forward kernel."""

    @staticmethod
    def forward(ctx, score, perturbed, batch, t, smin, smax, likelihood_weighting, reduce_mean):
        per, dscore = _native.sm_loss(score, perturbed, batch, t, smin, smax, likelihood_weighting, reduce_mean, want_grad=True)
        ctx.save_for_backward(dscore)
        return per

    @staticmethod
    def backward(ctx, gper):
        (dscore,) = ctx.saved_tensors
        g = gper.reshape(-1, *([1] * (dscore.dim() - 1))) * dscore
        return g, None, None, None, None, None, None, None


def sm_loss(score, perturbed, batch, t, smin, smax, likelihood_weighting, reduce_mean):
    return _SmLossFn.apply(score, perturbed, batch, t, smin, smax, likelihood_weighting, reduce_mean)
