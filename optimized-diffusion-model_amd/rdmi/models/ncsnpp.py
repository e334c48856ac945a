"""NCSN++ score network: a PARAMETER SHELL over librdmi.

Host mirror of Reflected-Diffusion/models/ncsnpp.py ("RD/models/ncsnpp.py") NCSNpp (registered as
'ncsnpp'): same constructor config keys, the same 261 state-dict names / shapes / registration order
(EMA, Adam and checkpoints depend on it), the same initialisers drawn in the same order from the torch
RNG (so `torch.manual_seed(s); create_model(cfg)` gives bit-identical parameters), and
`forward(x, time_cond, class_labels=None)`.

No layer arithmetic happens here: the leaf modules only own storage; forward() hands the parameter
pointers to the HIP launch plan (csrc/rdmi.hip), which runs the whole U-Net in fused MFMA kernels.
"""
import numpy as np
import torch
import torch.nn as nn

from . import utils
from .. import _native


# ------------------------------------------------------------------ initialisers (RD/models/layers.py:39-76)
def _fan_avg_uniform_(shape, scale):
    """variance_scaling(scale, 'fan_avg', 'uniform') with in_axis=1, out_axis=0; scale 0 means 1e-10."""
    scale = 1e-10 if scale == 0 else scale
    receptive = np.prod(shape) / shape[1] / shape[0]
    fan_in, fan_out = shape[1] * receptive, shape[0] * receptive
    variance = scale / ((fan_in + fan_out) / 2)
    return (torch.rand(*shape, dtype=torch.float32, device='cpu') * 2. - 1.) * np.sqrt(3 * variance)


def _conv3x3(cin, cout, stride=1, padding=1, init_scale=1.):
    """Storage of ddpm_conv3x3 (RD/models/layers.py:103-109): OIHW weight, zero bias.  nn.Conv2d is
    constructed first because the reference does so (its default init consumes RNG draws)."""
    m = nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=padding, bias=True)
    m.weight.data = _fan_avg_uniform_(m.weight.data.shape, init_scale)
    nn.init.zeros_(m.bias)
    return m


class NIN(nn.Module):
    """Storage of RD/models/layers.py:531-540: W [in, out], b [out]."""

    def __init__(self, in_dim, num_units, init_scale=0.1):
        super().__init__()
        self.W = nn.Parameter(_fan_avg_uniform_((in_dim, num_units), init_scale), requires_grad=True)
        self.b = nn.Parameter(torch.zeros(num_units), requires_grad=True)


class GaussianFourierProjection(nn.Module):
    """Storage of RD/models/layerspp.py:19-28: fixed W ~ N(0, scale^2), part of the checkpoint, not trained."""

    def __init__(self, embedding_size=256, scale=1.0):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embedding_size) * scale, requires_grad=False)


def _gn(ch):
    return nn.GroupNorm(num_groups=min(ch // 4, 32), num_channels=ch, eps=1e-6)


class ResnetBlockDDPMpp(nn.Module):
    """Storage of RD/models/layerspp.py:171-197."""

    def __init__(self, in_ch, out_ch, temb_dim, dropout, init_scale):
        super().__init__()
        self.GroupNorm_0 = _gn(in_ch)
        self.Conv_0 = _conv3x3(in_ch, out_ch)
        self.Dense_0 = nn.Linear(temb_dim, out_ch)
        self.Dense_0.weight.data = _fan_avg_uniform_(self.Dense_0.weight.data.shape, 1.)
        nn.init.zeros_(self.Dense_0.bias)
        self.GroupNorm_1 = _gn(out_ch)
        self.Dropout_0 = nn.Dropout(dropout)
        self.Conv_1 = _conv3x3(out_ch, out_ch, init_scale=init_scale)
        if in_ch != out_ch:
            self.NIN_0 = NIN(in_ch, out_ch)


class AttnBlockpp(nn.Module):
    """Storage of RD/models/layerspp.py:70-78."""

    def __init__(self, channels, init_scale):
        super().__init__()
        self.GroupNorm_0 = _gn(channels)
        self.NIN_0 = NIN(channels, channels)
        self.NIN_1 = NIN(channels, channels)
        self.NIN_2 = NIN(channels, channels)
        self.NIN_3 = NIN(channels, channels, init_scale=init_scale)


class _Resample(nn.Module):
    """Storage of Upsample / Downsample with_conv=True, fir=False (RD/models/layerspp.py:99-168)."""

    def __init__(self, ch, down):
        super().__init__()
        self.Conv_0 = _conv3x3(ch, ch, stride=2, padding=0) if down else _conv3x3(ch, ch)


@utils.register_model(name='ncsnpp')
class NCSNpp(nn.Module):
    rdmi_native = True

    def __init__(self, config):
        super().__init__()
        m = config.model
        self.config = config
        self.nf = nf = m.nf
        self.ch_mult = ch_mult = list(m.ch_mult)
        self.num_res_blocks = nrb = m.num_res_blocks
        self.attn_resolutions = list(m.attn_resolutions)
        self.dropout = m.dropout
        self.conditional = m.conditional
        self.cond_drop_prob = m.cond_drop_prob if hasattr(m, 'cond_drop_prob') else 0.0
        self.num_classes = getattr(m, 'num_classes', 1)
        self.init_scale = m.init_scale
        self.skip_rescale = m.skip_rescale
        self.image_size = m.image_size
        self.image_width = getattr(m, 'image_width', m.image_size)
        self.channels = m.channels
        self.scale_by_sigma = getattr(m, 'scale_by_sigma', False)
        # not a reference key: 'f32' (default, the reference's arithmetic) or 'bf16' (bf16 MFMA operands, fp32 accumulate; tiled plan)
        self.compute_dtype = str(getattr(m, 'compute_dtype', 'f32'))
        if self.compute_dtype not in ('f32', 'bf16'):
            raise NotImplementedError(f'compute_dtype {self.compute_dtype!r}: f32 or bf16')
        # 'bf16': the TRAINING step's contractions (forward, data gradient AND weight gradient) take bf16 MFMA operands from bf16 weight
        # copies, and the activations the forward keeps / the backward's scratch tensors are stored as bf16; fp32 master weights, fp32
        # accumulation, fp32 gradients and optimizer (BASELINE config #4).  Sampling / evaluation are not affected.
        self.train_dtype = str(getattr(m, 'train_dtype', 'f32'))
        if self.train_dtype not in ('f32', 'bf16'):
            raise NotImplementedError(f'train_dtype {self.train_dtype!r}: f32 or bf16')
        # what the HIP plan implements; anything else fails here, loudly, not in a fallback
        if m.embedding_type != 'fourier':
            raise NotImplementedError('Only fourier embedding supported')          # as RD/models/ncsnpp.py:100
        if m.nonlinearity.lower() != 'swish':
            raise NotImplementedError('librdmi builds the swish/SiLU activation only (all shipped configs use it)')
        if m.fir or not m.resamp_with_conv or not m.skip_rescale:
            raise NotImplementedError('librdmi builds fir=False, resamp_with_conv=True, skip_rescale=True '
                                      '(RD/configs/model/ncsnpp.yaml)')

        self.act = nn.SiLU()
        self.time_embed = GaussianFourierProjection(embedding_size=nf, scale=m.fourier_scale)
        self.time_mlp = nn.Sequential(nn.Linear(2 * nf, nf * 4), self.act, nn.Linear(nf * 4, nf * 4))
        if self.conditional:
            self.label_emb = nn.Linear(self.num_classes, nf * 4)
        self.input_conv = _conv3x3(self.channels, nf)

        self.down_blocks, self.down_attn, self.downsample = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        nlev = len(ch_mult)
        self.attn_levels = [self.image_size // (2 ** i) in self.attn_resolutions for i in range(nlev)]
        in_ch, skip = nf, []
        for i, mult in enumerate(ch_mult):
            for _ in range(nrb):
                self.down_blocks.append(ResnetBlockDDPMpp(in_ch, nf * mult, nf * 4, self.dropout, self.init_scale))
                in_ch = nf * mult
                self.down_attn.append(AttnBlockpp(in_ch, self.init_scale) if self.attn_levels[i] else None)
                skip.append(in_ch)
            skip.append(in_ch)
            self.downsample.append(_Resample(in_ch, down=True) if i != nlev - 1 else None)
        self.skip_channels = skip
        self.mid_block1 = ResnetBlockDDPMpp(in_ch, in_ch, nf * 4, self.dropout, self.init_scale)
        if self.attn_levels[-1]:
            raise NotImplementedError('attention at the bottleneck resolution (mid_attn) is not built')
        self.mid_attn = None
        self.mid_block2 = ResnetBlockDDPMpp(in_ch, in_ch, nf * 4, self.dropout, self.init_scale)

        self.up_blocks, self.up_attn, self.upsample = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        rskip = list(reversed(skip))
        for i in reversed(range(nlev)):
            for _ in range(nrb + 1):
                self.up_blocks.append(ResnetBlockDDPMpp(in_ch + rskip.pop(0), nf * ch_mult[i], nf * 4, self.dropout,
                                                        self.init_scale))
                in_ch = nf * ch_mult[i]
                self.up_attn.append(AttnBlockpp(in_ch, self.init_scale) if self.attn_levels[i] else None)
            self.upsample.append(_Resample(in_ch, down=False) if i != 0 else None)
        self.out_norm = _gn(in_ch)
        self.out_act = self.act
        self.out_conv = _conv3x3(in_ch, self.channels, init_scale=self.init_scale)
        self._ctx = {}

    # ------------------------------------------------------------------ native plumbing
    def _arch(self):
        a = _native.Arch()
        a.nf, a.n_levels, a.num_res_blocks = self.nf, len(self.ch_mult), self.num_res_blocks
        for i, v in enumerate(self.ch_mult):
            a.ch_mult[i] = v
        a.attn_levels = sum(1 << i for i, on in enumerate(self.attn_levels) if on)
        a.channels, a.num_classes = self.channels, self.num_classes
        a.conditional, a.scale_by_sigma = int(bool(self.conditional)), int(bool(self.scale_by_sigma))
        a.compute_dtype = 1 if self.compute_dtype == 'bf16' else 0
        return a

    def native_context(self, model_batch, H, W, device):
        """One rdmi_ctx per (device, H, W); grown when a larger model batch shows up."""
        device = torch.device(device)
        key = (str(device), H, W)
        ctx = self._ctx.get(key)
        if ctx is None or ctx.max_batch < model_batch or getattr(ctx, 'compute_dtype', self.compute_dtype) != self.compute_dtype:
            if ctx is not None:
                ctx.close()
            ctx = _native.Context(self._arch(), max(model_batch, 16), H, W, device)
            ctx.compute_dtype = self.compute_dtype      # a plan is built for ONE dtype: changing model.compute_dtype rebuilds it
            self._ctx[key] = ctx
        ctx.bind(self._named_tensors())
        return ctx

    def _named_tensors(self):
        """(name, tensor) of every parameter in state_dict order, cached: state_dict() walks all 184 modules (0.4 ms per call).  The
        tensors are looked up again whenever the module tree's parameter objects changed (load_state_dict copies in place and keeps
        them; .to() / ._apply() on a CPU->GPU move replaces .data, which bind() sees as a new pointer)."""
        cache = getattr(self, '_nt_cache', None)
        plist = self._plist_now if getattr(self, '_plist_now', None) is not None else list(self.parameters())
        if cache is None or len(cache[0]) != len(plist) or any(a is not b for a, b in zip(cache[0], plist)):
            cache = (plist, list(self.state_dict(keep_vars=True).items()))
            object.__setattr__(self, '_nt_cache', cache)
        return cache[1]

    def train_context(self, model_batch, H, W, device):
        """A dedicated rdmi_ctx for training (layer plan, per-tensor activation storage, gradient workspace)."""
        device = torch.device(device)
        key = ('train', str(device), H, W)
        ctx = self._ctx.get(key)
        if ctx is None or ctx.max_batch < model_batch or ctx.train_dtype != self.train_dtype:
            if ctx is not None:
                ctx.close()
            arch = self._arch()
            if self.train_dtype == 'bf16':
                arch.compute_dtype = 1
            ctx = _native.Context(arch, max(model_batch, 16), H, W, device)
            ctx.enable_training()
            ctx.train_dtype = self.train_dtype
            self._ctx[key] = ctx
        ctx.bind(self._named_tensors())
        return ctx

    def _prep(self, x):
        _native.require_device(x)
        p = next(self.parameters())
        if p.device != x.device:
            raise RuntimeError(f'model parameters are on {p.device} but the input is on {x.device}')
        if x.dim() != 4 or x.shape[1] != self.channels:
            raise ValueError(f'expected x of shape [B, {self.channels}, H, W], got {tuple(x.shape)}')
        return x.contiguous().float()

    def native_score(self, x, t, class_labels, sigma_min, sigma_max):
        x = self._prep(x)
        ctx = self.native_context(x.shape[0], x.shape[2], x.shape[3], x.device)
        out = torch.empty_like(x)
        lab = None if class_labels is None else class_labels.contiguous().float()
        ctx.score(x, t.contiguous().float(), lab, out, float(sigma_min), float(sigma_max))
        return out

    def native_cf_score(self, x, t, class_labels, weight, sigma_min, sigma_max):
        x = self._prep(x)
        ctx = self.native_context(2 * x.shape[0], x.shape[2], x.shape[3], x.device)
        out = torch.empty_like(x)
        ctx.cf_score(x, t.contiguous().float(), class_labels.contiguous().float(), weight.contiguous().float(), out,
                     float(sigma_min), float(sigma_max))
        return out

    # ------------------------------------------------------------------ nn.Module surface
    def forward(self, x, time_cond, class_labels=None):
        """RD/models/ncsnpp.py:226-354."""
        if self.conditional and self.training and self.cond_drop_prob > 0:
            # label drop for classifier-free guidance (:242-246): a [B] Bernoulli mask on the labels
            mask = (torch.rand(x.shape[0], device=x.device) < self.cond_drop_prob).float().unsqueeze(1)
            class_labels = class_labels * (1 - mask)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .. import autograd_fn
            return autograd_fn.ncsnpp_apply(self, x, time_cond, class_labels)
        if self.training and self.dropout > 0:
            raise NotImplementedError('train-mode forward without autograd (dropout active) is not built')
        x = self._prep(x)
        ctx = self.native_context(x.shape[0], x.shape[2], x.shape[3], x.device)
        out = torch.empty_like(x)
        lab = None if class_labels is None else class_labels.contiguous().float()
        ctx.forward(x, time_cond.contiguous().float(), lab, out)
        return out

    def get_tap(self, name, x, nb):
        """Debug hook: an intermediate activation of the last forward (needs RDMI_DEBUG_TAPS=1)."""
        ctx = self._ctx[(str(x.device), x.shape[2], x.shape[3])]
        return ctx.get_tap(name, x, nb)
