"""CPU oracle, torch-CPU flavour (TEST / BASELINE INFRASTRUCTURE ONLY -- same rules as rd_oracle.py).

A second restatement of the reference's score evaluation + PC update on plain torch CPU tensor ops
(F.conv2d, F.group_norm, matmul) written as pure functions over a state-dict.  It exists because the
reference's own CPU path runs on exactly these library kernels (oneDNN/MKL): timing THIS on the GPU box's host
cores is the fair `cpu_baseline` for bench.py (the numpy oracle is ~20x slower on narrow GEMMs and would
flatter the GPU).  Pinned against the same reference-recorded fixtures (tests/test_oracle_golden.py).
Citations as in rd_oracle.py.
"""
import math

import torch
import torch.nn.functional as F


def sigma_of(t, smin=0.01, smax=5.0):
    return smin * (smax / smin) ** t                                            # RD/sde_lib.py:143


def g_of(t, smin=0.01, smax=5.0):
    return sigma_of(t, smin, smax) * torch.sqrt(torch.tensor(2 * (math.log(smax) - math.log(smin)), dtype=torch.float32))


def reflect(x):
    m = torch.remainder(x, 2.0)                                                 # RD/cube.py:47-49
    return torch.where(m > 1, 2 - m, m)


def _gn(x, p, pre):
    C = x.shape[1]
    return F.group_norm(x, min(C // 4, 32), p[pre + '.weight'], p[pre + '.bias'], eps=1e-6)


def _nin(x, p, pre):
    return torch.einsum('bchw,co->bohw', x, p[pre + '.W']) + p[pre + '.b'][None, :, None, None]


def _resblock(p, pre, x, ta):
    h = F.conv2d(F.silu(_gn(x, p, pre + '.GroupNorm_0')), p[pre + '.Conv_0.weight'], p[pre + '.Conv_0.bias'], padding=1)
    h = h + F.linear(ta, p[pre + '.Dense_0.weight'], p[pre + '.Dense_0.bias'])[:, :, None, None]
    h = F.conv2d(F.silu(_gn(h, p, pre + '.GroupNorm_1')), p[pre + '.Conv_1.weight'], p[pre + '.Conv_1.bias'], padding=1)
    if (pre + '.NIN_0.W') in p:
        x = _nin(x, p, pre + '.NIN_0')
    return (x + h) / math.sqrt(2.)


def _attn(p, pre, x):
    B, C, H, W = x.shape
    h = _gn(x, p, pre + '.GroupNorm_0')
    q, k, v = (_nin(h, p, pre + f'.NIN_{i}').reshape(B, C, H * W) for i in range(3))
    w = torch.softmax(torch.matmul(q.transpose(1, 2), k) * (int(C) ** (-0.5)), dim=-1)
    h = torch.matmul(v, w.transpose(1, 2)).reshape(B, C, H, W)
    return (x + _nin(h, p, pre + '.NIN_3')) / math.sqrt(2.)


def ncsnpp_forward(p, x, sigma, labels, ch_mult=(1, 2, 2), nrb=2, attn_levels=(True, False, False), scale_by_sigma=False):
    """NCSNpp.forward, eval mode (RD/models/ncsnpp.py:226-354); scale_by_sigma: h / time_cond (:350-351)."""
    xp = (torch.log(sigma)[:, None] * p['time_embed.W'][None, :]) * 2 * math.pi
    temb = torch.cat([torch.sin(xp), torch.cos(xp)], dim=-1)
    temb = F.linear(F.silu(F.linear(temb, p['time_mlp.0.weight'], p['time_mlp.0.bias'])), p['time_mlp.2.weight'], p['time_mlp.2.bias'])
    temb = temb + F.linear(labels, p['label_emb.weight'], p['label_emb.bias'])
    ta = F.silu(temb)
    h = F.conv2d(x, p['input_conv.weight'], p['input_conv.bias'], padding=1)
    hs, d, nlev = [h], 0, len(ch_mult)
    for i in range(nlev):
        for _ in range(nrb):
            h = _resblock(p, f'down_blocks.{d}', h, ta)
            if attn_levels[i]:
                h = _attn(p, f'down_attn.{d}', h)
            hs.append(h); d += 1
        hs.append(h)
        if i != nlev - 1:
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), p[f'downsample.{i}.Conv_0.weight'], p[f'downsample.{i}.Conv_0.bias'], stride=2)
    h = _resblock(p, 'mid_block2', _resblock(p, 'mid_block1', h, ta), ta)
    u = 0
    for i in range(nlev):
        for _ in range(nrb + 1):
            sk = hs.pop()
            if h.shape[2:] != sk.shape[2:]:
                h = F.interpolate(h, size=sk.shape[2:], mode='nearest')
            h = _resblock(p, f'up_blocks.{u}', torch.cat([h, sk], dim=1), ta)
            if attn_levels[nlev - 1 - i]:
                h = _attn(p, f'up_attn.{u}', h)
            u += 1
        if i != nlev - 1:
            h = F.interpolate(h, scale_factor=2, mode='nearest')
            h = F.conv2d(h, p[f'upsample.{i}.Conv_0.weight'], p[f'upsample.{i}.Conv_0.bias'], padding=1)
    h = F.conv2d(F.silu(_gn(h, p, 'out_norm')), p['out_conv.weight'], p['out_conv.bias'], padding=1)
    return h / sigma.view(-1, 1, 1, 1) if scale_by_sigma else h


CIFAR_ARCH = dict(ch_mult=(1, 2, 2, 2), nrb=8, attn_levels=(False, True, False, False), scale_by_sigma=True)   # BASELINE config #5


def cf_score(p, x, t, labels, w, smax=5.0, **arch):
    """RD/models/utils.py:120-138."""
    B = x.shape[0]
    s = ncsnpp_forward(p, x.repeat(2, 1, 1, 1), sigma_of(t.repeat(2), smax=smax), torch.cat([labels, torch.zeros_like(labels)]), **arch)
    w = w.view(-1, 1, 1, 1)
    return (1 + w) * s[:B] - w * s[B:]


def pc_update(p, x, t, labels, w, z_pred, N, z_corr=None, snr=0.01, smax=5.0, **arch):
    """One iteration of the PC loop (RD/sampling.py:330-332): optional Langevin step, then Euler-Maruyama."""
    if z_corr is not None:
        s = cf_score(p, x, t, labels, w, smax=smax, **arch)
        gn = s.reshape(s.shape[0], -1).norm(dim=-1).mean(); zn = z_corr.reshape(s.shape[0], -1).norm(dim=-1).mean()
        step = (snr * zn / gn) ** 2 * 2
        x = reflect(x + step * s + torch.sqrt(step * 2) * z_corr)
    s = cf_score(p, x, t, labels, w, smax=smax, **arch)
    g = g_of(t, smax=smax)[:, None, None, None]
    x_mean = x + (g ** 2 * s) / N
    return reflect(x_mean + g * math.sqrt(1.0 / N) * z_pred)
