"""Seeded, NON-DEGENERATE synthetic NCSN++ weights (test infrastructure).

The reference's trained checkpoints are absent and its own init
multiplies every Conv_1 / NIN_3 / out_conv by 1e-10 (init_scale=0,
RD/models/layers.py:73-76), so an untouched random-init net outputs ~0 and would
hide most of the network from a parity check.  This recipe draws every tensor
at O(1) scale from a counter-based numpy generator, keyed by (seed, index in the
reference's state-dict order), so the build container (golden generation, where
the tensors are loaded into the *reference* model with strict=True) and the GPU
box (parity tests, bench) get bit-identical weights without shipping 25 MB.

param_specs() restates the 261-entry state-dict layout of RD/models/ncsnpp.py:42-224.
"""
import hashlib

import numpy as np

F32 = np.float32


def param_specs(nf=64, ch_mult=(1, 2, 2), num_res_blocks=2, attn_resolutions=(9,), image_size=9,
                channels=1, num_classes=1):
    """Ordered [(name, shape)] exactly as the reference registers them."""
    specs = []
    add = lambda n, *s: specs.append((n, tuple(s)))
    temb = nf * 4

    def gn(pre, c):
        add(pre + '.weight', c); add(pre + '.bias', c)

    def conv(pre, cin, cout):
        add(pre + '.weight', cout, cin, 3, 3); add(pre + '.bias', cout)

    def ninp(pre, cin, cout):
        add(pre + '.W', cin, cout); add(pre + '.b', cout)

    def resblock(pre, cin, cout):
        gn(pre + '.GroupNorm_0', cin); conv(pre + '.Conv_0', cin, cout)
        add(pre + '.Dense_0.weight', cout, temb); add(pre + '.Dense_0.bias', cout)
        gn(pre + '.GroupNorm_1', cout); conv(pre + '.Conv_1', cout, cout)
        if cin != cout:
            ninp(pre + '.NIN_0', cin, cout)

    def attn(pre, c):
        gn(pre + '.GroupNorm_0', c)
        for i in range(4):
            ninp(pre + f'.NIN_{i}', c, c)

    add('time_embed.W', nf)
    add('time_mlp.0.weight', temb, 2 * nf); add('time_mlp.0.bias', temb)
    add('time_mlp.2.weight', temb, temb); add('time_mlp.2.bias', temb)
    add('label_emb.weight', temb, num_classes); add('label_emb.bias', temb)
    conv('input_conv', channels, nf)
    nlev = len(ch_mult)
    attn_at = [image_size // (2 ** i) in attn_resolutions for i in range(nlev)]
    # down path: blocks, then attention blocks, then downsamplers (ModuleList registration order)
    in_ch, skip, down_attn, downs, d = nf, [], [], [], 0
    for i, m in enumerate(ch_mult):
        for _ in range(num_res_blocks):
            resblock(f'down_blocks.{d}', in_ch, nf * m)
            in_ch = nf * m
            if attn_at[i]:
                down_attn.append((d, in_ch))
            skip.append(in_ch)
            d += 1
        skip.append(in_ch)
        if i != nlev - 1:
            downs.append((i, in_ch))
    for d_, c in down_attn:
        attn(f'down_attn.{d_}', c)
    for i, c in downs:
        conv(f'downsample.{i}.Conv_0', c, c)
    resblock('mid_block1', in_ch, in_ch)
    resblock('mid_block2', in_ch, in_ch)
    skip = list(reversed(skip))
    up_attn, ups, u = [], [], 0
    for k, i in enumerate(reversed(range(nlev))):
        for _ in range(num_res_blocks + 1):
            resblock(f'up_blocks.{u}', in_ch + skip.pop(0), nf * ch_mult[i])
            in_ch = nf * ch_mult[i]
            if attn_at[i]:
                up_attn.append((u, in_ch))
            u += 1
        if i != 0:
            ups.append((k, in_ch))
    for u_, c in up_attn:
        attn(f'up_attn.{u_}', c)
    for k, c in ups:
        conv(f'upsample.{k}.Conv_0', c, c)
    gn('out_norm', in_ch)
    conv('out_conv', in_ch, channels)
    return specs


def make_params(seed=0, **arch):
    """name -> fp32 array.  Distribution per tensor kind (all O(1) activations):
    conv / NIN / Dense weights U(+-sqrt(3/fan_avg)); nn.Linear-style weights U(+-1/sqrt(fan_in));
    biases 0.05*N(0,1); GroupNorm gamma 1+0.1*N, beta 0.1*N; Fourier W = 16*N(0,1)."""
    out = {}
    for idx, (name, shape) in enumerate(param_specs(**arch)):
        rng = np.random.Generator(np.random.Philox(key=[seed, idx]))
        leaf = name.split('.')[-1]
        if name == 'time_embed.W':
            v = rng.standard_normal(shape) * 16.0
        elif 'GroupNorm' in name or name.startswith('out_norm'):
            v = (1.0 if leaf == 'weight' else 0.0) + 0.1 * rng.standard_normal(shape)
        elif leaf in ('bias', 'b'):
            v = 0.05 * rng.standard_normal(shape)
        elif len(shape) == 4:                               # conv OIHW
            fan_in, fan_out = shape[1] * 9, shape[0] * 9
            v = rng.uniform(-1, 1, shape) * np.sqrt(3.0 / ((fan_in + fan_out) / 2))
        elif leaf == 'W':                                   # NIN [in, out]
            v = rng.uniform(-1, 1, shape) * np.sqrt(3.0 / ((shape[0] + shape[1]) / 2))
        elif 'Dense_0' in name:                             # [out, in]
            v = rng.uniform(-1, 1, shape) * np.sqrt(3.0 / ((shape[0] + shape[1]) / 2))
        else:                                               # time_mlp / label_emb  [out, in]
            v = rng.uniform(-1, 1, shape) / np.sqrt(shape[1])
        out[name] = np.ascontiguousarray(v, dtype=F32)
    return out


def params_sha256(params):
    h = hashlib.sha256()
    for k in params:
        h.update(k.encode()); h.update(params[k].tobytes())
    return h.hexdigest()
