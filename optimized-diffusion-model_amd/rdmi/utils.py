"""Checkpoint compatibility: the reference's checkpoint layout (RD/utils.py:48-86) is
{'step', 'model': state_dict, 'optimizer': state_dict, 'ema': {'decay','num_updates','shadow_params': [tensor,...]},
 'scaler', 'config'} with the model's parameter names -- identical here, so a reference-trained file loads into the
rdmi model and a file written here loads into the reference.

Loading is restricted to `torch.load(weights_only=True)`: tensors and plain containers only, nothing in the file is
executed.  The reference stores its omegaconf `config` object in the file; a file that carries one is refused by that
loader with a clear message (re-save it without the config entry where omegaconf is available).  `save_checkpoint`
writes the config as a plain nested dict for the same reason.
"""
import logging
import os

import torch


def _plain(cfg):
    if cfg is None or isinstance(cfg, (int, float, str, bool)):
        return cfg
    if isinstance(cfg, dict):
        return {k: _plain(v) for k, v in cfg.items()}
    if isinstance(cfg, (list, tuple)):
        return [_plain(v) for v in cfg]
    if hasattr(cfg, '__dict__'):
        return {k: _plain(v) for k, v in vars(cfg).items()}
    return str(cfg)


def _load(path, device):
    try:
        return torch.load(path, map_location=device, weights_only=True)
    except Exception as e:   # pickle.UnpicklingError subclasses vary across torch versions
        raise RuntimeError(f'{path}: not loadable with torch.load(weights_only=True) ({type(e).__name__}: {e}); rdmi never '
                           'unpickles arbitrary objects -- re-save the checkpoint with its `config` entry as a plain dict') from e


def restore_checkpoint(ckpt_dir, state, device, ddp=True):
    """RD/utils.py:48-67."""
    if not os.path.exists(ckpt_dir):
        os.makedirs(os.path.dirname(ckpt_dir) or '.', exist_ok=True)
        logging.warning(f'No checkpoint found at {ckpt_dir}. Returned the same state as input')
        return state
    loaded = _load(ckpt_dir, device)
    state['optimizer'].load_state_dict(loaded['optimizer'])
    model = state['model'].module if hasattr(state['model'], 'module') else state['model']
    model.load_state_dict(loaded['model'], strict=False)
    state['ema'].load_state_dict(loaded['ema'])
    state['step'] = loaded['step']
    if state.get('scaler') is not None and loaded.get('scaler') is not None:
        state['scaler'].load_state_dict(loaded['scaler'])
    return state


def load_denoising_model(ckpt_dir, model, device=torch.device('cpu')):
    """RD/utils.py:70-75."""
    if not os.path.exists(ckpt_dir):
        raise ValueError(f'No checkpoint found at {ckpt_dir}.')
    model.load_state_dict(_load(ckpt_dir, device)['model'], strict=False)
    return model


def save_checkpoint(ckpt_dir, state):
    """RD/utils.py:78-86 (config flattened to plain containers)."""
    model = state['model'].module if hasattr(state['model'], 'module') else state['model']
    checkpoint = {
        'step': state['step'],
        'model': model.state_dict(),
        'optimizer': state['optimizer'].state_dict(),
        'ema': state['ema'].state_dict() if state.get('ema') is not None else None,
        'scaler': state['scaler'].state_dict() if state.get('scaler') is not None else None,
        'config': _plain(state.get('config')),
    }
    torch.save(checkpoint, ckpt_dir)
