"""Checkpoint compatibility: the reference's checkpoint layout (RD/utils.py:48-86) is
{'step', 'model': state_dict, 'optimizer': state_dict, 'ema': {'decay','num_updates','shadow_params': [tensor,...]},
 'scaler', 'config'} with the model's parameter names -- identical here, so a reference-trained file loads into the
rdmi model and a file written here loads into the reference.

Loading is restricted to `torch.load(weights_only=True)`: tensors and plain containers only, nothing in the file is
executed.  The reference stores its omegaconf `config` object in the file (RD/utils.py:85).  Such a file still loads here
WITHOUT omegaconf and without running anything from it: the names of the globals the pickle refers to are read with
`torch.serialization.get_unsafe_globals_in_checkpoint` (no unpickling), every `omegaconf.*` class among them is mapped to an
INERT stand-in (a class with no behaviour that only records the state the pickle hands it), `typing.Any` / `builtins.dict` /
`builtins.list` map to themselves, and any other name makes the load fail with the list of offenders.  The stand-in graph is
then flattened to plain dicts / lists / scalars, so `loaded['config']` is a nested dict.  (omegaconf is not installed in the
build image: the flattening is tested on a file whose pickle has the same class names and state layout, written in the test --
a file written by omegaconf itself is "parity unpinned".)  `save_checkpoint` writes the config as a plain nested dict.
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


class _Inert:
    """Stand-in for a class of a module that is absent here (omegaconf.*): no behaviour, it only keeps what the pickle gives it."""

    def __new__(cls, *args, **kwargs):
        return object.__new__(cls)

    def __init__(self, *args, **kwargs):
        self._inert_args = args

    def __setstate__(self, state):
        self._inert_state = state


_HARMLESS = {'typing.Any': None, 'builtins.dict': dict, 'builtins.list': list, 'builtins.tuple': tuple, 'builtins.int': int,
             'builtins.float': float, 'builtins.str': str, 'builtins.bool': bool, 'builtins.object': object}


def _stand_ins(names):
    import typing
    out, bad = [], []
    for n in names:
        if n.startswith('omegaconf.'):
            mod, _, cls = n.rpartition('.')
            out.append((type(cls, (_Inert,), {'__module__': mod}), n))
        elif n in _HARMLESS:
            out.append((typing.Any if n == 'typing.Any' else _HARMLESS[n], n))
        else:
            bad.append(n)
    return out, bad


def _flatten(o, seen=None):
    """Inert omegaconf stand-ins -> plain containers: a container's `_content`, a value node's `_val`; back-references (`_parent`)
    and metadata are dropped."""
    seen = set() if seen is None else seen
    if isinstance(o, _Inert):
        if id(o) in seen:
            return None
        seen.add(id(o))
        st = getattr(o, '_inert_state', None)
        if isinstance(st, tuple):                # (dict state, slots state)
            st = st[0] if isinstance(st[0], dict) else (st[1] if len(st) > 1 else None)
        if isinstance(st, dict):
            if '_content' in st:
                return _flatten(st['_content'], seen)
            if '_val' in st:
                return _flatten(st['_val'], seen)
        return None
    if isinstance(o, dict):
        return {(_flatten(k, seen) if isinstance(k, _Inert) else k): _flatten(v, seen) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_flatten(v, seen) for v in o]
    return o


def _load(path, device):
    try:
        names = torch.serialization.get_unsafe_globals_in_checkpoint(path)        # reads opcodes only, unpickles nothing
        if not names:
            return torch.load(path, map_location=device, weights_only=True)
        stubs, bad = _stand_ins(names)
        if bad:
            raise RuntimeError(f'the pickle refers to {bad}: only omegaconf.* containers (mapped to inert stand-ins) are accepted beside tensors')
        with torch.serialization.safe_globals(stubs):
            loaded = torch.load(path, map_location=device, weights_only=True)
        if isinstance(loaded, dict) and 'config' in loaded:
            loaded['config'] = _flatten(loaded['config'])
        return loaded
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
