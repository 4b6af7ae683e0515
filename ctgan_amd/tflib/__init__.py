"""Drop-in for the reference's `tflib` core: the name-keyed parameter registry.

Mirrors TF/tflib/__init__.py:10-48,101-113: `param(name, value)` creates the parameter on first
use and returns the existing one afterwards (that is what makes the 3-4 Discriminator(...) calls
of one step share weights); `params_with_name(substr)` selects optimizer variable lists.
Parameters are `torch.nn.Parameter`s on the current HIP device, stored in the reference's layouts
(`.Filters` HWIO / [k,k,out,in], `.W` [in,out]) so fixtures and checkpoints are layout-compatible.
"""
import collections
import zlib

import numpy as np
import torch

_params = collections.OrderedDict()
_param_aliases = {}
_non_trainable = set()
_device = None
_seed = 0
_epoch = [0]        # bumped whenever parameter VALUES change (optimizer step, load): invalidates derived caches
_group_epoch = {}   # network ('Discriminator', 'Generator', ...) -> extra bumps of that network's parameters only


def bump_epoch(group=None):
    """Parameter values changed: all of them (group=None) or those of one network (first component of their names)."""
    if group is None:
        _epoch[0] += 1
    else:
        _group_epoch[group] = _group_epoch.get(group, 0) + 1


def epoch(group=None):
    """Version of the parameter values of `group` (or of the whole registry)."""
    return (_epoch[0], _group_epoch.get(group, 0)) if group is not None else (_epoch[0], sum(_group_epoch.values()))


def group_of(tensor):
    """Network a registered parameter belongs to (first component of its name); None if not registered."""
    for name, p in _params.items():
        if p is tensor:
            return name.split('.')[0]
    return None


def set_device(device):
    """Device on which new parameters are created (default: current HIP device)."""
    global _device
    _device = torch.device(device) if device is not None else None


def set_seed(seed):
    """Seed of the per-name init streams (one independent numpy stream per parameter name)."""
    global _seed
    _seed = int(seed)


def rng_for(name):
    return np.random.default_rng([_seed, zlib.crc32(name.encode())])


def _dev():
    if _device is not None:
        return _device
    if not torch.cuda.is_available():
        raise RuntimeError('ctgan_amd.tflib: no HIP device visible (and no CPU fallback); '
                           'call tflib.set_device(...) explicitly in tests that only build registries')
    return torch.device('cuda', torch.cuda.current_device())


def param(name, value=None, trainable=True, **kwargs):
    """lib.param (TF/tflib/__init__.py:10-34).  `value` may be an ndarray, a tensor or a callable
    `rng -> ndarray` (evaluated only when the parameter does not exist yet)."""
    if name not in _params:
        if callable(value):
            value = value(rng_for(name))
        t = torch.as_tensor(np.asarray(value, dtype=np.float32) if not torch.is_tensor(value) else value)
        p = torch.nn.Parameter(t.to(device=_dev(), dtype=torch.float32).contiguous(), requires_grad=bool(trainable))
        p.param = True
        p.param_name = name
        _params[name] = p
        if not trainable:
            _non_trainable.add(name)
    result = _params[name]
    while result in _param_aliases:
        result = _param_aliases[result]
    return result


def params_with_name(name):
    """TF/tflib/__init__.py:36-37 - substring match; includes the non-trainable moving statistics."""
    return [p for n, p in _params.items() if name in n]


def named_params_with_name(name, trainable_only=False):
    return [(n, p) for n, p in _params.items() if name in n and not (trainable_only and n in _non_trainable)]


_delete_hooks = []


def on_delete_all_params(fn):
    """Register a callback run by delete_all_params (derived-weight caches hold references to parameters)."""
    if fn not in _delete_hooks:
        _delete_hooks.append(fn)


def delete_all_params():
    _params.clear()
    _non_trainable.clear()
    bump_epoch()
    for fn in _delete_hooks:
        fn()


def alias_params(replace_dict):
    for old, new in replace_dict.items():
        _param_aliases[old] = new


def delete_param_aliases():
    _param_aliases.clear()


def state_dict():
    """name -> fp32 CPU tensor (reference layouts)."""
    return collections.OrderedDict((n, p.detach().cpu().clone()) for n, p in _params.items())


def from_ls_names(sd):
    """A name -> array mapping under the parameter names of the LSUN tree's operator library - conv / deconv biases `name.b`
    (LS/tflib/ops/conv2d.py:117, deconv2d.py:108), Layernorm / Batchnorm offsets `name.b` (LS/tflib/ops/layernorm.py:15,22,
    batchnorm.py:24) - renamed to the names this library (and the TF tree: TF/tflib/ops/conv2d.py:116, layernorm.py:14) registers:
    `name.Biases` next to a `name.Filters`, `name.offset` next to a `name.scale`.  A Linear's `name.b` is the same in both trees."""
    out = collections.OrderedDict()
    for n, v in sd.items():
        if n.endswith('.b'):
            base = n[:-2]
            if base + '.Filters' in sd:
                n = base + '.Biases'
            elif base + '.scale' in sd:
                n = base + '.offset'
        if n in out:
            raise KeyError('from_ls_names: %s appears under both naming schemes' % n)
        out[n] = v
    return out


def to_ls_names(sd):
    """Inverse of from_ls_names: this registry's names -> the LSUN tree's (for a checkpoint the LS scripts' Saver / lib.param read)."""
    out = collections.OrderedDict()
    for n, v in sd.items():
        if n.endswith('.Biases') and n[:-7] + '.Filters' in sd:
            n = n[:-7] + '.b'
        elif n.endswith('.offset') and n[:-7] + '.scale' in sd:
            n = n[:-7] + '.b'
        out[n] = v
    return out


def load_state_dict(sd, strict=True, names='tf'):
    """Create / overwrite parameters from a name -> array mapping (explicit weight injection:
    the reference's numpy-seed replay is not reproducible).  names='ls': `sd` carries the LSUN tree's parameter
    names (config[4]'s script, LS/wgan_LSUN_Bedrooms128.py) - translated by from_ls_names first."""
    if names == 'ls':
        sd = from_ls_names(sd)
    elif names != 'tf':
        raise ValueError("load_state_dict: names must be 'tf' or 'ls'")
    for n, v in sd.items():
        v = torch.as_tensor(np.asarray(v, dtype=np.float32)) if not torch.is_tensor(v) else v
        if n in _params:
            with torch.no_grad():
                _params[n].copy_(v.to(_params[n].device, torch.float32))
        else:
            param(n, v.detach().to(torch.float32), trainable=not n.endswith(('.moving_mean', '.moving_variance')))
    bump_epoch()
    if strict:
        missing = [n for n in _params if n not in sd]
        if missing:
            raise KeyError('load_state_dict: missing %s' % missing[:5])


def print_model_settings(locals_):
    """TF/tflib/__init__.py:101-106"""
    print('Uppercase local vars:')
    all_vars = [(k, v) for (k, v) in locals_.items()
                if (k.isupper() and k != 'T' and k != 'SETTINGS' and k != 'ALL_SETTINGS')]
    for var_name, var_value in sorted(all_vars, key=lambda x: x[0]):
        print('\t{}: {}'.format(var_name, var_value))


def print_model_settings_dict(settings):
    print('Settings dict:')
    for var_name, var_value in sorted(settings.items(), key=lambda x: x[0]):
        print('\t{}: {}'.format(var_name, var_value))
