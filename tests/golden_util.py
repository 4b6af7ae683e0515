"""Helpers to replay tests/golden/resnet_trace.npz (shared by the CPU and GPU suites)."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def trace_weights(z):
    return {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w.')}


def trace_d_inputs(z, it, dtype=torch.float32, device='cpu'):
    pre = 'it%d.' % it
    t = lambda k: torch.from_numpy(z[pre + k]).to(dtype).to(device)   # noqa: E731
    rnd = {'z': [t('d.z.0'), t('d.z.1')], 'dequant': t('d.dequant'), 'alpha': t('d.alpha'),
           'u_pass1': [t('d.u_pass1.%d' % i) for i in range(3)], 'u_pass2': [t('d.u_pass2.%d' % i) for i in range(3)],
           'u_gp': [t('d.u_gp.%d' % i) for i in range(3)]}
    real = torch.from_numpy(z[pre + 'real']).to(device)
    labels = torch.from_numpy(z[pre + 'labels']).to(device)
    return real, labels, rnd


def trace_g_inputs(z, it, dtype=torch.float32, device='cpu'):
    pre = 'it%d.' % it
    t = lambda k: torch.from_numpy(z[pre + k]).to(dtype).to(device)   # noqa: E731
    return {'z': [t('g.z.0'), t('g.z.1')], 'label_u': [t('g.label_u.0'), t('g.label_u.1')],
            'u': [[t('g.u.%d.%d' % (j, i)) for i in range(3)] for j in range(2)]}
