#!/usr/bin/env python
"""Conv-level audit of one critic step and one generator step at full width on the GPU: every conv forward / data
gradient / weight gradient the host issues, with rows and geometry, and the FLOP total next to the algorithmic minimum of
SURVEY 8(d) (D step B*(13 F_D - 3 f1) without the generator forward, G step 128*(3 F_G + 2 F_D))."""
import collections
import torch
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctgan_amd.kernels as K
import ctgan_amd.gan_cifar_resnet as R
import ctgan_amd.tflib as lib

lib.set_seed(5)
R.configure()
R.build_params('cuda')
tr = R.Trainer()
calls = []


def flops(N, g):
    return 2.0 * N * g.P * g.Q * g.K * g.R * g.S * g.C


def wrap(name, fn):
    f = getattr(K, name)

    def w(*a, **k):
        calls.append(fn(*a, **k))
        return f(*a, **k)
    setattr(K, name, w)


def gs(g):
    return '%dx%d C%d K%d %dx%d s%d' % (g.H, g.W, g.C, g.K, g.R, g.S, g.stride)


wrap('conv_fwd', lambda x, w, b, g, **k: ('fwd', x.shape[0], gs(g), flops(x.shape[0], g)))
wrap('conv_dgrad', lambda gy, w, g, N, **k: ('dgrad', N, gs(g), flops(N, g)))
wrap('conv_wgrad', lambda x, gy, g, **k: ('wgrad', x.shape[0], gs(g), flops(x.shape[0], g)))
wrap('conv_wgrad_multi', lambda segs, g, dw, db=None: ('wgrad_multi', sum(s[0].shape[0] for s in segs), gs(g),
                                                        sum(flops(s[0].shape[0], g) for s in segs)))
B = 64
real = torch.randint(0, 256, (B, 3072), dtype=torch.int32, device='cuda')
labels = torch.randint(0, 10, (B,), dtype=torch.int32, device='cuda')
fake = tr.generate_fakes(labels)[0]
for which in ('d', 'g'):
    del calls[:]
    if which == 'd':
        tr.d_step(real, labels, fake=fake)
    else:
        tr.g_step()
    c = collections.OrderedDict()
    for kind, n, g, f in calls:
        k = (kind, n, g)
        c.setdefault(k, [0, 0.0])
        c[k][0] += 1; c[k][1] += f
    tot = sum(v[1] for v in c.values())
    print('== %s step: %d conv launches, %.1f GFLOP executed' % (which, len(calls), tot / 1e9))
    for (kind, n, g), (cnt, f) in c.items():
        print('  %-12s n=%4d %-28s x%2d  %8.2f GFLOP' % (kind, n, g, cnt, f / 1e9))
FD, FG, f1 = 544.148e-3, 844.366e-3, 7.275e-3
print('algorithmic minimum (reference formulation): D step w/o generator fwd %.1f GFLOP, G step %.1f GFLOP'
      % (64 * (13 * FD - 3 * f1), 128 * (3 * FG + 2 * FD)))
