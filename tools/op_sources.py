#!/usr/bin/env python
"""Which Python call sites launch the torch plumbing kernels (add / fill / copy / cat) of one eager D and G step.
usage: python tools/op_sources.py   (GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collections import Counter
import torch
from torch.profiler import profile, ProfilerActivity
import ctgan_amd.gan_cifar_resnet as R
import ctgan_amd.tflib as lib

lib.set_seed(1); R.configure(); R.build_params('cuda')
tr = R.Trainer(seed=1)
B = R.cfg.BATCH_SIZE
g = torch.Generator().manual_seed(0)
real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32).cuda(); lab = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32).cuda()
tr.d_step(real, lab); tr.g_step(); torch.cuda.synchronize()
for which in ('d', 'g'):
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        if which == 'd':
            tr.d_step(real, lab)
        else:
            tr.g_step()
        torch.cuda.synchronize()
    cnt = Counter()
    for e in prof.events():
        if e.name in ('aten::add', 'aten::add_', 'aten::zeros', 'aten::zero_', 'aten::fill_', 'aten::cat', 'aten::copy_', 'aten::ones_like',
                      'aten::zeros_like', 'aten::mul', 'aten::sum', 'aten::clone', 'aten::new_zeros', 'aten::neg', 'aten::div', 'aten::sub'):
            frames = [f for f in (e.stack or []) if 'ctgan_amd' in f]
            st = ' <- '.join(f.split('ctgan_amd/')[-1][:48] for f in frames[:2]) if frames else 'autograd engine'
            cnt[(e.name, str(e.input_shapes)[:48], st)] += 1
    print('==== %s step' % which)
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
        print('%3d  %-12s %-50s %s' % (v, k[0], k[1], k[2]))
