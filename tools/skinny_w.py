import sys, os
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (N, C, H, Ko) in [(128, 32, 32, 128), (64, 32, 32, 128), (128, 32, 16, 128)]:
    g = K.ConvGeom(C, H, H, Ko, 1, 1, 1, False)
    x = K.empty_cl(N, C, H, H, 'cuda').normal_(); gy = K.empty_cl(N, Ko, H, H, 'cuda').normal_()
    t = timeit(lambda: K.conv_wgrad(x, gy, g, with_bias=True))
    print(os.environ.get('CTGAN_WGRAD_K'), (N, C, H, Ko), '%.1f us' % t, K.last_kernel())
