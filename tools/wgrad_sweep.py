import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K
res = []
for N, H in [(128, 32), (64, 32), (192, 16), (128, 16), (64, 16), (192, 8), (128, 8), (64, 8), (128, 4)]:
    g = K.ConvGeom(128, H, H, 128, 3, 3, 1, False)
    x = K.empty_cl(N, 128, H, H, 'cuda').normal_(); gy = K.empty_cl(N, 128, H, H, 'cuda').normal_()
    K.conv_wgrad(x, gy, g, with_bias=True); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): K.conv_wgrad(x, gy, g, with_bias=True)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 30 * 1e-3
    res.append('%dx%d:%.0fus/%.0fTF(%s)' % (N, H, t * 1e6, 2.0 * N * H * H * 128 * 1152 / t / 1e12, K.last_kernel().split('<')[1].split(',bias')[0]))
print(os.environ.get('CTGAN_WGRAD_TILE', 'auto'), os.environ.get('CTGAN_WGRAD_K', 'auto'), ' '.join(res))
