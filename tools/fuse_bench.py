import sys, os
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
for N, H in [(128, 32), (128, 16), (192, 8)]:
    g = K.ConvGeom(128, H, H, 128, 3, 3, 1, False)
    x = K.empty_cl(N, 128, H, H, 'cuda').normal_(); w = torch.randn(3, 3, 128, 128, device='cuda') * 0.05
    gy = K.empty_cl(N, 128, H, H, 'cuda').normal_(); r = K.empty_cl(N, 128, H, H, 'cuda').normal_()
    wt = K.repack_filter(w, g)
    print(N, H, 'fwd %.1f  fwd_relu_in %.1f | dgrad %.1f  dgrad_mask %.1f  dgrad_mask_resid %.1f | wgrad %.1f wgrad_relu %.1f | lrelu_f %.1f lrelu_b %.1f' % (
        timeit(lambda: K.conv_fwd(x, w, None, g)), timeit(lambda: K.conv_fwd(x, w, None, g, relu_in=True)),
        timeit(lambda: K.conv_dgrad(gy, w, g, N, wt=wt)), timeit(lambda: K.conv_dgrad(gy, w, g, N, wt=wt, mask=x)),
        timeit(lambda: K.conv_dgrad(gy, w, g, N, wt=wt, mask=x, resid=r)),
        timeit(lambda: K.conv_wgrad(x, gy, g)), timeit(lambda: K.conv_wgrad(x, gy, g, relu_x=True)),
        timeit(lambda: K.lrelu_fwd(x, 0.0)), timeit(lambda: K.lrelu_bwd(gy, x, 0.0))))
