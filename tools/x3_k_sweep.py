import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ctgan_amd.kernels as K
K.X3_HYBRID = False      # families are compared explicitly here: 'f32' means the fp32 MFMA family on every layer
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
for (N, C, H, Ko, R) in [(192,128,32,128,1),(192,128,32,128,3),(192,128,32,128,5),(192,128,32,128,7),(192,256,32,128,3),(192,512,32,128,3)]:
    g = K.ConvGeom(C, H, H, Ko, R, R, 1, False)
    x = K.empty_cl(N, C, H, H, 'cuda').normal_()
    w = torch.randn(R, R, C, Ko, device='cuda') * 0.02
    K._STABLE_PTRS.add(w.data_ptr())
    fl = 2.0 * N * g.P * g.Q * Ko * R * R * C
    with K.mma_dtype('f32x3'):
        t = timed(lambda: K.conv_fwd(x, w, None, g))
    nk = R * R * C // 32
    print('C%d R%d nk=%d  %6.1f us  %5.0f TFLOP/s   per-round %.1f us' % (C, R, nk, t * 1e6, fl / t / 1e12, t * 1e6 / 6))
