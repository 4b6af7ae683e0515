"""Second, independent restatement of TF SAME conv / conv-transpose as plain numpy tap loops.

Oracle cross-check only (test infrastructure).  Written from the definition, with no shared code
with oracle/tf_ops.py, so that a mistake in the asymmetric-pad / crop handling of either one shows
up as a disagreement (tests/test_oracle_anchors.py).
  conv:       y[n,o,p,q]  = sum_{r,s,i} x[n,i,p*st - pt + r, q*st - pl + s] * w[r,s,i,o]
  transpose:  y[n,o,h,w] += x[n,i,p,q] * w[r,s,o,i]   for h = p*st - pt + r, w = q*st - pl + s
where (pt,pl) are the leading SAME pads of the *forward* conv on the large (output) side.
"""
import numpy as np


def _same(in_size, k, st):
    out = (in_size + st - 1) // st
    total = max((out - 1) * st + k - in_size, 0)
    return out, total // 2


def conv2d_same_np(x, w, stride=1):
    n, ci, H, W = x.shape
    R, S, ci2, co = w.shape
    assert ci == ci2
    P, pt = _same(H, R, stride)
    Q, pl = _same(W, S, stride)
    y = np.zeros((n, co, P, Q), dtype=np.float64)
    for r in range(R):
        for s in range(S):
            for p in range(P):
                ih = p * stride - pt + r
                if ih < 0 or ih >= H:
                    continue
                for q in range(Q):
                    iw = q * stride - pl + s
                    if iw < 0 or iw >= W:
                        continue
                    y[:, :, p, q] += x[:, :, ih, iw].astype(np.float64) @ w[r, s].astype(np.float64)
    return y


def conv2d_transpose_same_np(x, w, stride=2):
    n, ci, H, W = x.shape
    R, S, co, ci2 = w.shape
    assert ci == ci2
    OH, OW = H * stride, W * stride
    _, pt = _same(OH, R, stride)
    _, pl = _same(OW, S, stride)
    y = np.zeros((n, co, OH, OW), dtype=np.float64)
    for r in range(R):
        for s in range(S):
            for p in range(H):
                h = p * stride - pt + r
                if h < 0 or h >= OH:
                    continue
                for q in range(W):
                    ww = q * stride - pl + s
                    if ww < 0 or ww >= OW:
                        continue
                    y[:, :, h, ww] += x[:, :, p, q].astype(np.float64) @ w[r, s].astype(np.float64).T
    return y
