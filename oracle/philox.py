"""Counter-based random streams for the oracle.  TEST INFRASTRUCTURE ONLY.

The reference draws every random tensor with a TF op (`tf.random_normal` TF/CT_gan_cifar_resnet.py:157, `tf.random_uniform`
:202,277,319, `tf.nn.dropout` :173-177) from TF's own, unseeded generators - there is nothing to reproduce bit for bit, only
the DISTRIBUTIONS and WHERE each draw enters the graph.  The oracle takes every draw as an explicit input (`rnd` dicts of
oracle/steps.py).  This module regenerates, in plain numpy, the Philox4x32-10 streams (Salmon et al., SC'11; Random123
constants, known-answer vector checked in tests/test_oracle_philox.py) that the device path draws in its fused kernels, so the
oracle can be fed the IDENTICAL uniforms / normals / labels and the fused, graph-replayed step can be compared with the
reference graph as written (SURVEY.md 5.7, 8(d) "the parity harness must be able to inject the identical streams into the
CPU oracle").

Stream addressing (the contract both sides implement):
    value i of a stream  =  lane i & 3 of Philox(counter = {i >> 2, sid, step_lo, step_hi}, key = {seed_lo, seed_hi})
    sid  = (rank << 16) | call-site index within one step,   step = number of session.run-equivalents executed so far
    uniform  u = (x >> 8) * 2^-24  in [0,1)                 normal = Box-Muller on lane pairs, u1 = ((x>>8)+0.5)*2^-24
    a 4-D activation is addressed by its CHANNELS-LAST element index ((n*H + h)*W + w)*C + c
"""
import numpy as np
import torch

_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def philox_blocks(seed, sid, step, nblk, blk0=0):
    """uint32 [nblk,4]: Philox4x32-10 outputs of counters {blk0 + j, sid, step_lo, step_hi} under key `seed`."""
    c = np.zeros((nblk, 4), dtype=np.uint64)
    c[:, 0] = np.arange(blk0, blk0 + nblk, dtype=np.uint64)
    c[:, 1] = sid
    c[:, 2] = step & 0xffffffff
    c[:, 3] = step >> 32
    k0, k1 = seed & 0xffffffff, seed >> 32
    for _ in range(10):
        p0 = _M0 * c[:, 0]
        p1 = _M1 * c[:, 2]
        n0 = (p1 >> np.uint64(32)) ^ c[:, 1] ^ np.uint64(k0)
        n1 = p1 & np.uint64(0xffffffff)
        n2 = (p0 >> np.uint64(32)) ^ c[:, 3] ^ np.uint64(k1)
        n3 = p0 & np.uint64(0xffffffff)
        c = np.stack([n0, n1, n2, n3], 1) & np.uint64(0xffffffff)
        k0 = (k0 + _W0) & 0xffffffff
        k1 = (k1 + _W1) & 0xffffffff
    return c.astype(np.uint32)


def _u01(x):
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def uniform(seed, sid, step, n, lo=0.0, hi=1.0, first=0):
    """float32 [n]: elements first..first+n-1 of the uniform stream, U[lo,hi)."""
    b0, b1 = first >> 2, (first + n + 3) >> 2
    u = _u01(philox_blocks(seed, sid, step, b1 - b0, b0).reshape(-1))[first - 4 * b0: first - 4 * b0 + n]
    return (np.float32(lo) + np.float32(hi - lo) * u).astype(np.float32)


def normal(seed, sid, step, n, first=0):
    """float32 [n]: Box-Muller on lane pairs (0,1) and (2,3) of each block; `first` must be a multiple of 4."""
    assert first % 4 == 0
    b0, b1 = first >> 2, (first + n + 3) >> 2
    x = philox_blocks(seed, sid, step, b1 - b0, b0)
    out = np.empty((b1 - b0, 4), dtype=np.float32)
    for k in range(2):
        u1 = ((x[:, 2 * k] >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
        u2 = _u01(x[:, 2 * k + 1])
        r = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
        ang = (np.float32(6.283185307179586) * u2).astype(np.float32)
        out[:, 2 * k] = r * np.cos(ang)
        out[:, 2 * k + 1] = r * np.sin(ang)
    return out.reshape(-1)[:n]


def labels(seed, sid, step, n, nlab=10):
    """int32 [n] = trunc(u * nlab)  (tf.cast(float -> int32) truncates, TF/CT_gan_cifar_resnet.py:319)."""
    return (uniform(seed, sid, step, n) * np.float32(nlab)).astype(np.int32)


def dropout_u(seed, sid, step, N, C, H, W, first_row=0):
    """Logical [N,C,H,W] uniforms of a dropout site: the element of sample n, channel c, pixel (h,w) is value
    (((first_row + n)*H + h)*W + w)*C + c of the stream."""
    u = uniform(seed, sid, step, N * H * W * C, first=first_row * H * W * C)
    return np.ascontiguousarray(u.reshape(N, H, W, C).transpose(0, 3, 1, 2))


# ---- the draws of one step, in the call-site order of the device path ------------------------------------------------------
class Sites:
    """Call-site numbering of one step: restarts at 0 with every step, one index per random TENSOR in program order."""

    def __init__(self, rank=0):
        self.rank, self.n = rank, 0

    def next(self):
        s = self.n
        self.n += 1
        return (self.rank << 16) | s


def _t(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def rnd_resnet_d(seed, rank, step, B, DIM_D, dtype=torch.float64, z=None):
    """The `rnd` dict of oracle.steps.resnet_d_losses for the critic step executed at stream position `step`.
    Program order of the draws (= reference graph order of the live random ops): z :157 (skipped when `z` is given: the loop
    draws the fake batches of all N_CRITIC critic steps of an iteration in one earlier generator call), dequantisation noise
    :202, alpha :277, the gradient-penalty pass's three dropout masks :284 via :173-177, then the masks of the two dropout
    passes :226-227 - ONE stream per dropout site over the rows [pass 1: real, fake | pass 2: real]; the fake half of pass 2
    reaches no loss term (CT uses the real half only, :288-291) and gets a constant 0.75 (kept at every keep-prob used)."""
    st = Sites(rank)
    out = {}
    h = B // 2
    if z is None:
        zz = normal(seed, st.next(), step, B * 128).reshape(B, 128)
        out['z'] = [_t(zz[:h], dtype), _t(zz[h:], dtype)]
    else:
        out['z'] = [z[:h].to(dtype), z[h:].to(dtype)]
    out['dequant'] = _t(uniform(seed, st.next(), step, B * 3072, 0.0, 1.0 / 128).reshape(B, 3072), dtype)
    out['alpha'] = _t(uniform(seed, st.next(), step, B).reshape(B, 1), dtype)
    out['u_gp'] = [_t(dropout_u(seed, st.next(), step, B, DIM_D, 8, 8), dtype) for _ in range(3)]
    p1, p2 = [], []
    for _ in range(3):
        u = dropout_u(seed, st.next(), step, 3 * B, DIM_D, 8, 8)
        p1.append(_t(u[:2 * B], dtype))
        p2.append(torch.cat([_t(u[2 * B:], dtype), torch.full((B, DIM_D, 8, 8), 0.75, dtype=dtype)], 0))
    out['u_pass1'], out['u_pass2'] = p1, p2
    return out


def rnd_resnet_g(seed, rank, step, B, DIM_D, dtype=torch.float64, mult=2):
    """The `rnd` dict of oracle.steps.resnet_g_losses: fake labels :319, z :157, three dropout masks :321 via :173-177; the
    two towers are the two halves of each stream."""
    st = Sites(rank)
    n = mult * B
    h = n // 2
    lu = uniform(seed, st.next(), step, n)
    zz = normal(seed, st.next(), step, n * 128).reshape(n, 128)
    us = [dropout_u(seed, st.next(), step, n, DIM_D, 8, 8) for _ in range(3)]
    return {'label_u': [_t(lu[:h], torch.float32), _t(lu[h:], torch.float32)],      # fp32: the label is trunc(u * 10) in fp32 (:319)
            'z': [_t(zz[:h], dtype), _t(zz[h:], dtype)],
            'u': [[_t(u[:h], dtype) for u in us], [_t(u[h:], dtype) for u in us]]}


def fakes_z(seed, rank, step, n_rows):
    """z [n_rows,128] of the generator call that draws the fake batches of all critic steps of an iteration."""
    return torch.from_numpy(normal(seed, Sites(rank).next(), step, n_rows * 128).reshape(n_rows, 128).copy())
