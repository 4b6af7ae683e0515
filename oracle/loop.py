"""The training loop of the reference, restated over the oracle steps.  TEST INFRASTRUCTURE ONLY.

Ordering of TF/CT_gan_cifar_resnet.py:393-404: `[gen_train_op if iteration > 0]`, then N_CRITIC x (next batch,
disc_train_op), both optimizers at LR * max(0, 1 - iteration/ITERS) (:309-312).  Every session.run draws fresh random
tensors; here they come from the counter-based streams of oracle/philox.py, advanced once per run-equivalent in the order
    [G step]  ->  the generator call that draws the fake batches of the N_CRITIC critic steps  ->  N_CRITIC critic steps
(the generator does not change between the critic steps of an iteration, so their fake batches may be drawn together: the
critic step i uses rows [i*B, (i+1)*B) of that z, two BN towers of B/2 each, exactly the two-tower call of :196-199).
"""
import torch

from . import philox, steps


def resnet_train_loop(reg, cfg, next_batch, iters, B, seed, rank=0, start_iteration=0, start_step=0, ITERS=100000, LR=2e-4,
                      N_CRITIC=5, dtype=torch.float64, optD=None, optG=None, on_step=None):
    """Runs `iters` iterations; returns (list of per-critic-step loss dicts, list of per-iteration generator costs,
    optD, optG, stream position).  next_batch() -> (int32 [B,3072], int32 [B]).  on_step(kind, iteration, k, out) is called
    after every G ('g') and critic ('d') update."""
    optD = optD or steps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Discriminator.')], 0.0, 0.9)
    optG = optG or steps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Generator')], 0.0, 0.9)
    step = start_step
    d_recs, g_recs = [], []
    for it in range(start_iteration, start_iteration + iters):
        if it > 0:
            rg = philox.rnd_resnet_g(seed, rank, step, B, cfg.DIM_D, dtype=dtype)
            out = steps.resnet_g_step(reg, cfg, optG, rg, iteration=it, ITERS=ITERS, LR=LR, B=B)
            step += 1
            g_recs.append(float(out['cost'].detach()))
            if on_step:
                on_step('g', it, 0, out)
        batches = [next_batch() for _ in range(N_CRITIC)]
        z_all = philox.fakes_z(seed, rank, step, N_CRITIC * B)
        step += 1
        for k, (real, labels) in enumerate(batches):
            rnd = philox.rnd_resnet_d(seed, rank, step, B, cfg.DIM_D, dtype=dtype, z=z_all[k * B:(k + 1) * B])
            out = steps.resnet_d_step(reg, cfg, optD, real, labels, rnd, iteration=it, ITERS=ITERS, LR=LR, B=B)
            step += 1
            d_recs.append({n: float(out[n].detach()) for n in ('cost', 'wgan', 'acgan', 'ct', 'gp', 'wgan_only')})
            if on_step:
                on_step('d', it, k, out)
    return d_recs, g_recs, optD, optG, step
