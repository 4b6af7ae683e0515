"""Bit-exact resume on the MI355X: run N+1 iterations vs run N, checkpoint, restore into a fresh process
state, run 1 - weights, optimizer slots and the Philox stream position must continue identically."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_bit_exact_resume(tmp_path):
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    from ctgan_amd import checkpoint
    g = torch.Generator().manual_seed(0)
    batches = [(torch.randint(0, 256, (8, 3072), generator=g, dtype=torch.int32).cuda(),
                torch.randint(0, 10, (8,), generator=g, dtype=torch.int32).cuda()) for _ in range(4)]

    def run(n_iters, resume_from=None, save_at=None):
        lib.delete_all_params(); lib.set_device(None); lib.set_seed(4)
        R.configure(DIM_G=16, DIM_D=16, BATCH_SIZE=8)
        R.build_params()
        tr = R.Trainer(seed=9)
        start = 0
        if resume_from:
            start = checkpoint.load(resume_from, tr)
        k = [start * 5]

        def nxt():
            k[0] += 1
            return batches[k[0] % 4]
        for it in range(start, n_iters):
            tr.train_iteration(it, nxt)
            if save_at is not None and it + 1 == save_at:
                checkpoint.save(str(tmp_path / 'ck.pt'), tr, iteration=it + 1)
        return tr.d_opt.theta.clone(), tr.g_opt.theta.clone(), tr.d_opt.m.clone()
    try:
        a = run(3, save_at=2)
        b = run(3, resume_from=str(tmp_path / 'ck.pt'))
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    finally:
        lib.delete_all_params(); R.configure()


def test_reference_style_training_loop(tmp_path):
    """`train()` = the loop of TF/CT_gan_cifar_resnet.py:393-434 on a fake CIFAR directory: generator-factory
    feed with device prefetch, hipGraph steps, metric series, sample grid, checkpoint."""
    import os
    import pickle

    import numpy as np
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    rng = np.random.default_rng(0)
    for name in ['data_batch_%d' % k for k in range(1, 6)] + ['test_batch']:
        with open(os.path.join(str(tmp_path), name), 'wb') as f:
            pickle.dump({'data': rng.integers(0, 256, (64, 3072), dtype=np.uint8), 'labels': [int(v) for v in rng.integers(0, 10, 64)]}, f, protocol=2)
    lib.delete_all_params(); lib.set_device(None); lib.set_seed(1)
    R.configure(DIM_G=16, DIM_D=16, BATCH_SIZE=8, ITERS=6)
    try:
        tr = R.train(str(tmp_path), n_examples=200, out_dir=str(tmp_path), sample_every=3, checkpoint_every=4, log=lambda *a: None)
        assert tr.d_opt.t == 30 and tr.g_opt.t == 5
        assert os.path.exists(os.path.join(str(tmp_path), 'samples_2.png')) and os.path.exists(os.path.join(str(tmp_path), 'checkpoint.pt'))
        assert sum(1 for _ in open(os.path.join(str(tmp_path), 'log.jsonl'))) == 6
    finally:
        lib.delete_all_params(); R.configure()
