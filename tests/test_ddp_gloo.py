"""world_size-2 data-parallel step over gloo on CPU (HIP kernel wrappers swapped for the test-only
torch-CPU stand-ins): the flat-bucket all-reduce + 1/world averaging inside Adam must reproduce the
average of the per-rank gradients, and every rank must end with identical weights."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _patch_cpu():
    import ctgan_amd.kernels as K
    import ctgan_amd.tflib as lib
    from tests import cpu_kernels as M
    for name in M.__all__:
        setattr(K, name, getattr(M, name))
    lib.delete_all_params()
    lib.set_device('cpu')
    return lib


def _inputs(rank, B, dim):
    from oracle import steps as osteps
    g = torch.Generator().manual_seed(100 + rank)
    real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
    labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
    rnd = osteps.make_rnd_resnet_d(B, dim, g, dtype=torch.float32)
    rg = osteps.make_rnd_resnet_g(B, dim, g, dtype=torch.float32)
    return real, labels, rnd, rg


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    lib = _patch_cpu()
    import ctgan_amd.gan_cifar_resnet as R
    from ctgan_amd import ddp
    r, w, _ = ddp.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    lib.set_seed(7 + rank)                        # deliberately different inits: broadcast must fix it
    R.configure(DIM_G=8, DIM_D=8, BATCH_SIZE=4)
    R.build_params('cpu')
    tr = R.Trainer(seed=1, rank=rank, world_size=world, allreduce=ddp.FlatAllReduce())
    ddp.broadcast_params([tr.d_opt.theta, tr.g_opt.theta])
    real, labels, rnd, rg = _inputs(rank, 4, 8)
    o1 = tr.d_step(real, labels, rnd, iteration=0)
    o2 = tr.g_step(rg, iteration=1)
    snap = {'d': tr.d_opt.theta.clone(), 'g': tr.g_opt.theta.clone(), 'dcost': o1['cost'].detach(), 'gcost': o2['cost'].detach()}
    # then the loop body in the reference's order through the engine ([G] + 5 x D, default fused path, per-rank Philox
    # streams and batches): 1 + 5 more all-reduces; the replicas must still hold the same bits afterwards
    from ctgan_amd.engine import GraphedTrainer
    eng = GraphedTrainer(tr, use_graphs=False)
    g = torch.Generator().manual_seed(500 + rank)
    feed = [(torch.randint(0, 256, (4, 3072), generator=g, dtype=torch.int32), torch.randint(0, 10, (4,), generator=g, dtype=torch.int32))
            for _ in range(5)]
    k = [0]

    def nb():
        k[0] += 1
        return feed[(k[0] - 1) % 5]
    o3 = eng.train_iteration(2, nb)
    snap.update(d_loop=tr.d_opt.theta.clone(), g_loop=tr.g_opt.theta.clone(), loop_cost=o3['cost'].detach(), d_t=tr.d_opt.t, g_t=tr.g_opt.t)
    torch.save(snap, os.path.join(out_dir, 'rank%d.pt' % rank))
    ddp.barrier()
    torch.distributed.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def test_two_rank_step_equals_average_of_rank_gradients(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r)) for r in range(world)]
    assert torch.equal(res[0]['d'], res[1]['d']) and torch.equal(res[0]['g'], res[1]['g'])      # replicas stay in sync
    assert not torch.equal(res[0]['dcost'], res[1]['dcost'])                                     # different shards
    assert torch.equal(res[0]['d_loop'], res[1]['d_loop']) and torch.equal(res[0]['g_loop'], res[1]['g_loop'])
    assert not torch.equal(res[0]['d_loop'], res[0]['d']) and not torch.equal(res[0]['loop_cost'], res[1]['loop_cost'])
    assert (res[0]['d_t'], res[0]['g_t']) == (6, 2)

    # single-process reference: rank-0 init, per-rank gradients computed separately, averaged, one Adam step
    lib = _patch_cpu()
    import ctgan_amd.gan_cifar_resnet as R
    try:
        lib.set_seed(7)
        R.configure(DIM_G=8, DIM_D=8, BATCH_SIZE=4)
        R.build_params('cpu')
        tr = R.Trainer(seed=1)
        gsum = None
        for r in range(world):
            real, labels, rnd, _ = _inputs(r, 4, 8)
            tr.rng.begin_step()
            out = tr.d_losses(real, labels, rnd)
            grads = torch.autograd.grad(out['cost'], tr.d_params, allow_unused=True)
            flat = tr.d_opt.gather_grads(grads).clone()
            gsum = flat if gsum is None else gsum + flat
        tr.d_opt.grad.copy_(gsum)
        tr.d_opt.set_lr(tr.lr(0))
        tr.d_opt.step(grad_scale=1.0 / world)
        assert torch.allclose(tr.d_opt.theta, res[0]['d'], rtol=0, atol=1e-6)
        gsum = None
        for r in range(world):
            _, _, _, rg = _inputs(r, 4, 8)
            tr.rng.begin_step()
            out = tr.g_losses(rg)
            grads = torch.autograd.grad(out['cost'], tr.g_params, allow_unused=True)
            flat = tr.g_opt.gather_grads(grads).clone()
            gsum = flat if gsum is None else gsum + flat
        tr.g_opt.grad.copy_(gsum)
        tr.g_opt.set_lr(tr.lr(1))
        tr.g_opt.step(grad_scale=1.0 / world)
        # entries whose gradient is analytically zero (biases feeding a batch norm) carry fp32 noise that
        # Adam normalises into O(lr) steps: compare those loosely, everything else tightly
        live = gsum.abs() > 1e-6
        assert torch.allclose(tr.g_opt.theta[live], res[0]['g'][live], rtol=0, atol=2e-6)
        assert torch.allclose(tr.g_opt.theta[~live], res[0]['g'][~live], rtol=0, atol=5e-4)
    finally:
        lib.delete_all_params(); lib.set_device(None); R.configure()


def _split_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    lib = _patch_cpu()
    import ctgan_amd.critic_schedule as CS
    import ctgan_amd.functional as F
    import ctgan_amd.gan_cifar_resnet as R
    from ctgan_amd import ddp
    from ctgan_amd.engine import GraphedTrainer
    ddp.init_from_env(backend='gloo')
    res = {}
    for split in (False, True):
        lib.delete_all_params(); lib.set_device('cpu'); lib.set_seed(7)
        R.configure(DIM_G=8, DIM_D=64, BATCH_SIZE=2)                  # DIM_D = 64: the width from which the hand-scheduled step applies
        R.build_params('cpu')
        tr = R.Trainer(seed=1, rank=rank, world_size=world, allreduce=ddp.FlatAllReduce())
        tr.split_flush = split
        ddp.broadcast_params([tr.d_opt.theta, tr.g_opt.theta])
        calls, early_calls = [], []
        orig, orig_early = CS.critic_step, tr.early_reduce
        CS.critic_step = lambda *a, **k: (calls.append(k.get('early') is not None), orig(*a, **k))[1]
        tr.early_reduce = lambda gp: (early_calls.append(sorted(gp)), orig_early(gp))[1]
        try:
            g = torch.Generator().manual_seed(100 + rank)
            real = torch.randint(0, 256, (2, 3072), generator=g, dtype=torch.int32)
            labels = torch.randint(0, 10, (2,), generator=g, dtype=torch.int32)
            F.prepare_filters()
            fake = tr.generate_fakes(labels)[0]
            out = tr.d_step(real, labels, iteration=0, fake=fake)
            grads = tr.d_opt.grad.clone()                              # the reduced bucket
            eng = GraphedTrainer(tr, use_graphs=False)                 # and through the engine's eager loop
            o2 = eng.train_iteration(1, lambda: (real, labels))
        finally:
            CS.critic_step = orig
        assert calls and all(c == split for c in calls), calls
        assert (len(early_calls) == len(calls)) if split else not early_calls
        if split:
            assert all(n.startswith(('Discriminator.1.', 'Discriminator.2.')) for n in early_calls[0]) and len(early_calls[0]) == 12
        res[split] = {'grad': grads, 'theta': tr.d_opt.theta.clone(), 'g': tr.g_opt.theta.clone(), 'cost': out['cost'].detach().clone(),
                      'cost2': o2['cost'].detach().clone()}
    torch.save(res, os.path.join(out_dir, 'split_rank%d.pt' % rank))
    ddp.barrier()
    torch.distributed.destroy_process_group()
    R.configure()


def test_split_flush_hands_blocks_1_2_over_early_and_changes_no_bit(tmp_path):
    """Trainer.split_flush (the all-reduce overlapped with backward the north star asks for, behind a switch): the
    hand-scheduled critic step flushes the weight gradients of blocks 1-2 when the penalty's double backward has left them and hands that
    prefix of the flat bucket to its own all-reduce; the rest follows at the end of the step.  World 2 over gloo: the reduced bucket, the
    weights after the step and after a further loop iteration are BIT-identical to the one-bucket form on every rank (the stand-in
    kernels are row-order independent; on the GPU the two grouped launches plan their split-K chunks separately - same values to fp32
    rounding), and the replicas stay in sync."""
    world = 2
    mp.spawn(_split_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), 'split_rank%d.pt' % r)) for r in range(world)]
    for r in range(world):
        a, b = res[r][False], res[r][True]
        for k in a:
            assert torch.equal(a[k], b[k]), (r, k)
    assert torch.equal(res[0][True]['theta'], res[1][True]['theta']) and torch.equal(res[0][True]['g'], res[1][True]['g'])
    assert not torch.equal(res[0][True]['cost'], res[1][True]['cost'])
