"""Helper of tests/test_gpu_graph_loop.py::test_all_reduce_captured_in_the_step_graphs (run as a subprocess: it owns a process group).
A 1-rank RCCL group on one GPU: the same training loop (whole-iteration hipGraph replay) with the gradient all-reduce + Adam captured
inside the graphs (engine.AR_IN_GRAPH) and without any collective must produce bit-identical weights - the sum over one rank is the
identity, 1/world = 1 - which exercises RCCL enqueue under stream capture and graph replay on this stack.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist


def run(in_graph, dim, B, iters, world=1, split=False):
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    from ctgan_amd import ddp
    from ctgan_amd.engine import GraphedTrainer
    lib.delete_all_params(); lib.set_device(None); lib.set_seed(0)
    R.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B)
    R.build_params()
    # split: Trainer.split_flush - the bucket's prefix on a collective of its own, forked onto a side stream INSIDE the capture
    ar = ddp.FlatAllReduce(always=True, side_stream=torch.cuda.Stream() if split else None) if in_graph else None
    # world = 2 on a 1-rank group: the sum over one rank is the identity, so the only effect is Adam's grad_scale = 1 / world = 0.5 -
    # the in-graph path (gather, all-reduce, step(1/world)) against the path without a collective at the same scale
    tr = R.Trainer(seed=2024, world_size=world, allreduce=ar)
    tr.split_flush = bool(split)
    handed = []
    if split:
        orig = tr.early_reduce
        tr.early_reduce = lambda gp: (handed.append(len(gp)), orig(gp))[1]
    eng = GraphedTrainer(tr, use_graphs=True, ar_in_graph=in_graph)
    assert not split or handed, 'the split flush never handed a part of the bucket over (is the hand-scheduled step in use?)'
    assert eng.graphed and (eng.it_graph is not None or (world > 1 and not in_graph)), eng.graph_error
    assert eng.ar_in_graph == bool(in_graph)
    nrng = np.random.default_rng(1)
    batches = [(torch.from_numpy(nrng.integers(0, 256, (B, 3072), dtype=np.int32)).cuda(),
                torch.from_numpy(nrng.integers(0, 10, (B,), dtype=np.int32)).cuda()) for _ in range(8)]
    k = [0]

    def nb():
        k[0] = (k[0] + 1) % len(batches)
        return batches[k[0]]
    out = None
    for it in range(1, 1 + iters):
        out = eng.train_iteration(it, nb)
    torch.cuda.synchronize()
    res = (tr.d_opt.theta.clone(), tr.g_opt.theta.clone(), float(out['cost'].item()))
    lib.delete_all_params(); R.configure()
    return res


if __name__ == '__main__':
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    dim, B, iters = (int(v) for v in sys.argv[1:4])
    a = run(False, dim, B, iters)
    b = run(True, dim, B, iters)
    a2 = run(False, dim, B, iters, world=2)
    b2 = run(True, dim, B, iters, world=2)
    split_rec = None
    if len(sys.argv) > 4:          # width / batch of the split-flush runs (the hand-scheduled critic step needs DIM_D in {64, 128, 256})
        sd, sB = int(sys.argv[4]), int(sys.argv[5])
        c1 = run(True, sd, sB, 2, world=2)
        c2 = run(True, sd, sB, 2, world=2, split=True)
        dd = (c1[0] - c2[0]).abs()
        split_rec = {'cost_one_bucket': c1[2], 'cost_split': c2[2], 'theta_max_abs_diff': float(dd.max().item()),
                     'theta_frac_above_2e-5': float((dd > 2e-5).float().mean().item()), 'g_equal': bool(torch.equal(c1[1], c2[1]))}
    print(json.dumps({'split_flush': split_rec, 'scaled_d_equal': bool(torch.equal(a2[0], b2[0])), 'scaled_g_equal': bool(torch.equal(a2[1], b2[1])),
                      'scaled_differs_from_unscaled': bool(not torch.equal(a[0], a2[0])),
                      'd_equal': bool(torch.equal(a[0], b[0])), 'g_equal': bool(torch.equal(a[1], b[1])), 'cost_plain': a[2], 'cost_in_graph': b[2],
                      'backend': dist.get_backend(), 'd_moved': float((a[0] - torch.zeros_like(a[0])).abs().max().item())}))
    dist.destroy_process_group()
