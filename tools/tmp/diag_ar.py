import os, sys, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544'); os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)

def run(in_graph, dim, B, iters):
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    from ctgan_amd import ddp
    from ctgan_amd.engine import GraphedTrainer
    lib.delete_all_params(); lib.set_device(None); lib.set_seed(0)
    R.configure(DIM_G=dim, DIM_D=dim, BATCH_SIZE=B)
    R.build_params()
    ar = ddp.FlatAllReduce(always=True) if in_graph else None
    tr = R.Trainer(seed=2024, allreduce=ar)
    eng = GraphedTrainer(tr, use_graphs=True, ar_in_graph=in_graph)
    nrng = np.random.default_rng(1)
    batches = [(torch.from_numpy(nrng.integers(0, 256, (B, 3072), dtype=np.int32)).cuda(),
                torch.from_numpy(nrng.integers(0, 10, (B,), dtype=np.int32)).cuda()) for _ in range(8)]
    k = [0]
    def nb():
        k[0] = (k[0] + 1) % len(batches); return batches[k[0]]
    for it in range(1, 1 + iters):
        out = eng.train_iteration(it, nb)
    torch.cuda.synchronize()
    res = {'d_theta': tr.d_opt.theta.clone(), 'g_theta': tr.g_opt.theta.clone(), 'd_m': tr.d_opt.m.clone(), 'd_v': tr.d_opt.v.clone(), 'g_m': tr.g_opt.m.clone(),
           'g_v': tr.g_opt.v.clone(), 'state': tr._opt_state.clone(), 'ctr': tr.rng.ctr.clone(), 'd_grad': tr.d_opt.grad.clone(), 'g_grad': tr.g_opt.grad.clone()}
    lib.delete_all_params(); R.configure()
    return res

for iters in (3, 4):
    a = run(False, 32, 8, iters); b = run(True, 32, 8, iters); c = run(False, 32, 8, iters)
    print(iters, 'F vs T', {k: bool(torch.equal(a[k], b[k])) for k in a})
    print(iters, 'F vs F', {k: bool(torch.equal(a[k], c[k])) for k in a})
    print('state', a['state'].tolist(), b['state'].tolist(), 'ctr', a['ctr'].item(), b['ctr'].item())
dist.destroy_process_group()
