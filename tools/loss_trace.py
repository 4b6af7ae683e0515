#!/usr/bin/env python
"""Per-critic-step loss trace of the benchmarked loop (bench.py's seeds, batches and ordering): prints cost / wgan / ct / gp /
acgan of every D step plus max|theta| of both networks, for hipGraph replay or eager launches.  Fusion switches come from
the CTGAN_* environment variables (see gan_cifar_resnet.py / functional.py), so a shell loop bisects them."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=8)
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--dim', type=int, default=128)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--tag', default='')
    args = ap.parse_args()
    import numpy as np
    import torch
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    from ctgan_amd.engine import GraphedTrainer
    dev = torch.device('cuda', 0)
    lib.delete_all_params(); lib.set_seed(0)
    R.configure(DIM_G=args.dim, DIM_D=args.dim, BATCH_SIZE=args.batch)
    R.build_params(dev)
    tr = R.Trainer(seed=2024)
    B = R.cfg.BATCH_SIZE
    nrng = np.random.default_rng(1234)
    batches = [(torch.from_numpy(nrng.integers(0, 256, (B, 3072), dtype=np.int32)).to(dev),
                torch.from_numpy(nrng.integers(0, 10, (B,), dtype=np.int32)).to(dev)) for _ in range(16)]
    cur = [0]

    def nb():
        cur[0] = (cur[0] + 1) % 16
        return batches[cur[0]]

    eng = GraphedTrainer(tr, use_graphs=not args.no_graph)
    print(json.dumps({'tag': args.tag, 'graphed': eng.graphed, 'graph_error': eng.graph_error,
                      'env': {k: v for k, v in os.environ.items() if k.startswith('CTGAN_')}}), flush=True)
    for it in range(1, args.iters + 1):
        eng.g_step(it)
        bs = [nb() for _ in range(R.cfg.N_CRITIC)]
        if eng.graphed:
            for i, (_, lab) in enumerate(bs):
                eng.labels_all[i * B:(i + 1) * B].copy_(lab)
            eng.f_graph.replay()
            fakes = eng.fake_all
        else:
            fakes = tr.generate_fakes(torch.cat([lab for _, lab in bs], 0))
        for i, (x, lab) in enumerate(bs):
            out = eng.d_step(x, lab, it, fake=fakes[i])
            rec = {k: float(out[k].item()) for k in ('cost', 'wgan', 'ct', 'gp', 'acgan') if out.get(k) is not None}
            rec.update(it=it, d=i, th_d=float(tr.d_opt.theta.abs().max().item()), th_g=float(tr.g_opt.theta.abs().max().item()),
                       fake_max=float(fakes[i].abs().max().item()))
            print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
