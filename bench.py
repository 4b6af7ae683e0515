#!/usr/bin/env python
"""bench.py - img/s of the CT-WGAN adversarial iteration (N_CRITIC critic steps + 1 generator step)
of CT_gan_cifar_resnet.py on MI355X, synthetic 32x32x3 data, fp32 (the reference's dtype).

  python bench.py --gpus N --steps K --warmup W
N > 1: either launched under torch.distributed.run (one rank per GPU, RCCL; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from
the environment), or - when WORLD_SIZE is not set - this process becomes a LAUNCHER: before anything touches the GPU it
starts N rank processes of itself (127.0.0.1 rendezvous on a free port), relays rank 0's JSON line and exits with the
first non-zero rank status; it refuses (exit 2) when fewer than N devices are visible (--backend gloo lets N ranks share
one device to exercise the multi-rank code path).  A "step" here is one full
iteration of the reference loop body (TF/CT_gan_cifar_resnet.py:393-404): 1 G step + 5 x (batch,
D step) at BATCH_SIZE=64 per GPU = 320 real images per GPU.  Inputs are resident in HBM before the
timed region.  Rank 0 prints ONE JSON line including
  roofline     : the dominant kernel (by GPU time, keyed by device symbol) of the iteration, its
                 algorithmic FLOPs per launch / average launch duration, measured with HIP events on
                 the launch stream in an instrumented eager iteration of the same launch mix, against
                 the peak of the matrix pipe it runs on (fp32 MFMA 157.3 TFLOP/s; split mode 2500/6)
  cpu_baseline : the CPU oracle (PyTorch-CPU fp32 restatement of the reference graph AS WRITTEN;
                 TF1 cannot be installed here) timed on the host cores - 1 warm-up + 2 timed D steps and
                 G steps, extrapolated to an iteration G + 5 D (N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_16BIT_MFMA_TFLOPS = 2500.0       # dense bf16 / fp16 MFMA (same guide); never the 2:1-sparsity figure
# --config name -> (module under ctgan_amd, kernels.mma_dtype, workload description)
ALT_CONFIGS = {
    'cifar_dcgan_bf16': ('gan_cifar', 'bf16', 'CT_gan_cifar.py DCGAN 32x32 CT-WGAN, batch 64, DIM 128, 5x5 stride-2 convs / transposed convs on bf16 MFMA (fp32 accumulate, fp32 master weights)'),
    'cifar_dcgan_f32': ('gan_cifar', None, 'CT_gan_cifar.py DCGAN 32x32 CT-WGAN, batch 64, DIM 128, fp32 MFMA'),
    'lsun128_f16': ('gan_lsun128', 'f16', '128x128 ResNet CT-WGAN (LS/wgan_LSUN_Bedrooms128.py nets), batch 64/GPU, convs on fp16 MFMA (fp32 accumulate, fp32 master weights)'),
    'lsun128_bf16': ('gan_lsun128', 'bf16', '128x128 ResNet CT-WGAN (LS/wgan_LSUN_Bedrooms128.py nets), batch 64/GPU, convs on bf16 MFMA'),
    'lsun128_f32': ('gan_lsun128', None, '128x128 ResNet CT-WGAN (LS/wgan_LSUN_Bedrooms128.py nets), batch 64/GPU, fp32 MFMA'),
}
ITER_GFLOP = 2990.5                   # algorithmic GFLOP per GPU per iteration


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)       # warm-up 20, measure >= 100 (2 s at 19 ms / step)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--gp-unit-only', action='store_true', help='run only the critic-forward + GP-backward sub-benchmark')
    ap.add_argument('--config', default='resnet', choices=['resnet'] + sorted(ALT_CONFIGS),
                    help='resnet = the headline (BASELINE.json configs[2]/[3], fp32); the others: configs[1] / configs[4] on the '
                         '16-bit matrix cores (or their fp32 twins for comparison)')
    ap.add_argument('--backend', default=None, help='torch.distributed backend (default nccl = RCCL); gloo lets the\n'
                    'multi-rank code path be exercised on a single-GPU box (all ranks share cuda:0)')
    ap.add_argument('--launcher', default='auto', choices=['auto', 'always', 'never'],
                    help='auto: start the N rank processes here when --gpus N > 1 and WORLD_SIZE is unset; always: also for N = 1')
    ap.add_argument('--no-ar-in-graph', action='store_true', help='N > 1: the eager side-stream all-reduce + Adam outside the step graphs (the default for gloo)')
    ap.add_argument('--ar-in-graph', action='store_true',
                    help='N > 1: capture the gradient all-reduces (RCCL) and Adam inside the step graphs - one graph per iteration as at N = 1 '
                         '(engine.AR_IN_GRAPH; default: eager all-reduce on a side stream between per-step graphs)')
    ap.add_argument('--no-ab-legs', action='store_true',
                    help='N > 1: do not start the A/B legs (in-graph RCCL collectives, split flush) after the record has been printed')
    ap.add_argument('--leg-timeout', type=float, default=150.0, help='N > 1: seconds an A/B leg (a fresh group of rank processes) may take before it is killed')
    ap.add_argument('--leg', default=None, choices=sorted(LEGS),
                    help='(internal) this process is one rank of an A/B leg started by a rank of the main run AFTER it printed its record: '
                         'prints a short record of its own, never starts legs')
    ap.add_argument('--feed', default='both', choices=['device', 'both'],
                    help='device: the timed loop cycles 16 device-resident batches (the headline `value`); both: also re-time the loop with the\n'
                         'real input path - tflib.cifar10.EpochFeed over synthetic uint8 images -> pinned host buffers -> H2D on a copy stream ->\n'
                         'staging kernel (the span TF/CT_gan_cifar_resnet.py:394-412 times includes the feed) - reported as config.host_feed')
    ap.add_argument('--spawn-check', action='store_true',
                    help='ranks only join the process group, all-reduce their rank numbers and report (no GPU work): the launcher test')
    args = ap.parse_args()
    if 'WORLD_SIZE' not in os.environ and (args.launcher == 'always' or (args.launcher == 'auto' and args.gpus > 1)):
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if args.spawn_check:
        return spawn_check(args)
    if args.leg is not None:
        args.no_roofline = args.no_cpu_baseline = args.no_ab_legs = True
        args.feed = 'device'
        args.ar_in_graph, args.no_graph = LEGS[args.leg]['ar_in_graph'], LEGS[args.leg]['no_graph']
    # N > 1: the RECORD leg runs the eager side-stream collective (ddp.FlatAllReduce between per-step graphs: the path the 2-rank gloo tests
    # cover) unless --ar-in-graph asks for the captured one; the captured RCCL collectives and the split flush are A/B legs, run as fresh
    # rank groups after the record is out (run_ab_legs): a hang or a watchdog abort there cannot cost the record
    ar_in_graph = bool(args.ar_in_graph) and not args.no_ar_in_graph

    import numpy as np
    import torch
    import torch.distributed as dist

    from ctgan_amd import ddp
    if args.config != 'resnet':
        return run_unconditional(args)
    rank, world, local = init_ranks(args.backend)
    if local >= torch.cuda.device_count():
        if world > 1 and dist.get_backend() != 'gloo':
            raise SystemExit('rank %d: LOCAL_RANK %d but only %d device(s) visible - two RCCL ranks cannot share a GPU '
                             '(use --backend gloo to exercise the multi-rank code path on one device)'
                             % (rank, local, torch.cuda.device_count()))
        local = local % max(torch.cuda.device_count(), 1)       # gloo test mode only (ranks share a device)
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: the record would not say what ran' % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.kernels as K
    import ctgan_amd.tflib as lib
    from ctgan_amd.engine import GraphedTrainer

    lib.delete_all_params()
    lib.set_seed(0)
    R.configure()                                      # full-width reference hyper-parameters
    R.build_params(dev)
    side = torch.cuda.Stream() if world > 1 else None
    trainer = R.Trainer(seed=2024, rank=rank, world_size=world, allreduce=ddp.FlatAllReduce(side_stream=side))
    ddp.broadcast_params([trainer.d_opt.theta, trainer.g_opt.theta])

    B = R.cfg.BATCH_SIZE
    nrng = np.random.default_rng(1234 + rank)          # 16 distinct synthetic batches, cycled
    batches = [(torch.from_numpy(nrng.integers(0, 256, (B, 3072), dtype=np.int32)).to(dev),
                torch.from_numpy(nrng.integers(0, 10, (B,), dtype=np.int32)).to(dev)) for _ in range(16)]
    cursor = [0]

    def next_batch():
        cursor[0] = (cursor[0] + 1) % len(batches)
        return batches[cursor[0]]

    if args.gp_unit_only:
        print(json.dumps(measure_gp_unit(trainer, batches[0], torch)))
        return
    if args.leg is not None:
        trainer.split_flush = LEGS[args.leg]['split_flush']
    eng = GraphedTrainer(trainer, use_graphs=not args.no_graph, ar_in_graph=ar_in_graph)
    if world > 1:
        # every rank must run the same path: if any rank could not capture the in-graph collective, all fall back to the side-stream one.
        # (Only a failed capture of the STEP graphs counts: the whole-iteration graph is optional - off by design with CTGAN_ITERATION_GRAPH=0
        # or without the batched fake draw - and the per-step graphs carry the collective without it; it_graph_error is reported separately.)
        bad = torch.tensor([1.0 if (eng.ar_in_graph and (eng.graph_error is not None or not eng.graphed)) else 0.0], device=dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if bad.item() > 0 and eng.ar_in_graph:
            if rank == 0:
                print('bench: in-graph all-reduce not captured on every rank (%s / %s): falling back to the side-stream collective'
                      % (eng.graph_error, eng.it_graph_error), file=sys.stderr)
            eng = GraphedTrainer(trainer, use_graphs=not args.no_graph, ar_in_graph=False)
    if eng.graph_error and rank == 0:
        print('hipGraph capture failed, running eager: ' + eng.graph_error, file=sys.stderr)
    elif eng.it_graph_error and rank == 0:
        print('whole-iteration graph not captured (per-step graphs in use): ' + eng.it_graph_error, file=sys.stderr)

    it = 1
    for _ in range(args.warmup):
        eng.train_iteration(it, next_batch); it += 1
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # per-step times (p50) without host syncs
    ddp.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        out = eng.train_iteration(it, next_batch); it += 1
        marks[k + 1].record()
    torch.cuda.synchronize(); ddp.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    p50 = per_step[len(per_step) // 2]
    last = {k: float(out[k].item()) for k in ('cost', 'wgan', 'ct', 'gp', 'acgan') if out.get(k) is not None}
    last_cost = last['cost']
    ms_per_step = 1e3 * dt / args.steps
    imgs = R.cfg.N_CRITIC * B * world * args.steps
    value = imgs / dt
    # Loss guard: a WGAN-GP critic on bounded inputs cannot leave this band in a few hundred Adam steps of <= 3*lr each.
    # Round 1 timed a loop whose cost had run to -6e18 (graph outputs aliased in a shared pool) and nothing looked.
    sane = all(v == v and abs(v) < 1e4 for v in last.values())

    # N > 1: the replicas must have stayed bit-identical through the timed loop (same initial weights, averaged gradients, the same Adam):
    # every rank's parameter checksums (bit patterns of the fp64 sums of both networks' flat buffers), gathered and compared
    replicas_identical = None
    if world > 1:
        cs = torch.stack([trainer.d_opt.theta.double().sum(), trainer.g_opt.theta.double().sum(),
                          trainer.d_opt.theta.double().abs().sum(), trainer.g_opt.theta.double().abs().sum()]).view(torch.int64)
        allcs = [torch.zeros_like(cs) for _ in range(world)]
        dist.all_gather(allcs, cs)
        replicas_identical = all(torch.equal(allcs[0], c) for c in allcs[1:])
        sane = sane and replicas_identical

    # N > 1: what the six gradient all-reduces of an iteration cost on the critical path.  Three engines on the same loop: the default (one
    # bucket per step), the split flush (Trainer.split_flush: blocks 1-2 of the critic's bucket all-reduced under the rest of the backward -
    # the overlap the north star asks for, behind a switch) and the same launches with the collective replaced by nothing
    # (the replicas drift apart from there on: nothing below compares them).  exposed = loop time - loop time without the collective.
    collective = None
    if world > 1:
        def timed_loop(e, k):
            nonlocal it
            for _ in range(3):
                e.train_iteration(it, next_batch); it += 1
            ddp.barrier(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(k):
                e.train_iteration(it, next_batch); it += 1
            torch.cuda.synchronize(); ddp.barrier()
            tl = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
            dist.all_reduce(tl, op=dist.ReduceOp.MAX)
            return 1e3 * tl.item() / k

        class _NoComm:          # same call surface as ddp.FlatAllReduce, no communication: the launches of the step without the collective
            always = True
            def inline(self, flat): pass
            def __call__(self, flat): pass
            def wait(self): pass
        k2 = max(5, min(20, args.steps))
        collective = {'all_reduces_per_step': R.cfg.N_CRITIC + 1, 'in_graph': bool(eng.ar_in_graph), 'steps': k2,
                      'bucket_bytes': {'critic': 4 * trainer.d_opt.theta.numel(), 'generator': 4 * trainer.g_opt.theta.numel()},
                      'critic_bucket_prefix_bytes_split_flush': 4 * trainer.d_opt.offsets[trainer._n_early] if trainer._n_early < len(trainer.d_opt.offsets) else None}
        saved_ar = trainer.allreduce
        try:
            trainer.allreduce = _NoComm()
            eng_n = GraphedTrainer(trainer, use_graphs=not args.no_graph, ar_in_graph=eng.ar_in_graph)
            ms_local = timed_loop(eng_n, k2)
            del eng_n
            collective['ms_per_step_without_all_reduce'] = round(ms_local, 3)
            collective['exposed_ms_per_step'] = round(ms_per_step - ms_local, 3)
            # the arithmetic behind the >= 0.85 scaling target: the per-GPU work is fixed (weak scaling), so whole-job efficiency at N ranks
            # = ms_per_step(1 GPU) / ms_per_step(N) ~ 1 - exposed / ms_per_step as long as the launches themselves take what they take alone
            collective['exposed_frac_of_step'] = round((ms_per_step - ms_local) / ms_per_step, 4)
            collective['scaling_efficiency_bound'] = round(ms_local / ms_per_step, 4)
        except Exception as e:          # noqa: BLE001
            collective['without_all_reduce_error'] = '%s: %s' % (type(e).__name__, e)
        finally:
            trainer.allreduce = saved_ar
        collective['note'] = ('in_graph: the all-reduces are nodes of the iteration graph (RCCL under stream capture); otherwise (the record leg\'s '
                              'default) eager collectives on a side stream between per-step graphs, the next step\'s input staging enqueued while the '
                              'bucket is in flight (DESIGN 5).  exposed = this loop - the same loop with the collective replaced by nothing; '
                              'scaling_efficiency_bound = that loop / this loop.  The captured-collective and split-flush engines are timed by A/B legs '
                              'started after this record is printed (stderr lines "bench: leg ...")')

    # The same loop fed the way the reference's loop is (TF/CT_gan_cifar_resnet.py:394-412 times the feed too; TF/tflib/cifar10.py:40-63 is
    # the contract): uint8 epochs shuffled on the host, pinned prefetch two batches deep, H2D on a copy stream, the iteration's staging
    # kernel.  Device-resident batches stay the headline (`value`); this is the PCIe-inclusive rate beside it.
    host_feed = None
    if args.feed == 'both':
        from ctgan_amd.tflib import cifar10
        frng = np.random.default_rng(99 + rank)
        n_img = 10000
        feed_src = cifar10.EpochFeed(frng.integers(0, 256, (n_img, 3072), dtype=np.uint8), frng.integers(0, 10, (n_img,), dtype=np.uint8).astype(np.int32), B)
        feed = cifar10.prefetch_to_device(cifar10.inf_train_gen(feed_src), dev, depth=2 * R.cfg.N_CRITIC)
        nxt = lambda: next(feed)      # noqa: E731
        k3 = max(5, min(args.steps, 50))
        for _ in range(3):
            eng.train_iteration(it, nxt); it += 1
        ddp.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(k3):
            eng.train_iteration(it, nxt); it += 1
        torch.cuda.synchronize(); ddp.barrier()
        dth = time.perf_counter() - t1
        if world > 1:
            th = torch.tensor([dth], dtype=torch.float64, device=dev)
            dist.all_reduce(th, op=dist.ReduceOp.MAX)
            dth = th.item()
        host_feed = {'value': round(R.cfg.N_CRITIC * B * world * k3 / dth, 2), 'ms_per_step': round(1e3 * dth / k3, 3), 'steps': k3,
                     'h2d_bytes_per_step': R.cfg.N_CRITIC * (B * 3072 * 4 + B * 4),
                     'path': 'tflib.cifar10.EpochFeed (synthetic uint8, %d images) -> int32 pinned ring -> H2D copy stream, two iterations deep -> staging kernel' % n_img}

    roofline = None
    step_clock = None
    if not args.no_roofline and rank == 0:
        if eng.graphed and world == 1:
            step_clock, it = measure_step_clock(eng, it, next_batch, K, torch)
        roofline = measure_roofline(trainer, next_batch, K, torch, ms_per_step)
        if roofline is not None and world == 1:
            try:
                # the critic-step launch of the dominant kernel: its five launches of the iteration are the largest same-FLOPs group
                fl_c = roofline.pop('_critic_launch_flops', None)
                ins = measure_in_situ(trainer, eng, batches, K, torch, roofline['kernel'], fl_c, it) if fl_c else None
            except Exception as e:          # noqa: BLE001  (report, never fail the record over the cross-check)
                ins = {'error': '%s: %s' % (type(e).__name__, e)}
            if ins is not None and ins.get('us_per_launch'):
                br = roofline.pop('_critic_launch_bracket_us', None)
                ins['frac'] = round(ins['tflops'] / roofline['peak'], 4)
                ins['bracket_us_same_launch'] = round(br, 2) if br else None
                if br:
                    ins['bracket_vs_in_situ_pct'] = round(100.0 * (br - ins['us_per_launch']) / ins['us_per_launch'], 1)
                    ins['crosscheck'] = 'ok' if abs(ins['bracket_vs_in_situ_pct']) <= 10.0 else 'FAIL: the event bracket and the in-graph difference disagree by more than 10 %'
                if step_clock and step_clock.get('sclk_mhz'):
                    ins['frac_at_loop_clock'] = round(ins['tflops'] / (roofline['peak'] * step_clock['sclk_mhz'] / NOMINAL_MHZ), 4)
            roofline['in_situ'] = ins
            if ins and str(ins.get('crosscheck', '')).startswith('FAIL'):
                print('bench: roofline cross-check: %r' % (ins,), file=sys.stderr)
        if roofline is not None:
            roofline.pop('_critic_launch_flops', None); roofline.pop('_critic_launch_bracket_us', None)
            roofline['loop_clock'] = step_clock
    # the same loop with every layer on the fp32 MFMA family (no split-mode routing), for comparison: a second engine, same warm-up and
    # step counts, same clock (the comparison used to be 10 / 50 steps on wall clock against 20 / 100 with a p50)
    fp32_only = None
    if K.X3_HYBRID and K.MMA_DTYPE is None and rank == 0 and world == 1 and not args.no_roofline:
        K.X3_HYBRID = False
        try:
            K.clear_pack16_cache()                 # no split-mode launch can be routed: its packed filter images are not refreshed either
            eng2 = GraphedTrainer(trainer, use_graphs=not args.no_graph)
            for _ in range(args.warmup):
                eng2.train_iteration(it, next_batch); it += 1
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                eng2.train_iteration(it, next_batch); it += 1
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            fp32_only = {'value': round(R.cfg.N_CRITIC * B * args.steps / dt2, 2), 'ms_per_step': round(1e3 * dt2 / args.steps, 3), 'steps': args.steps,
                         'warmup': args.warmup, 'note': 'CTGAN_X3_HYBRID=0: every layer on the fp32 MFMA family'}
            del eng2
        finally:
            K.X3_HYBRID = True
            K.clear_pack16_cache()

    # the same loop with the critic step on the autograd path (Trainer.d_losses + two autograd calls: the penalty's first backward as a 64-row
    # chain of its own) instead of the hand-scheduled step: what merging the penalty rows into the main backward is worth in the step
    critic_ab = None
    import ctgan_amd.critic_schedule as CS
    if CS.MERGED_BWD and rank == 0 and world == 1 and not args.no_roofline:
        CS.MERGED_BWD = False
        try:
            eng3 = GraphedTrainer(trainer, use_graphs=not args.no_graph)
            k3 = max(10, args.steps // 2)
            for _ in range(5):
                eng3.train_iteration(it, next_batch); it += 1
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(k3):
                eng3.train_iteration(it, next_batch); it += 1
            torch.cuda.synchronize()
            ms_auto = 1e3 * (time.perf_counter() - t1) / k3
            critic_ab = {'autograd_path_ms_per_step': round(ms_auto, 3), 'hand_scheduled_ms_per_step': round(ms_per_step, 3), 'steps': k3,
                         'saved_ms_per_critic_step': round((ms_auto - ms_per_step) / R.cfg.N_CRITIC, 4)}
            del eng3
        finally:
            CS.MERGED_BWD = True

    gp_unit = None
    step_exec = None
    if not args.no_roofline and rank == 0:
        gp_unit = measure_gp_unit(trainer, batches[0], torch)
        if critic_ab is not None and isinstance(gp_unit, dict) and 'ms' in gp_unit:
            # the unit above is the penalty ALONE (forward + its own 64-row backward chain + double backward, as the autograd path runs it).
            # In the training step its first backward rides the launches of the main backward: the unit's cost inside the step = its
            # stand-alone time minus what the merged schedule saves per critic step (measured, same box, same run).
            ms_in = gp_unit['ms'] - critic_ab['saved_ms_per_critic_step']
            gf_by_pipe = gp_unit['gflop_executed_by_pipe']
            gp_unit['in_step'] = {'ms': round(ms_in, 4),
                                  'frac_executed': round(sum(gf / PIPE_PEAK[pp] for pp, gf in gf_by_pipe.items()) / ms_in, 4),
                                  'note': 'stand-alone ms - critic_step_ab.saved_ms_per_critic_step; same executed FLOPs'}
        if roofline is not None:
            gf = roofline['all_conv_kernels']['gflop_executed']
            ideal_ms = sum(v['gflop_executed'] / v['peak'] for v in roofline['by_pipe'].values())
            step_exec = {'gflop_executed': gf, 'achieved': round(gf / ms_per_step, 2), 'unit': 'TFLOP/s',
                         'frac_executed': round(ideal_ms / ms_per_step, 4),
                         'frac_note': 'time the launched conv FLOPs take at the peak of the pipe each kernel runs on / ms_per_step',
                         'conv_time_share': round(roofline['all_conv_kernels']['time_ms'] / ms_per_step, 3),
                         'by_pipe': {k: {'frac': v['frac'], 'share_of_step_time': v['share_of_step_time']} for k, v in roofline['by_pipe'].items()},
                         'note': 'conv-family FLOPs actually launched in one iteration (sum of 2*N*P*Q*K*R*S*C over the launches) / ms_per_step; '
                                 'the roofline fractions are per matrix pipe and time-weighted (roofline.by_pipe): fp32-pipe kernels against 157.3 '
                                 'TFLOP/s, split-mode kernels against 2500/6'}
    cpu = None
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        cpu = cpu_baseline(lib, torch)

    critic_scheduled = bool(CS.usable(R, None, trainer.rng, batches[0][0], torch.zeros(B, R.cfg.OUTPUT_DIM, device=dev)))
    # a finite-gradient guard fired during the timed loops = the run overflowed or diverged: not a valid measurement
    sane = sane and trainer.d_opt.skipped() == 0 and trainer.g_opt.skipped() == 0
    if rank == 0:
        rec = {
            'metric': 'img/s per (n_critic D + 1 G) step, CIFAR-10 ResNet',
            'value': round(value, 2), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'CT_gan_cifar_resnet.py ResNet G/D 32x32 CT-WGAN (GP+CT+ACGAN), batch 64/GPU, '
                                   'N_CRITIC=5 + 1 G step (128 samples) per step', 'global_batch': B * world,
                       'images_per_step': R.cfg.N_CRITIC * B * world, 'parallelism': 'dp%d' % world,
                       'hipgraph': bool(eng.graphed), 'last_d_cost': last_cost, 'last_d_terms': last, 'loss_sane': sane,
                       'backend': dist_info(world)[0], 'rccl_world': dist_info(world)[1], 'replicas_identical': replicas_identical,
                       'host_feed': host_feed, 'collective': collective,
                       'arithmetic': ('fp32 throughout; the large stride-1 conv layers (roofline.by_kernel: conv16x3h) compute the fp32 products as '
                                      'three-bf16-term splits on the bf16 matrix cores (six MFMAs per product, fp32 accumulate; error vs fp64 '
                                      'no larger than the fp32 MFMA family\'s, tests/test_gpu_kernels16.py, DESIGN 4.3), every other layer on '
                                      'v_mfma_f32_32x32x2_f32' if (K.X3_HYBRID and K.MMA_DTYPE is None) else 'fp32 MFMA (v_mfma_f32_32x32x2_f32) throughout'),
                       'fp32_mfma_only': fp32_only},
            'ms_per_step_p50': round(p50, 3),
            # (also at the top level, so that a parse of the line shows them: the host-fed loop, the all-fp32-MFMA number, Adam's skip count)
            'host_feed': host_feed, 'fp32_mfma_only': fp32_only,
            'adam_skipped_elements': {'critic': trainer.d_opt.skipped(), 'generator': trainer.g_opt.skipped()},
            'critic_step': 'hand-scheduled (critic_schedule.py: one backward chain over dropout-pass rows + penalty rows)' if critic_scheduled else 'autograd',
            'critic_step_ab': critic_ab,
            'step': step_exec,
            'step_effective_frac': round(ITER_GFLOP * 1e9 / (ms_per_step * 1e-3) / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
            'step_effective_frac_note': 'EFFECTIVE rate, not a roofline fraction: FLOPs of the REFERENCE formulation (SURVEY 8(d), 2990.5 GFLOP / iteration) / time / fp32 MFMA peak; the executed count is lower (resampled convs run as stride-2 convs with the spread filter): see step.frac_executed',
            'roofline': roofline, 'gp_unit': gp_unit, 'cpu_baseline': cpu, 'build': build_provenance(),
        }
        if args.leg is not None:
            rec = {'leg': args.leg, 'n_gpus': world, 'ms_per_step': rec['ms_per_step'], 'value': rec['value'], 'steps': args.steps, 'in_graph': bool(eng.ar_in_graph),
                   'split_flush': bool(trainer.split_flush), 'hipgraph': bool(eng.graphed), 'graph_error': eng.graph_error, 'it_graph_error': eng.it_graph_error,
                   'backend': dist_info(world)[0], 'replicas_identical': replicas_identical, 'loss_sane': sane,
                   'exposed_ms_per_step': (collective or {}).get('exposed_ms_per_step')}
        print(json.dumps(rec))
        sys.stdout.flush()               # the record is OUT before any A/B leg starts
    if world > 1:
        legs = [] if (args.no_ab_legs or not sane) else [n for n, l in sorted(LEGS.items()) if (dist_info(world)[0] or 'nccl') in l['backends']]
        ports = pick_ports(len(legs), rank, dev)
        dist.barrier()
        dist.destroy_process_group()
        if legs:
            run_ab_legs(args, rank, world, legs, ports, max(5, min(20, args.steps)))
    if not sane:
        print('bench: critic loss terms out of band %r (replicas identical: %r) - the timed loop is not computing the reference step' % (last, replicas_identical), file=sys.stderr)
        sys.exit(3)


# A/B legs of an N > 1 run: each a FRESH group of N rank processes, started by the ranks of the main run after the record is printed and
# their own process group is gone, killed after --leg-timeout.  in_graph*: the gradient all-reduces captured as nodes of the step graphs
# (engine.GraphedTrainer(ar_in_graph=True)) - RCCL only, never run on more than one device so far; *split_flush: blocks 1-2 of the critic
# bucket on their own all-reduce under the rest of the backward (Trainer.split_flush; with eager collectives only the eager engine overlaps).
LEGS = {
    'in_graph': {'ar_in_graph': True, 'split_flush': False, 'no_graph': False, 'backends': ('nccl',)},
    'in_graph_split_flush': {'ar_in_graph': True, 'split_flush': True, 'no_graph': False, 'backends': ('nccl',)},
    'eager_split_flush': {'ar_in_graph': False, 'split_flush': True, 'no_graph': True, 'backends': ('gloo',)},
    'spawn': {'ar_in_graph': False, 'split_flush': False, 'no_graph': False, 'backends': ()},        # --spawn-check's leg (launcher tests, no GPU)
}


def pick_ports(n, rank, dev=None):
    """n free TCP ports chosen by rank 0 and broadcast (the legs' rendezvous ports); needs the process group still alive."""
    import socket
    import torch.distributed as dist
    ports = [0] * n
    if rank == 0:
        socks = []
        for i in range(n):
            sk = socket.socket(); sk.bind(('127.0.0.1', 0)); socks.append(sk)
            ports[i] = sk.getsockname()[1]
        for sk in socks:
            sk.close()
    if n:
        box = [ports]
        dist.broadcast_object_list(box, src=0, device=dev)
        ports = box[0]
    return ports


def run_ab_legs(args, rank, world, legs, ports, steps, extra=()):
    """Every rank of the finished main run starts ITS rank of each leg as a child process (same RANK / LOCAL_RANK / WORLD_SIZE, the leg's
    own MASTER_PORT), waits at most --leg-timeout and kills exactly that child on expiry.  Works the same under torch.distributed.run
    (the driver's launch) and under this script's own launcher.  Rank 0 relays each leg's short record to STDERR: stdout carries the one
    record line only, and that line is already out."""
    import subprocess
    for name, port in zip(legs, ports):
        env = {k: v for k, v in os.environ.items() if not k.startswith('TORCHELASTIC_')}     # (the agent's store is not the leg's store)
        env.update(MASTER_PORT=str(port), MASTER_ADDR='127.0.0.1')
        cmd = [sys.executable, os.path.abspath(__file__), '--gpus', str(world), '--steps', str(steps), '--warmup', '3', '--leg', name,
               '--launcher', 'never'] + (['--backend', args.backend] if args.backend else []) + list(extra)
        t0 = time.time()
        pr = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, stderr=sys.stderr)
        try:
            out, _ = pr.communicate(timeout=args.leg_timeout)
            res = None
            for ln in (out or b'').decode(errors='replace').splitlines():
                if ln.startswith('{'):
                    res = json.loads(ln)
            if res is None:
                res = {'leg': name, 'error': 'rank 0 of the leg exited with status %s and no record' % pr.returncode}
        except subprocess.TimeoutExpired:
            pr.kill()
            pr.communicate()
            res = {'leg': name, 'error': 'killed after --leg-timeout %.0f s' % args.leg_timeout}
        if rank == 0:
            res['wall_s'] = round(time.time() - t0, 1)
            print('bench: leg %s: %s' % (name, json.dumps(res)), file=sys.stderr)
            sys.stderr.flush()


def launch_ranks(args, argv):
    """The launcher side of `python bench.py --gpus N`: start N rank processes of this script (one per GPU), relay rank 0's
    stdout (the JSON line), exit status = the first non-zero rank status.  This process never initialises the GPU: it only counts
    devices (torch.cuda.device_count() does not create a HIP context on this image) - a process that has touched the GPU must not
    start other programs in its place, and this one does not exec, it waits for its children."""
    import socket
    import subprocess
    n = args.gpus
    if n < 1:
        print('bench: --gpus must be >= 1', file=sys.stderr)
        return 2
    backend = args.backend or 'nccl'
    if not args.spawn_check:
        import torch
        ndev = torch.cuda.device_count()
        if ndev < 1 or (backend != 'gloo' and ndev < n):
            print('bench: --gpus %d but %d device(s) visible - two RCCL ranks cannot share a GPU (use --backend gloo to exercise the '
                  'multi-rank code path on one device)' % (n, ndev), file=sys.stderr)
            return 2
    # A rendezvous that fails (the port picked by bind-close-reuse can be taken in between) makes the ranks exit with RENDEZVOUS_EXIT:
    # the launch is repeated on another port.  SIGTERM / SIGINT to the launcher (a driver timeout) end the ranks too: without that they
    # would sit in a collective, holding the GPUs, until the RCCL timeout.
    import signal
    import threading
    import time as _time

    def _term(signum, frame):
        raise SystemExit(128 + signum)
    old_handlers = {sg: signal.signal(sg, _term) for sg in (signal.SIGTERM, signal.SIGINT)}
    status, chunks = 0, []
    try:
        for attempt in range(3):
            sock = socket.socket()
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
            sock.close()
            procs, chunks = [], []
            reader = None                    # (assigned once every rank is started: a failed Popen / a signal in between must not hit an unbound name)
            try:
                for r in range(n):
                    env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
                    # rank 0's stdout carries the record; the other ranks' stdout (library banners, diagnostics) goes to the launcher's stderr
                    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                                  stdout=subprocess.PIPE if r == 0 else sys.stderr))
                # rank 0's stdout is drained by a thread (a full pipe would block it); the launcher polls ALL ranks: the first rank that
                # dies with a non-zero status ends the run - the others would otherwise sit in a collective until the RCCL timeout
                def relay(pr=procs[0]):
                    # rank 0's lines go out as they arrive: the record must be on the launcher's stdout BEFORE the A/B legs start (a hang there
                    # ends in a kill by the driver; the record is out by then).  stdout carries records only; library chatter
                    # (gloo / RCCL print banners to stdout) -> stderr
                    for raw in iter(pr.stdout.readline, b''):
                        line = raw.decode(errors='replace').rstrip('\n')
                        print(line, file=sys.stdout if line.startswith('{') else sys.stderr)
                        (sys.stdout if line.startswith('{') else sys.stderr).flush()
                reader = threading.Thread(target=relay, daemon=True)
                reader.start()
                status = 0
                while True:
                    codes = [pr.poll() for pr in procs]
                    bad = [c for c in codes if c not in (None, 0)]
                    if bad:
                        status = bad[0]
                        break
                    if all(c == 0 for c in codes):
                        break
                    _time.sleep(0.05)
            finally:
                for pr in procs:                 # end exactly the processes started here (never by pattern)
                    if pr.poll() is None:
                        pr.kill()
                for pr in procs:
                    pr.wait()
                if reader is not None:
                    reader.join(timeout=5)
            if status != RENDEZVOUS_EXIT:
                break
            print('bench: rendezvous on port %d failed, retrying on another port' % port, file=sys.stderr)
    finally:
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    sys.stdout.flush()
    return status


RENDEZVOUS_EXIT = 75      # exit status of a rank whose torch.distributed rendezvous failed (launch_ranks retries on another port)


def init_ranks(backend):
    """ddp.init_from_env, with a failed rendezvous turned into RENDEZVOUS_EXIT for the launcher."""
    from ctgan_amd import ddp
    try:
        return ddp.init_from_env(backend=backend)
    except Exception as e:      # noqa: BLE001  (DistNetworkError / RuntimeError, by torch version)
        import torch.distributed as dist
        net = isinstance(e, getattr(dist, 'DistNetworkError', ())) or 'address already in use' in str(e).lower() or 'EADDRINUSE' in str(e)
        if net and int(os.environ.get('WORLD_SIZE', '1')) > 1:
            print('bench: rank %s: rendezvous failed: %r' % (os.environ.get('RANK'), e), file=sys.stderr)
            sys.exit(RENDEZVOUS_EXIT)
        raise


def dist_info(world):
    """(backend, rccl_world): rccl_world is the world size ONLY when the collectives really are RCCL."""
    import torch.distributed as dist
    backend = dist.get_backend() if (world > 1 and dist.is_initialized()) else None
    return backend, (world if backend == 'nccl' else None)


def spawn_check(args):
    """Launcher self-test (no GPU): every rank joins the group, the ranks' numbers are summed, rank 0 reports."""
    import torch
    import torch.distributed as dist

    from ctgan_amd import ddp
    if os.environ.get('CTGAN_TEST_DIE_RANK') == os.environ.get('RANK'):       # launcher test: this rank dies before the rendezvous
        sys.exit(7)
    if os.environ.get('CTGAN_TEST_HANG_RANK') == os.environ.get('RANK'):      # launcher test: this rank never joins (a stuck collective)
        import time
        open(os.environ['CTGAN_TEST_PID_FILE'] + '.' + os.environ['RANK'], 'w').write(str(os.getpid()))
        time.sleep(600)
    if args.leg is not None and os.environ.get('CTGAN_TEST_HANG_LEG') == os.environ.get('RANK'):   # launcher test: this rank of the LEG never joins
        import time
        time.sleep(600)
    rank, world, local = init_ranks(args.backend or 'gloo')
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
    backend, rccl = dist_info(world)
    if rank == 0:
        rec = {'spawn_check': True, 'n_gpus': world, 'rank_sum': t.item(), 'backend': backend, 'rccl_world': rccl,
               'local_rank': local, 'master_port': os.environ.get('MASTER_PORT')}
        if args.leg is not None:
            rec['leg'] = args.leg
        print(json.dumps(rec))
        sys.stdout.flush()
    if world > 1:
        # the A/B-leg mechanism of the training bench, exercised without a GPU (tests/test_bench_launcher.py): fresh rank groups AFTER the record
        legs = ['spawn'] if (args.leg is None and not args.no_ab_legs and os.environ.get('CTGAN_TEST_SPAWN_LEG')) else []
        ports = pick_ports(len(legs), rank)
        dist.barrier()
        dist.destroy_process_group()
        if legs:
            if not args.backend:
                args.backend = 'gloo'
            run_ab_legs(args, rank, world, legs, ports, 1, extra=['--spawn-check'])


def run_unconditional(args):
    """BASELINE.json configs[1] / configs[4]: the unconditional CT-WGAN step (dcgan_step.DCGANTrainer) under hipGraph replay,
    same timing contract and JSON line as the headline config."""
    import importlib

    import numpy as np
    import torch
    import torch.distributed as dist

    import ctgan_amd.kernels as K
    import ctgan_amd.tflib as lib
    from ctgan_amd import ddp
    from ctgan_amd.dcgan_step import DCGANTrainer
    from ctgan_amd.engine import GraphedDCGANTrainer
    modname, dtype, workload = ALT_CONFIGS[args.config]
    rank, world, local = init_ranks(args.backend)
    if local >= torch.cuda.device_count():
        if world > 1 and dist.get_backend() != 'gloo':
            raise SystemExit('two RCCL ranks cannot share a GPU')
        local = local % max(torch.cuda.device_count(), 1)
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: the record would not say what ran' % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    M = importlib.import_module('ctgan_amd.' + modname)
    lib.delete_all_params(); lib.set_seed(0); lib.set_device(None)
    M.configure()
    B = M.cfg.BATCH_SIZE
    if hasattr(M, 'build_params'):
        M.build_params(dev)
    else:
        with torch.no_grad():
            M.Discriminator(M.Generator(2, noise=torch.zeros(2, 128, device=dev)), u=[torch.ones(2, *s, device=dev) for s in M.feat_shapes()])
    K.set_mma_dtype(dtype)
    side = torch.cuda.Stream() if world > 1 else None
    tr = DCGANTrainer(M, seed=2024, rank=rank, world_size=world, allreduce=ddp.FlatAllReduce(side_stream=side))
    if dtype == 'f16':
        tr.loss_scale = 1024.0              # power-of-two loss scale of the fp16 mode (dcgan_step.DCGANTrainer.loss_scale)
    ddp.broadcast_params([tr.d_opt.theta, tr.g_opt.theta])
    nrng = np.random.default_rng(1234 + rank)
    batches = [torch.from_numpy(nrng.integers(0, 256, (B, M.cfg.OUTPUT_DIM), dtype=np.int32)).to(dev) for _ in range(8)]
    cursor = [0]

    def next_batch():
        cursor[0] = (cursor[0] + 1) % len(batches)
        return batches[cursor[0]]
    eng = GraphedDCGANTrainer(tr, (B, M.cfg.OUTPUT_DIM), torch.int32, use_graphs=not args.no_graph)
    if eng.graph_error and rank == 0:
        print('hipGraph capture failed, running eager: ' + eng.graph_error, file=sys.stderr)
    it = 1
    for _ in range(args.warmup):
        eng.train_iteration(it, next_batch); it += 1
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    ddp.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        out = eng.train_iteration(it, next_batch); it += 1
        marks[k + 1].record()
    torch.cuda.synchronize(); ddp.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    last = {k: float(out[k].item()) for k in ('cost', 'wgan_only', 'ct', 'gp') if out.get(k) is not None}
    sane = all(v == v and abs(v) < 1e4 for v in last.values())
    ms_per_step = 1e3 * dt / args.steps
    n_crit = M.cfg.CRITIC_ITERS
    roofline = None
    if not args.no_roofline and rank == 0:
        peak = PEAK_16BIT_MFMA_TFLOPS if dtype else PEAK_F32_MFMA_TFLOPS
        saved_world, tr.world = tr.world, 1
        try:
            tr.train_iteration(1, next_batch)
            torch.cuda.synchronize()
            K.PROFILE, K.PROFILE_REPS = [], 4
            try:
                tr.train_iteration(1, next_batch)
                torch.cuda.synchronize()
                prof = K.PROFILE
            finally:
                K.PROFILE, K.PROFILE_REPS = None, 1
        finally:
            tr.world = saved_world
        agg = {}
        for name, flops, e0, e1, reps, _shape, _sym in prof:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1; a[1] += flops; a[2] += e0.elapsed_time(e1) * 1e-3 / reps
        total_t = sum(a[2] for a in agg.values()); total_f = sum(a[1] for a in agg.values())
        mm = {k: v for k, v in agg.items() if ('16' in k) == bool(dtype)} or agg     # the dominant kernel of the family this config is about
        name, (cnt, fl, tt) = max(mm.items(), key=lambda kv: kv[1][2])
        traffic = None
        try:        # HBM-side bytes of this kernel from the committed PMC passes (tools/pmc_run16.sh), scaled by FLOPs to this launch mix
            pmc = json.load(open(_profile_path('r02_pmc_traffic_conv16.json'))).get(name)
            if pmc:
                k = (fl / cnt) / pmc['flops_per_launch']
                traffic = {'hbm_bytes_per_launch': round(pmc['hbm_bytes_per_launch'] * k), 'algorithmic_bytes_per_launch': round(pmc['algorithmic_bytes_per_launch'] * k),
                           'source': 'profiles/history/r02_pmc_traffic_conv16.json (%s; FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, scaled to this launch mix)' % pmc['geometry']}
        except Exception:
            pass
        roofline = {'bound': 'mfma', 'kernel': name, 'launches': cnt, 'flops_per_launch': round(fl / cnt / 1e9, 3),
                    'avg_launch_us': round(tt / cnt * 1e6, 2), 'achieved': round(fl / tt / 1e12, 2), 'peak': peak, 'unit': 'TFLOP/s',
                    'frac': round(fl / tt / 1e12 / peak, 4), 'traffic': traffic,
                    'all_conv_kernels': {'time_ms': round(total_t * 1e3, 3), 'gflop_executed': round(total_f / 1e9, 2), 'launches': len(prof)},
                    'by_kernel': {k: {'launches': v[0], 'tflops': round(v[1] / v[2] / 1e12, 2), 'ms': round(v[2] * 1e3, 3)}
                                  for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])}}
    if rank == 0:
        print(json.dumps({
            'metric': 'img/s per (n_critic D + 1 G) step', 'value': round(n_crit * B * world * args.steps / dt, 2), 'unit': 'img/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
            'ms_per_step_p50': round(per_step[len(per_step) // 2], 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': dtype or 'f32', 'data': 'synthetic',
            'config': {'workload': workload, 'name': args.config, 'global_batch': B * world, 'images_per_step': n_crit * B * world,
                       'parallelism': 'dp%d' % world, 'hipgraph': bool(eng.graphed), 'last_d_terms': last, 'loss_sane': sane,
                       'loss_scale': tr.loss_scale, 'backend': dist_info(world)[0], 'rccl_world': dist_info(world)[1],
                       'adam_skipped_elements': {'critic': tr.d_opt.skipped(), 'generator': tr.g_opt.skipped()}},
            'roofline': roofline, 'cpu_baseline': None, 'build': build_provenance()}))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    if not sane:
        print('bench: critic loss terms out of band %r' % (last,), file=sys.stderr)
        sys.exit(3)


def pipe_of(variant):
    """'bf16x6' for the split-mode kernels (fp32 products as six bf16 MFMAs: peak = dense bf16 peak / 6), 'bf16' / 'f16' for the plain
    16-bit kernels, else 'f32' (v_mfma_f32_32x32x2_f32 or packed-fp32 FMA kernels: priced against the fp32 MFMA peak)."""
    family = variant.split('<')[0].split('(')[0]        # the kernel FAMILY, not a substring: 'igemm_fwd_pipe<32x32,k4>' contains "x3" 
    if family.startswith(('conv16x3', 'wgrad16x3', 'chain8x8')):
        return 'bf16x6'
    if family.startswith(('conv16', 'wgrad16')):
        return '16bit'
    return 'f32'


PIPE_PEAK = {'bf16x6': PEAK_16BIT_MFMA_TFLOPS / 6.0, '16bit': PEAK_16BIT_MFMA_TFLOPS, 'f32': PEAK_F32_MFMA_TFLOPS}
NOMINAL_MHZ = 2400.0                  # the clock the guide's peaks are quoted at


def measure_in_situ(trainer, eng, batches, K, torch, sym, flops_critic, it):
    """The dominant kernel's time WHERE IT RUNS - inside the replayed critic-step graph, behind the step's own launches, at the clock the
    loop sustains: a second capture of the same step with the filter-column weight-gradient kernel launched twice (K.WGRAD_GROUP_EXTRA;
    same operands, same result), replayed in alternation with the step's own graph; the difference of the two replay times is one launch.
    Cross-checks the HIP-event bracket of the instrumented eager iteration (0.427 on the driver's line vs 0.540 in
    profiles/ with nothing in the record to tell a slow box from a mis-timed bracket)."""
    import ctgan_amd.engine as E
    from ctgan_amd.engine import GraphedTrainer
    if not sym.startswith('wgrad16c_group_kernel') or not eng.graphed or eng.d_graph is None:
        return None
    old_it = E.ITERATION_GRAPH
    try:
        E.ITERATION_GRAPH = False
        K.WGRAD_GROUP_EXTRA = 1
        eng2 = GraphedTrainer(trainer, use_graphs=True, ar_in_graph=False)
    finally:
        K.WGRAD_GROUP_EXTRA = 0
        E.ITERATION_GRAPH = old_it
    if not eng2.graphed:
        return {'error': eng2.graph_error}
    real, labels = batches[0]
    fake = trainer.generate_fakes(labels)[0]
    n, rounds = 10, 7
    times = {0: [], 1: []}
    for r in range(rounds + 1):
        for which, e in ((0, eng), (1, eng2)):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e._stage_d_inputs(real, labels, fake)
            e0.record()
            for _ in range(n):
                e.d_graph.replay()
            e1.record()
            torch.cuda.synchronize()
            if r:                                   # (round 0: warm-up of both graphs)
                times[which].append(e0.elapsed_time(e1) * 1e3 / n)
    trainer.d_opt.t += 2 * n * (rounds + 1)
    eng._weights_moved('Discriminator')
    del eng2
    m0 = sorted(times[0])[rounds // 2]; m1 = sorted(times[1])[rounds // 2]
    us = m1 - m0
    return {'us_per_launch': round(us, 2), 'critic_step_graph_us': round(m0, 2), 'with_second_launch_us': round(m1, 2), 'replays': n * rounds,
            'tflops': round(flops_critic / us / 1e6, 2) if us > 0 else None,
            'what': 'median replay time of the critic-step graph captured with the kernel launched twice - median of the step\'s own graph (alternating, 7 x 10 replays each)'}


def measure_step_clock(eng, it, next_batch, K, torch, iters=8):
    """Average shader clock over `iters` replayed iterations (one K.ClockProbe wave on a side stream around the loop)."""
    eng.train_iteration(it, next_batch); it += 1
    torch.cuda.synchronize()
    with K.ClockProbe(cap_ms=1500.0) as pr:
        for _ in range(iters):
            eng.train_iteration(it, next_batch); it += 1
    torch.cuda.synchronize()
    mhz, us, seen = pr.result()
    return ({'sclk_mhz': round(mhz, 1), 'iterations': iters, 'probe_ms': round(us / 1e3, 2), 'nominal_mhz': NOMINAL_MHZ,
             'note': 's_memtime cycles / s_memrealtime (100 MHz) of a one-wave probe spanning the replayed iterations: the average shader clock of the loop'}
            if seen else {'error': 'probe gave up before the loop ended'}), it


def _profile_path(fn):
    """profiles/<fn>, or profiles/history/<fn> (rounds 1-4 were moved there in round 6)."""
    p = os.path.join(ROOT, 'profiles', fn)
    return p if os.path.exists(p) else os.path.join(ROOT, 'profiles', 'history', fn)


def load_pmc_traffic():
    """profiles/r03_pmc_traffic_x3.json (tools/pmc_x3.sh: separate rocprofv3 --pmc passes, gfx950 corrections of the guide) + the
    round-1 files of the fp32 tiles, keyed by device symbol."""
    out = {}
    for fn in ('r03_pmc_traffic_x3.json', 'r04_pmc_traffic_x3.json', 'r04_pmc_wgrad_col.json', 'r05_pmc_traffic_x3.json', 'r05_pmc_wgrad_col.json',
               'r06_pmc_traffic_x3.json', 'r06_pmc_wgrad_col.json'):      # (later files override earlier ones per symbol; *_wgrad_col: tools/pmc_wgrad_col.sh)
        try:
            path = _profile_path(fn)
            for sym, rec in json.load(open(path)).items():
                if isinstance(rec, dict) and 'hbm_bytes_per_launch' in rec and 'flops_per_launch' in rec:
                    out[sym] = dict(rec, file=os.path.relpath(path, ROOT), measured_in_round=int(fn[1:3]))
        except Exception:
            pass
    for fn, sym in (('r01_pmc_traffic_64x128.json', 'igemm_fwd_pipe_kernel<1, 4, 1, 2, 1, 1, false, 1>'),
                    ('r01_pmc_traffic.json', 'igemm_fwd_pipe_kernel<2, 2, 1, 2, 2, 1, false, 1>')):
        try:
            rec = json.load(open(_profile_path(fn)))
            out.setdefault(sym, dict(rec, file=os.path.relpath(_profile_path(fn), ROOT), measured_in_round=1))
        except Exception:
            pass
    return out


def measure_roofline(trainer, next_batch, K, torch, ms_per_step=None):
    """One instrumented EAGER iteration of the SAME launch mix the graph replays (grouped weight gradients included): every
    conv-family launch is bracketed by HIP events on its launch stream (4 back-to-back repeats per bracket, time / 4: a
    single-launch bracket carries ~10 us of event overhead and disagrees with rocprofv3's kernel durations).  Kernels are keyed by
    their device symbol as rocprofv3 prints it (ctgan_last_symbol), so every row can be looked up in the committed summaries under
    profiles/.  The dominant kernel = the symbol with the largest summed duration."""
    # rank 0 only: this pass must not enter a collective (the other ranks are not here)
    saved_world, trainer.world = trainer.world, 1
    try:
        trainer.train_iteration(1, next_batch)         # eager warm-up (lazy allocations)
        torch.cuda.synchronize()
        K.PROFILE = []
        K.PROFILE_CLOCKS = []
        K.PROFILE_REPS = 4          # each (idempotent) conv launch runs 4x inside its event bracket: amortises the event overhead
        try:
            trainer.train_iteration(1, next_batch)
            torch.cuda.synchronize()
            prof = K.PROFILE
            clocks = [(sym_, fl_) + pr.result() for sym_, fl_, pr in K.PROFILE_CLOCKS]
        finally:
            K.PROFILE = None
            K.PROFILE_CLOCKS = []
            K.PROFILE_REPS = 1
    finally:
        trainer.world = saved_world
    agg = {}
    wide = {}          # few-channel kernels (csrc/fewch.hip) stream the wide tensor once: they are rated against HBM, not against a matrix pipe
    # launches of one symbol on one problem (same shape, same FLOPs) are repeats of the same measurement - the five critic steps of the
    # iteration: each such group enters with its MEDIAN bracket (one slow bracket must not move the line)
    same = {}
    for name, flops, e0, e1, reps, _shape, sym in prof:
        same.setdefault((sym, round(flops), tuple(_shape)), []).append(e0.elapsed_time(e1) * 1e-3 / reps)
    med = {k: sorted(v)[len(v) // 2] for k, v in same.items()}
    for name, flops, e0, e1, reps, _shape, sym in prof:
        a = agg.setdefault(sym, [0, 0.0, 0.0, set()])
        a[0] += 1; a[1] += flops; a[2] += med[(sym, round(flops), tuple(_shape))]
        a[3].add(name.replace(',ph4', '').split(',split')[0].rstrip('>') + ('>' if '<' in name else ''))
        if name.startswith('fewch') and len(_shape) == 8:
            n_, c_, h_, w_, k_, _r, st_, _up = _shape
            px = (h_ // st_) * (w_ // st_) if k_ > c_ else h_ * w_          # pixels of the wide side (output grid when it is the output)
            wide[sym] = wide.get(sym, 0.0) + 4.0 * n_ * px * max(c_, k_)
    # shader clock per symbol: the probe's cycles / its wall time, summed over the symbol's brackets whose probe ended on the flag
    clk = {}
    for sym_, fl_, mhz, us, seen in clocks:
        if seen and us > 0:
            c = clk.setdefault(sym_, [0.0, 0.0])
            c[0] += mhz * us; c[1] += us
    clk = {k: v[0] / v[1] for k, v in clk.items() if v[1] > 0}
    if not agg:
        return None
    total_t = sum(a[2] for a in agg.values())
    total_f = sum(a[1] for a in agg.values())
    sym, (cnt, fl, tt, variants) = max(agg.items(), key=lambda kv: kv[1][2])
    pipe = pipe_of(sorted(variants)[0])
    peak = PIPE_PEAK[pipe]
    achieved = fl / tt / 1e12
    pmc = load_pmc_traffic()

    def traffic_of(symbol, flops_per_launch):
        rec = pmc.get(symbol)
        if not rec:
            return None
        k = flops_per_launch / rec['flops_per_launch']
        out = {'hbm_bytes_per_launch': round(rec['hbm_bytes_per_launch'] * k), 'algorithmic_bytes_per_launch': round(rec['algorithmic_bytes_per_launch'] * k),
               'traffic_over_algorithmic': round(rec['hbm_bytes_per_launch'] / rec['algorithmic_bytes_per_launch'], 2),
               'measured_in_round': rec.get('measured_in_round'),      # the round whose kernel code / routing the counters were taken on
               'source': '%s (%s; FETCH_SIZE and WRITE_SIZE from separate rocprofv3 --pmc passes, scaled by FLOPs to this launch mix)'
                         % (rec['file'], rec.get('geometry', 'geometry in the file'))}
        for key in ('mfma_busy_frac', 'valu_active_frac', 'lds_wait_frac', 'l2_hit_rate'):
            if key in rec:
                out[key] = rec[key]
        return out

    # time-weighted roofline fractions per matrix pipe (a mixed FLOP sum over one peak is not a fraction of anything)
    pipes = {}
    for k, v in agg.items():
        pp = pipes.setdefault(pipe_of(sorted(v[3])[0]), [0.0, 0.0, 0])
        pp[0] += v[1]; pp[1] += v[2]; pp[2] += v[0]
    by_pipe = {k: {'launches': v[2], 'time_ms': round(v[1] * 1e3, 3), 'gflop_executed': round(v[0] / 1e9, 2), 'achieved': round(v[0] / v[1] / 1e12, 2),
                   'peak': round(PIPE_PEAK[k], 1), 'frac': round(v[0] / v[1] / 1e12 / PIPE_PEAK[k], 4),
                   'share_of_conv_time': round(v[1] / total_t, 3),
                   'share_of_step_time': round(v[1] * 1e3 / ms_per_step, 3) if ms_per_step else None}
               for k, v in sorted(pipes.items(), key=lambda kv: -kv[1][1])}
    # the dominant symbol's most frequent problem = its critic-step launch (five per iteration)
    grp = max(((k_, ts) for k_, ts in same.items() if k_[0] == sym), key=lambda kv: (len(kv[1]), kv[0][1]))
    sclk = clk.get(sym)
    brackets = sorted(1e6 * t for (s_, f_, sh_), ts in same.items() if s_ == sym for t in ts)
    return {
        'bound': 'mfma', 'kernel': sym, 'variant': sorted(variants), 'pipe': pipe, 'launches': cnt,
        'flops_per_launch': round(fl / cnt / 1e9, 3), 'avg_launch_us': round(tt / cnt * 1e6, 2),
        'achieved': round(achieved, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
        'frac': round(achieved / peak, 4), 'traffic': traffic_of(sym, fl / cnt),
        # the shader clock this kernel ran at inside its brackets (K.ClockProbe: s_memtime cycles / s_memrealtime) and the fraction against
        # the peak AT that clock: peak is quoted at 2.4 GHz, the chip clocks to its power budget (MI355X_MICROARCH.md, "DVFS give-back")
        'sclk_mhz': round(sclk, 1) if sclk else None,
        'frac_at_measured_clock': round(achieved / (peak * sclk / NOMINAL_MHZ), 4) if sclk else None,
        '_critic_launch_flops': float(grp[0][1]), '_critic_launch_bracket_us': 1e6 * med[grp[0]],
        'bracket_us': {'n': len(brackets), 'min': round(brackets[0], 2), 'median': round(brackets[len(brackets) // 2], 2), 'max': round(brackets[-1], 2),
                       'note': 'every HIP-event bracket of this symbol in the instrumented iteration; launches on the same problem enter avg_launch_us with their median'},
        'peak_note': ('dense bf16 MFMA peak / 6: a split-mode kernel issues six bf16 MFMAs per fp32 product' if pipe == 'bf16x6' else
                      ('dense 16-bit MFMA peak' if pipe == '16bit' else 'fp32 MFMA peak')),
        'kernel_share_of_conv_time': round(tt / total_t, 3),
        'by_pipe': by_pipe,
        'all_conv_kernels': {'time_ms': round(total_t * 1e3, 3), 'launches': len(prof), 'gflop_executed': round(total_f / 1e9, 2),
                             'achieved': round(total_f / total_t / 1e12, 2),
                             'note': 'mixed pipes: see by_pipe for the roofline fractions'},
        'by_kernel': {k: {'launches': v[0], 'avg_launch_us': round(v[2] / v[0] * 1e6, 2), 'tflops': round(v[1] / v[2] / 1e12, 2), 'ms': round(v[2] * 1e3, 3),
                          'frac': round(v[1] / v[2] / 1e12 / PIPE_PEAK[pipe_of(sorted(v[3])[0])], 4), 'pipe': pipe_of(sorted(v[3])[0]),
                          'sclk_mhz': round(clk[k], 1) if k in clk else None,
                          'variant': sorted(v[3]), 'traffic': traffic_of(k, v[1] / v[0]),
                          **({'hbm': {'wide_tensor_bytes_per_launch': round(wide[k] / v[0]), 'achieved_GBps': round(wide[k] / v[2] / 1e9, 1),
                                      'peak_GBps': 8000.0, 'frac': round(wide[k] / v[2] / 8e12, 4),
                                      'note': 'HBM-bound kernel: bytes of the wide (128-channel) tensor / time; the MFMA fraction above does not bound it'}}
                             if k in wide else {})}
                      for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])},
        'note': 'by_kernel keys are device symbols: look them up in profiles/r06_kernel_stats_resnet.txt / r06_steady_state_resnet.txt '
                '(tools/roofline_crosscheck.py prints both side by side)',
    }


def measure_gp_unit(trainer, batch, torch):
    """The north star's primary unit: critic forward + gradient-penalty backward at B=64
    = forward (F_D) + data gradient to x_hat (F_D) + double backward (conv(ggx,W) and wgrad(ggx,gy): 2 F_D)
    = 4*B*F_D = 139.30 GFLOP; target <= 1.48 ms (60 % of the fp32 MFMA peak)."""
    import ctgan_amd.functional as F
    import ctgan_amd.gan_cifar_resnet as R
    B = R.cfg.BATCH_SIZE
    real_int, labels = batch
    x = torch.randn(B, R.cfg.OUTPUT_DIM, device=real_int.device).mul_(0.5)

    import ctgan_amd.tflib as lib

    def unit():
        lib.bump_epoch()                      # as in a training step: the weights changed since the last unit ...
        F.prepare_filters()                   # ... so every derived filter layout is rebuilt (one or two launches)
        trainer.rng.begin_step()
        xi = x.detach().requires_grad_(True)
        gp, _, _ = R.gradient_penalty_branch(xi, labels, trainer.rng)       # exactly the branch the critic step runs
        with F.deferred_wgrads():
            return torch.autograd.grad(gp, trainer.d_params, allow_unused=True)

    try:
        for _ in range(2):
            unit()
        torch.cuda.synchronize()
        import ctgan_amd.kernels as K
        K.PROFILE = []                          # one instrumented eager pass: the FLOPs this unit actually launches, per matrix pipe
        try:
            unit()
            torch.cuda.synchronize()
            executed = sum(p[1] for p in K.PROFILE) / 1e9
            by_pipe = {}
            for p in K.PROFILE:
                by_pipe[pipe_of(p[0])] = by_pipe.get(pipe_of(p[0]), 0.0) + p[1] / 1e9
        finally:
            K.PROFILE = None
        from ctgan_amd.engine import _capture_kw, quiesce_collectives
        quiesce_collectives()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, **_capture_kw()):
            unit()
        reps = 20
        graph.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
    except Exception as e:          # report, never fail the bench over the sub-benchmark
        return {'error': '%s: %s' % (type(e).__name__, e)}
    gflop = 4 * B * 544.148e-3
    return {'what': 'critic forward + GP backward (dD/dx_hat, then d(GP)/d(theta)), B=64, hipGraph replay',
            'ms': round(ms, 4), 'target_ms': 1.48,
            'gflop_executed': round(executed, 2), 'achieved': round(executed / ms, 2), 'unit': 'TFLOP/s',
            'frac_executed': round(sum(gf / PIPE_PEAK[pp] for pp, gf in by_pipe.items()) / ms, 4),
            'gflop_executed_by_pipe': {pp: round(gf, 2) for pp, gf in by_pipe.items()},
            'gflop_reference_formulation': round(gflop, 2), 'effective': round(gflop / ms, 2),
            'effective_frac': round(gflop / ms / PEAK_F32_MFMA_TFLOPS, 4), 'target_frac': 0.60,
            'note': 'frac_executed = (time the launched FLOPs take at the peak of the pipe each kernel runs on: fp32 MFMA 157.3, split mode '
                    '2500/6 TFLOP/s) / measured time - the roofline fraction (ConvMeanPool runs as a 4x4 stride-2 conv: 2.25x fewer MACs on '
                    '69 % of F_D); effective_frac prices the same time with the reference formulation\'s 139.3 GFLOP at the fp32 MFMA peak'}


def build_provenance():
    """Which sources the loaded libctgan_hip.so was built from (stamp written by __graft_entry__.build) and whether they are the
    sources in this tree."""
    try:
        import __graft_entry__ as ge
        stamp = json.load(open(os.path.join(ROOT, 'ctgan_amd', 'libctgan_hip.build.json')))
        return {'sources_sha256': stamp['sources_sha256'][:16], 'built_at': stamp['built_at'], 'matches_tree': stamp['sources_sha256'] == ge.source_digest()}
    except Exception as e:
        return {'error': '%s: %s' % (type(e).__name__, e)}


def host_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU boxes
    expose 256 logical CPUs but grant a 16-CPU quota; 256 torch threads there run ~20x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(lib, torch):
    """The oracle (reference graph as written, torch-CPU fp32, all host cores): 1 D step + 1 G step."""
    from oracle import nets as onets, steps as osteps, tflib_ref as oref
    cores = host_cores()
    torch.set_num_threads(cores)
    reg = oref.Registry(dtype=torch.float32)
    for n, p in lib._params.items():
        t = p.detach().cpu().clone()
        tr_ = n not in lib._non_trainable
        reg[n] = t.requires_grad_(tr_)
        if not tr_:
            reg.non_trainable.add(n)
    cfg = onets.ResnetCfg()
    B = 64
    g = torch.Generator().manual_seed(0)
    real = torch.randint(0, 256, (B, 3072), generator=g, dtype=torch.int32)
    labels = torch.randint(0, 10, (B,), generator=g, dtype=torch.int32)
    optD = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Discriminator.')], 0.0, 0.9)
    optG = osteps.TFAdam(reg, [n for n, _ in reg.trainable_with_name('Generator')], 0.0, 0.9)
    rnd = osteps.make_rnd_resnet_d(B, 128, g, dtype=torch.float32)
    rg = osteps.make_rnd_resnet_g(B, 128, g, dtype=torch.float32)
    # 1 warm-up + 2 timed steps of each kind (the first call of a shape pays for oneDNN primitive creation and the
    # allocator's first touch; a cold D step measured ~15 % slower than a warm one)
    osteps.resnet_d_step(reg, cfg, optD, real, labels, rnd, iteration=1, B=B)
    osteps.resnet_g_step(reg, cfg, optG, rg, iteration=1, B=B)
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        osteps.resnet_d_step(reg, cfg, optD, real, labels, rnd, iteration=1, B=B)
    td = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        osteps.resnet_g_step(reg, cfg, optG, rg, iteration=1, B=B)
    tg = (time.perf_counter() - t0) / reps
    t_iter = tg + 5 * td
    return {'value': round(5 * B / t_iter, 3), 'unit': 'img/s', 'cores': cores, 'kind': 'port',
            'sample': '1 warm-up + %d timed critic steps (%.2f s each) and generator steps (%.2f s each) of the oracle at full width, B=64, '
                      'torch-CPU fp32 with %d threads; iteration = G + 5*D = %.2f s (TF1 itself is not installable)'
                      % (reps, td, tg, cores, t_iter)}


if __name__ == '__main__':
    main()
