"""hipGraph-captured D/G steps.

A critic step is ~700 kernel launches issued from Python autograd; at MI355X speeds the host cannot
keep up (launch-bound), so the whole (loss graph + backward + gradient packing [+ Adam]) of each step
is captured once into a hipGraph and replayed.  Replay-safety: every random draw reads its Philox
step counter from device memory, Adam reads lr / beta powers from device memory, inputs live in
static buffers.  With world_size > 1 the all-reduce and the Adam kernels stay outside the graph.
"""
import torch

from . import functional as F
from . import kernels as K
from . import gan_cifar_resnet as R
from . import tflib as lib


import os as _os
# A/B switch: the whole iteration (G step + fake batches + N_CRITIC critic steps) as ONE hipGraph when world == 1
ITERATION_GRAPH = _os.environ.get('CTGAN_ITERATION_GRAPH', '1') != '0'
# world > 1: capture the gradient all-reduce (RCCL supports stream capture) and the Adam step INSIDE the step graphs, so that the
# multi-GPU loop is the same one-graph-per-iteration replay as the single-GPU loop (no graph boundaries, no eager launches between
# the steps).  Default (round 4) when the process group's backend is RCCL ('nccl'); the eager side-stream all-reduce
# (ddp.FlatAllReduce) stays the path for gloo (its CUDA collectives stage through the host: not capturable) and is what the CPU tests
# cover.  CTGAN_AR_IN_GRAPH=0 / 1 forces either.  Covered on one GPU by a 1-rank RCCL group incl. grad_scale = 1 / world != 1
# (tests/test_gpu_graph_loop.py); bench.py falls back to the side-stream path on every rank if any rank's capture fails.
_AR_ENV = _os.environ.get('CTGAN_AR_IN_GRAPH')
AR_IN_GRAPH = None if _AR_ENV is None else (_AR_ENV != '0')


def ar_in_graph_default():
    """In-graph collectives by default exactly when they are RCCL collectives."""
    if AR_IN_GRAPH is not None:
        return AR_IN_GRAPH
    import torch.distributed as dist
    return bool(dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl')


def _capture_kw():
    """Keyword arguments of torch.cuda.graph for this process.  With a process group alive, the c10d RCCL watchdog thread polls its work
    events (hipEventQuery) at any time - under the default GLOBAL capture error mode a poll that lands inside a capture aborts the
    process ("operation not permitted when stream is capturing": seen once in four runs of the 1-rank test).  THREAD_LOCAL mode restricts
    the capture's legality checks to the capturing thread, which is what the PyTorch notes prescribe for graphs next to NCCL."""
    import torch.distributed as dist
    return {'capture_error_mode': 'thread_local'} if (dist.is_available() and dist.is_initialized()) else {}


def quiesce_collectives():
    """Call before a capture when a process group is alive.  Collectives enqueued eagerly (warm-up passes, communicator start-up) leave
    work items on the c10d watchdog's list; it retires them by polling their events every ~100 ms, and a poll that lands inside a
    capture raised a HIP error in the watchdog thread on this stack even under THREAD_LOCAL mode (2 of 3 runs once the warm-up got
    shorter).  After a device synchronize every item is complete: one poll period later the list is empty and nothing is polled
    while capturing (collectives enqueued under capture are never put on that list)."""
    import time
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and torch.cuda.is_available():
        torch.cuda.synchronize()
        time.sleep(0.35)


class GraphedTrainer:
    def __init__(self, trainer, use_graphs=True, warmup=2, ar_in_graph=None):
        self.t = trainer
        B = R.cfg.BATCH_SIZE
        dev = trainer.dev
        self.real = torch.zeros(B, R.cfg.OUTPUT_DIM, dtype=torch.int32, device=dev)
        self.labels = torch.zeros(B, dtype=torch.int32, device=dev)
        # fake batches come from one batched generator forward per iteration (Trainer.generate_fakes)
        self.batch_fakes = R.BATCH_FAKES
        # static inputs of a whole iteration: the N_CRITIC real batches and their labels in ONE allocation, so that an iteration's
        # inputs are staged by one launch (K.pack with a pointer table) instead of 2 x N_CRITIC copies
        n_real = R.cfg.N_CRITIC * B * R.cfg.OUTPUT_DIM if self.batch_fakes else 0
        self._stage = torch.zeros(n_real + B * R.cfg.N_CRITIC, dtype=torch.int32, device=dev)
        self.labels_all = self._stage[n_real:]
        self.fake = torch.zeros(B, R.cfg.OUTPUT_DIM, dtype=torch.float32, device=dev) if self.batch_fakes else None
        self.f_graph = None
        self.fake_all = None
        self.it_graph = None          # ONE graph for a whole iteration (G step + fake batches + N_CRITIC critic steps), world == 1
        self.it_out = None
        self.real_all = self._stage[:n_real].view(R.cfg.N_CRITIC, B, R.cfg.OUTPUT_DIM) if self.batch_fakes else None
        self.ar_in_graph = bool(use_graphs and trainer.allreduce is not None and (trainer.world > 1 or getattr(trainer.allreduce, 'always', False))
                                and (ar_in_graph_default() if ar_in_graph is None else ar_in_graph))
        self.adam_in_graph = trainer.world == 1 or self.ar_in_graph
        self.d_graph = self.g_graph = None
        self.d_out = self.g_out = None
        self.graph_error = None
        self.it_graph_error = None    # only the whole-iteration graph failed to capture: the per-step graphs are in use (still `graphed`)
        if use_graphs:
            try:
                self._capture(warmup)
            except Exception as e:      # fall back to eager launches; bench.py reports it
                self.graph_error = '%s: %s' % (type(e).__name__, e)
                self.d_graph = self.g_graph = self.f_graph = self.it_graph = None
                self.fake_all = self.it_out = None
                torch.cuda.synchronize()

    def _weights_moved(self, group=None):
        """A replay with Adam inside the graph changed the weights AFTER the graph's own rebuild of the derived / packed filter
        images: tell the host-side version counter, so that an eager consumer after the replay (a stand-alone step, bench.py's
        instrumented passes, sample grids) rebuilds them instead of mixing fresh weights with images one step old."""
        if self.adam_in_graph:
            lib.bump_epoch(group)

    # -- the region that is captured (everything between input copy and all-reduce / Adam)
    def _d_body(self, real=None, labels=None, fake=None):
        t = self.t
        real = self.real if real is None else real
        labels = self.labels if labels is None else labels
        fake = self.fake if fake is None else fake
        # the critic's weights change between replays (its Adam step): its derived filter layouts must be rebuilt INSIDE
        # this graph - all of them in one or two launches.  The generator's are not used here (fake batches are inputs).
        lib.bump_epoch('Discriminator' if self.batch_fakes else None)
        prep_done = F.prepare_filters_async()        # on a side stream, under the step's input staging and first (few-channel) conv
        t.rng.begin_step()
        handed = [0]
        early = None
        if t.split_flush and t.world > 1 and t.allreduce is not None and t._n_early and self.ar_in_graph:
            # the bucket's prefix (blocks 1-2) goes to its all-reduce from inside the step: on the side stream - forked into the capture
            # when the collectives are graph nodes (RCCL) - under the rest of the penalty's double backward.  Not with the collective outside
            # the graph (gloo under graph replay): there the graph ends at the packed bucket.
            def early(gpart):
                handed[0] = t.early_reduce(gpart)
        out, grads = t.d_grads(real, labels, fake=fake, early=early)
        if prep_done is not None:
            torch.cuda.current_stream().wait_event(prep_done)       # (every consumer has waited already; the fork is joined whatever the step routed)
        self._finish(t.d_opt, grads, lo=handed[0])
        return {k: out[k].detach() for k in ('cost', 'wgan', 'acgan', 'acc_real', 'acc_fake', 'ct', 'gp') if out.get(k) is not None}

    def _finish(self, opt, grads, lo=0):
        """End of a captured step body.  Adam inside the graph: bucket + update in one launch and the step end in another (single rank), or bucket,
        in-graph all-reduce, update + step end.  Adam outside (side-stream all-reduce): the graph ends at the packed bucket and at
        the Philox counter's own advance - Trainer.reduce_and_update must not advance it again, hence rng=None there."""
        t = self.t
        if self.adam_in_graph and not self.ar_in_graph:
            opt.update(grads, 1.0 / t.world, rng=t.rng)
            return
        part = opt.gather_grads(grads, lo)
        self._reduce_in_graph(part)
        if lo and hasattr(t.allreduce, 'wait'):
            t.allreduce.wait()              # join the forked collective of the bucket's prefix (Trainer.early_reduce)
        if self.adam_in_graph:
            opt.step(1.0 / t.world, rng=t.rng)
        else:
            t.rng.end_step()

    def _reduce_in_graph(self, flat):
        """The flat gradient bucket (or its remaining part) summed over the ranks ON the capturing stream: the collective becomes a node of
        the step graph."""
        if self.ar_in_graph:
            self.t.allreduce.inline(flat)

    def _it_body(self):
        """A whole iteration of the loop body (TF/CT_gan_cifar_resnet.py:393-404) with it > 0: generator step, the fake batches
        of the N_CRITIC critic steps, then the critic steps on rows of the static batch buffers."""
        B = R.cfg.BATCH_SIZE
        g_out = self._g_body()
        fakes = self._f_body()
        d_outs = [self._d_body(self.real_all[i], self.labels_all[i * B:(i + 1) * B], fakes[i]) for i in range(R.cfg.N_CRITIC)]
        return {'g': g_out, 'd': d_outs}

    def _f_body(self):
        lib.bump_epoch('Generator')      # runs after the generator update of the iteration
        return self.t.generate_fakes(self.labels_all)

    def _g_body(self):
        t = self.t
        lib.bump_epoch()
        F.prepare_filters()
        t.rng.begin_step()
        out = t.g_losses()
        with F.deferred_wgrads():
            grads = torch.autograd.grad(out['cost'], t.g_params, grad_outputs=t.cost_seed(out['cost']), allow_unused=True)
        self._finish(t.g_opt, grads)
        return {'cost': out['cost'].detach()}

    def _capture(self, warmup):
        """Warm-up + capture of the three graphs.  Everything a body changes besides its outputs is restored afterwards
        (also when capture fails): Adam slots and step counts, the Philox step counter - so a captured trainer continues
        exactly where the eager one would (checkpoint resume, graph-vs-eager parity tests).

        Each graph owns a PRIVATE memory pool.  The graphs are replayed in an order other than the capture order
        (g, f, then d x N_CRITIC) and their outputs outlive other graphs' replays - `fake_all` is read by all N_CRITIC
        critic replays.  With one shared pool a block freed while capturing graph A is handed to graph B as storage of a
        live OUTPUT, and A's next replay scribbles over it (round 1: the fake batches of critic steps 2..5 were whatever the
        first critic replay left in that memory, hence a critic cost of -6e18).  A private pool costs ~3 GB per graph of
        the 288 GB and makes an output valid until its own graph is replayed again."""
        t = self.t
        from . import kernels as K
        K.reset_capture_workspaces()
        opts = (t.d_opt, t.g_opt)
        bufs = [b for o in opts for b in (o.m, o.v, o.state)] + [t.rng.ctr]
        snap = [b.clone() for b in bufs]
        steps = [o.t for o in opts]
        for o in opts:
            o.set_lr(0.0)       # warm-up / capture passes must not move the weights
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(warmup):
                    if self.batch_fakes:
                        self._f_body()
                    self._d_body()
                    self._g_body()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            quiesce_collectives()
            self.d_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.d_graph, **_capture_kw()):
                self.d_out = self._d_body()
            self.g_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_graph, **_capture_kw()):
                self.g_out = self._g_body()
            if self.batch_fakes:
                self.f_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.f_graph, **_capture_kw()):
                    self.fake_all = self._f_body()
                if self.adam_in_graph and ITERATION_GRAPH:
                    # Seven graph launches per iteration leave ~150 us of idle GPU at each boundary (profiles/
                    # r02_steady_state_resnet_v2.txt: 1.07 ms of 20.5 ms); one graph per iteration has one boundary.
                    # Optional: if only this capture fails (a fourth private pool), the three per-step graphs stay in use.
                    try:
                        self.it_graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(self.it_graph, **_capture_kw()):
                            self.it_out = self._it_body()
                    except Exception as e:
                        self.it_graph = self.it_out = None
                        self.it_graph_error = '%s: %s' % (type(e).__name__, e)
                        torch.cuda.synchronize()
                        # an exception inside an active capture can leave the stream / pools in a bad state: prove the per-step graphs
                        # still replay (weights do not move: the learning rate is 0 here) before relying on them
                        self.d_graph.replay(); self.g_graph.replay(); self.f_graph.replay()
                        torch.cuda.synchronize()
        finally:
            torch.cuda.synchronize()
            for b, sn in zip(bufs, snap):
                b.copy_(sn)
            for o, n in zip(opts, steps):
                o.t, o._lr_last = n, None
            torch.cuda.synchronize()

    @property
    def graphed(self):
        return self.d_graph is not None

    def _stage_d_inputs(self, real_int, labels, fake):
        self.real.copy_(real_int, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)
        if self.batch_fakes:
            if fake is None:          # stand-alone critic step: draw its fake batch now (eager, no-grad)
                fake = self.t.generate_fakes(self.labels)[0]
            self.fake.copy_(fake, non_blocking=True)

    def _stage_iteration(self, batches):
        """The iteration's N_CRITIC (data, labels) batches -> the static input buffers, in one launch when they are device-resident
        dense int32 tensors (4-byte elements move as bit patterns through the gradient-bucket gather kernel)."""
        B, n_out = R.cfg.BATCH_SIZE, R.cfg.OUTPUT_DIM
        srcs = [d for d, _ in batches] + [lab for _, lab in batches]
        ok = all(x.is_cuda and x.device == self._stage.device and x.dtype == torch.int32 and x.is_contiguous() for x in srcs)
        ok = ok and all(d.numel() == B * n_out for d, _ in batches) and all(lab.numel() == B for _, lab in batches)
        if not ok or len(batches) != R.cfg.N_CRITIC:
            for i, (data, lab) in enumerate(batches):
                self.labels_all[i * B:(i + 1) * B].copy_(lab, non_blocking=True)
                self.real_all[i].copy_(data, non_blocking=True)
            return
        n = len(batches)
        offs = [i * B * n_out for i in range(n)] + [n * B * n_out + i * B for i in range(n)]
        counts = [B * n_out] * n + [B] * n
        K.pack([x.view(torch.float32) for x in srcs], offs, counts, self._stage.view(torch.float32))

    def d_step(self, real_int, labels, iteration=0, fake=None, staged=False, between=None):
        """One critic update.  world > 1: the graph ends at the packed gradient; the all-reduce runs on the side stream
        while `between()` (the NEXT step's input staging) is enqueued, then Adam (Trainer.reduce_and_update)."""
        t = self.t
        if not self.graphed:
            return t.d_step(real_int, labels, iteration=iteration, fake=fake)
        if not staged:
            self._stage_d_inputs(real_int, labels, fake)
        t.d_opt.set_lr(t.lr(iteration))
        self.d_graph.replay()
        if not self.adam_in_graph:
            t.reduce_and_update(t.d_opt, t.d_opt.grad, between, end_rng=False)
        else:
            t.d_opt.t += 1
            self._weights_moved('Discriminator')
            if between is not None:
                between()
        return self.d_out

    def g_step(self, iteration=0, between=None):
        t = self.t
        if not self.graphed:
            return t.g_step(iteration=iteration)
        t.g_opt.set_lr(t.lr(iteration))
        self.g_graph.replay()
        if not self.adam_in_graph:
            t.reduce_and_update(t.g_opt, t.g_opt.grad, between, end_rng=False)
        else:
            t.g_opt.t += 1
            self._weights_moved('Generator')
            if between is not None:
                between()
        return self.g_out

    def train_iteration(self, iteration, next_batch):
        """[G step if it>0] + N_CRITIC x (next batch, D step)   (TF/CT_gan_cifar_resnet.py:393-404).  The inputs of step
        k+1 are staged into the static buffers right after step k's graph has been enqueued - stream order keeps that
        safe - i.e. while step k's gradient all-reduce is in flight on the side stream (world > 1)."""
        if not self.batch_fakes:
            if iteration > 0:
                self.g_step(iteration)
            out = None
            for _ in range(R.cfg.N_CRITIC):
                data, labels = next_batch()
                out = self.d_step(data, labels, iteration)
            return out
        batches = [next_batch() for _ in range(R.cfg.N_CRITIC)]
        B = R.cfg.BATCH_SIZE
        if not self.graphed:
            if iteration > 0:
                self.g_step(iteration)
            fakes = self.t.generate_fakes(torch.cat([lab for _, lab in batches], 0))
            out = None
            for i, (data, labels) in enumerate(batches):
                out = self.d_step(data, labels, iteration, fake=fakes[i])
            return out

        def stage_labels():
            for i, (_, lab) in enumerate(batches):
                self.labels_all[i * B:(i + 1) * B].copy_(lab, non_blocking=True)
        if self.it_graph is not None and iteration > 0:
            t = self.t
            self._stage_iteration(batches)
            t.set_lr(t.lr(iteration))
            self.it_graph.replay()
            t.g_opt.t += 1
            t.d_opt.t += len(batches)
            self._weights_moved()
            self.g_out = self.it_out['g']
            return self.it_out['d'][-1]
        if iteration > 0:
            self.g_step(iteration, between=stage_labels)
        else:
            stage_labels()
        self.f_graph.replay()
        fakes = self.fake_all
        self._stage_d_inputs(batches[0][0], batches[0][1], fakes[0])
        out = None
        for i in range(len(batches)):
            nxt = None
            if i + 1 < len(batches):
                nxt = (lambda j: lambda: self._stage_d_inputs(batches[j][0], batches[j][1], fakes[j]))(i + 1)
            out = self.d_step(None, None, iteration, staged=True, between=nxt)
        return out


class GraphedDCGANTrainer:
    """hipGraph replay of the shared unconditional CT-WGAN step (dcgan_step.DCGANTrainer: the DCGAN scripts, the 64x64 and
    128x128 ResNets).  Same rules as GraphedTrainer: one private memory pool per graph, the Philox step counter / Adam state
    live in device memory, capture leaves no trace in them, Adam inside the graph when world == 1."""

    def __init__(self, trainer, real_shape, real_dtype, use_graphs=True, warmup=2):
        from . import dcgan_step
        self.t = trainer
        self.real = torch.zeros(real_shape, dtype=real_dtype, device=trainer.dev)
        # the fake batches of an iteration's critic steps come from ONE generator forward (a third graph, replayed once per iteration):
        # the critic graph then neither runs the generator nor rebuilds its derived / packed filters five times per iteration
        self.batch_fakes = dcgan_step.BATCH_FAKES
        self.fake = torch.zeros(real_shape[0], trainer.mod.cfg.OUTPUT_DIM, dtype=torch.float32, device=trainer.dev) if self.batch_fakes else None
        self.f_graph = None
        self.fake_all = None
        # ONE graph for a whole iteration (generator step, the fake batches, CRITIC_ITERS critic steps on rows of a static batch buffer),
        # as GraphedTrainer's: seven graph launches and their input copies per iteration leave ~20 idle gaps of 8-20 us (rocprofv3,
        # config[1]: 4 % of the iteration)
        self.it_graph = None
        self.it_out = None
        self.it_graph_error = None
        n_it = trainer.mod.cfg.CRITIC_ITERS
        self.real_all = torch.zeros((n_it,) + tuple(real_shape), dtype=real_dtype, device=trainer.dev) if self.batch_fakes else None
        self.adam_in_graph = trainer.world == 1
        self.d_graph = self.g_graph = None
        self.d_out = self.g_out = None
        self.graph_error = None
        if use_graphs:
            try:
                self._capture(warmup)
            except Exception as e:
                self.graph_error = '%s: %s' % (type(e).__name__, e)
                self.d_graph = self.g_graph = self.f_graph = self.it_graph = None
                self.fake_all = self.it_out = None
                torch.cuda.synchronize()

    def _it_body(self):
        g_out = self._body('g')
        fakes = self._f_body()
        return {'g': g_out, 'd': [self._body('d', self.real_all[i], fakes[i]) for i in range(self.t.mod.cfg.CRITIC_ITERS)]}

    def _f_body(self):
        lib.bump_epoch('Generator')       # runs after the generator update of the iteration
        F.prepare_filters()
        return self.t.generate_fakes(self.t.mod.cfg.CRITIC_ITERS)

    def _body(self, which, real=None, fake=None):
        t = self.t
        # weights changed since the last replay: derived / packed filters are rebuilt in-graph - the critic's only in the critic graph
        # when its fake batch is an input (the generator's are rebuilt by the fake-batch graph)
        lib.bump_epoch('Discriminator' if (which == 'd' and self.batch_fakes) else None)
        F.prepare_filters()
        t.rng.begin_step()
        if which == 'd':
            out, grads = t.d_grads(self.real if real is None else real, fake=self.fake if fake is None else fake)
            opt = t.d_opt
        else:
            out = t.g_losses()
            params, opt = t.g_params, t.g_opt
            with F.deferred_wgrads():       # the step's queued weight gradients in one grouped launch, as the ResNet step
                grads = torch.autograd.grad(out['cost'], params, grad_outputs=t.cost_seed().reshape(out['cost'].shape), allow_unused=True)
        if self.adam_in_graph:
            opt.update(grads, 1.0 / t.loss_scale, rng=t.rng)
        else:
            opt.gather_grads(grads)
            t.rng.end_step()
        return {k: out[k].detach() for k in ('cost', 'wgan_only', 'ct', 'gp') if out.get(k) is not None}

    def _capture(self, warmup):
        t = self.t
        from . import kernels as K
        K.reset_capture_workspaces()
        opts = (t.d_opt, t.g_opt)
        bufs = [b for o in opts for b in (o.m, o.v, o.state)] + [t.rng.ctr]
        snap = [b.clone() for b in bufs]
        steps = [o.t for o in opts]
        for o in opts:
            o.set_lr(0.0)
        try:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(warmup):
                    if self.batch_fakes:
                        self.fake.copy_(self._f_body()[0])      # (a real fake batch, not the zeros the static buffer starts with)
                    self._body('d')
                    self._body('g')
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            quiesce_collectives()
            self.d_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.d_graph, **_capture_kw()):
                self.d_out = self._body('d')
            self.g_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_graph, **_capture_kw()):
                self.g_out = self._body('g')
            if self.batch_fakes:
                self.f_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.f_graph, **_capture_kw()):
                    self.fake_all = self._f_body()
                if self.adam_in_graph and ITERATION_GRAPH:
                    try:
                        self.it_graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(self.it_graph, **_capture_kw()):
                            self.it_out = self._it_body()
                    except Exception as e:      # only this capture failed: the per-step graphs stay in use
                        self.it_graph = self.it_out = None
                        self.it_graph_error = '%s: %s' % (type(e).__name__, e)
                        torch.cuda.synchronize()
                        self.d_graph.replay(); self.g_graph.replay(); self.f_graph.replay()      # (still replayable; lr is 0 here)
                        torch.cuda.synchronize()
        finally:
            torch.cuda.synchronize()
            for b, sn in zip(bufs, snap):
                b.copy_(sn)
            for o, n in zip(opts, steps):
                o.t, o._lr_last = n, None
            torch.cuda.synchronize()

    @property
    def graphed(self):
        return self.d_graph is not None

    def _lr(self):
        m = self.t.mod
        return m.lr(self.t.iteration) if hasattr(m, 'lr') else m.cfg.LR

    def d_step(self, real_in, fake=None):
        t = self.t
        if not self.graphed:
            return t.d_step(real_in, fake=fake)
        self.real.copy_(real_in, non_blocking=True)
        if self.batch_fakes:
            if fake is None:            # a stand-alone critic step: draw its fake batch now
                fake = t.generate_fakes(1)[0]
            self.fake.copy_(fake, non_blocking=True)
        t.d_opt.set_lr(self._lr())
        self.d_graph.replay()
        if self.adam_in_graph:
            t.d_opt.t += 1
            lib.bump_epoch()           # the in-graph Adam step ran after the graph's own filter rebuild (see GraphedTrainer._weights_moved)
        else:
            self._reduce_update(t.d_opt)
        return self.d_out

    def g_step(self):
        t = self.t
        if not self.graphed:
            return t.g_step()
        t.g_opt.set_lr(self._lr())
        self.g_graph.replay()
        if self.adam_in_graph:
            t.g_opt.t += 1
            lib.bump_epoch()
        else:
            self._reduce_update(t.g_opt)
        return self.g_out

    def _reduce_update(self, opt):
        t = self.t
        if t.allreduce is not None and t.world > 1:
            t.allreduce(opt.grad)
            if hasattr(t.allreduce, 'wait'):
                t.allreduce.wait()
        opt.step(1.0 / (t.world * t.loss_scale))

    def train_iteration(self, iteration, next_batch):
        """[G step if it > 0] + CRITIC_ITERS x (batch, D step)  (TF/CT_gan_cifar.py:190-204)."""
        self.t.iteration = iteration
        n = self.t.mod.cfg.CRITIC_ITERS
        if self.it_graph is not None and iteration > 0:
            t = self.t
            for i in range(n):
                self.real_all[i].copy_(next_batch(), non_blocking=True)
            lr = self._lr()
            t.d_opt.set_lr(lr); t.g_opt.set_lr(lr)
            self.it_graph.replay()
            t.g_opt.t += 1
            t.d_opt.t += n
            lib.bump_epoch()
            self.g_out = self.it_out['g']
            return self.it_out['d'][-1]
        if iteration > 0:
            self.g_step()
        out = None
        fakes = None
        if self.batch_fakes:
            if self.graphed:
                self.f_graph.replay()
                fakes = self.fake_all
            else:
                fakes = self.t.generate_fakes(n)
        for i in range(n):
            out = self.d_step(next_batch(), fake=None if fakes is None else fakes[i])
        return out
