"""Batch-sharded adversarial step: one process per GPU, gradients averaged over RCCL / xGMI.

The reference has no collective (its "multi-GPU" is in-graph device placement of roles,
TF/CT_gan_cifar_resnet.py:205-213); this is new design.  Every rank runs the
full D/G step on its own batch of BATCH_SIZE reals; after each backward the flat gradient bucket of
the network being updated (critic 1,055,115 floats = 4.2 MB, generator 1,218,307 = 4.9 MB) is
summed with ONE all-reduce and the 1/world average is folded into the Adam kernel (`grad_scale`).
The messages are latency-bound on xGMI (7 point-to-point links), hence one flat bucket per network
instead of per-layer buckets.  The reference averages tower costs (`/ len(DEVICES)`, :295,328), so
averaging is the faithful generalisation; generator BatchNorm statistics stay per rank, like the
reference's per-tower statistics (no SyncBN).
"""
import os
import sys

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        elif torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        # RCCL (and gloo) print banner lines to the process's C-level stdout when the first communicator is created ("Librccl path : ...",
        # "[Gloo] Rank 0 is connected ..."); the rank-0 stdout of bench.py is a one-JSON-line contract, so file descriptor 1 points at
        # stderr while the group and its first collective are set up.
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
            t = torch.zeros(1, device=torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else 'cpu')
            dist.all_reduce(t)
            if t.is_cuda:
                torch.cuda.synchronize()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
    return rank, world, local


class FlatAllReduce:
    """Sum a flat fp32 bucket across ranks.

    With `side_stream` the collective is ASYNCHRONOUS with respect to the caller's stream: `__call__` enqueues it on the
    side stream behind the producer's work (event wait) and returns; the caller's stream is made to wait for it only by
    `wait()`, which the optimizer step calls just before the Adam kernel.  Whatever the caller enqueues in between - the
    next step's input staging (engine.GraphedTrainer) - overlaps the collective.  Without a side stream (gloo / CPU tests)
    the call is synchronous and `wait()` is a no-op."""

    def __init__(self, group=None, side_stream=None, always=False):
        self.group = group
        self.side = side_stream
        self.always = always          # tests: issue the collective for a 1-rank group too (exercises the RCCL enqueue / capture path)
        self._pending = []            # completion events of the collectives in flight (a step may hand its bucket over in parts)

    def inline(self, flat):
        """The collective on the CALLER's current stream, synchronously in stream order - the form that can be captured into a
        hipGraph (engine.GraphedTrainer with CTGAN_AR_IN_GRAPH=1)."""
        if dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.always):
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)

    def __call__(self, flat):
        if not dist.is_initialized() or (dist.get_world_size(self.group) == 1 and not self.always):
            return
        if self.side is None or not flat.is_cuda:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            return
        assert len(self._pending) < 4, 'earlier all-reduces were never waited for'
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            done = torch.cuda.Event()
            done.record(self.side)
        self._pending.append(done)

    def wait(self):
        """Order the caller's current stream behind the outstanding collectives (if any)."""
        for done in self._pending:
            torch.cuda.current_stream().wait_event(done)
        self._pending = []


def broadcast_params(flat_buffers, src=0, group=None):
    """Make every rank start from rank `src`'s weights (the reference has a single copy)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        for b in flat_buffers:
            dist.broadcast(b, src=src, group=group)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
