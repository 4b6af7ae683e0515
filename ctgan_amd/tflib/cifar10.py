"""tflib.cifar10 - CIFAR-10 batch feed with the reference's generator-factory contract.

Contract kept (what callers of TF/tflib/cifar10.py:40-70 observe): `load(batch_size, data_dir, n_examples)` returns
`(train_gen, dev_gen)`; calling a factory starts one epoch and yields `(images uint8 [B,3072], labels [B])`.  The training
set is the first `n_examples` rows of data_batch_1..5, the dev set is the whole test batch; each epoch visits the set in a
fresh order drawn from numpy's GLOBAL generator (so `np.random.seed` reproduces a run, as with the reference) with images and
labels permuted together; a trailing partial batch is dropped.

Built differently: an `EpochFeed` object holds the arrays once and draws ONE index permutation per epoch (the reference
shuffles both arrays in place under a saved / restored generator state - the same permutation and the same generator
state afterwards, without moving 150 MB of pixels twice per epoch); batches are gathered rows.
`prefetch_to_device` (build-only) stages batches in pinned host memory and copies them on a side stream, `depth` batches
ahead, so the 0.79 MB/step feed stays off the critical path.
"""
import collections
import os
import pickle

import numpy as np

TRAIN_FILES = tuple('data_batch_%d' % i for i in range(1, 6))
TEST_FILES = ('test_batch',)


def _read_batch_file(path):
    with open(path, 'rb') as f:
        rec = pickle.load(f, encoding='latin1')
    return np.asarray(rec['data'], dtype=np.uint8), np.asarray(rec['labels'])


class EpochFeed:
    """Callable epoch factory over (images, labels)."""

    def __init__(self, images, labels, batch_size):
        assert len(images) == len(labels)
        self.images, self.labels, self.batch_size = images, labels, int(batch_size)

    @classmethod
    def from_files(cls, data_dir, names, batch_size, limit=None):
        parts = [_read_batch_file(os.path.join(data_dir, n)) for n in names]
        images = np.concatenate([p[0] for p in parts], axis=0)[:limit]
        labels = np.concatenate([p[1] for p in parts], axis=0)[:limit]
        return cls(images, labels, batch_size)

    def __len__(self):
        return len(self.images) // self.batch_size

    def __call__(self):
        order = np.arange(len(self.images))
        np.random.shuffle(order)                      # global generator, one draw sequence per epoch
        # the visiting order compounds from epoch to epoch, as repeated in-place shuffles do
        self.images, self.labels = self.images[order], self.labels[order]
        B = self.batch_size
        for k in range(len(self)):
            yield self.images[k * B:(k + 1) * B], self.labels[k * B:(k + 1) * B]


def load(batch_size, data_dir, n_examples):
    return (EpochFeed.from_files(data_dir, TRAIN_FILES, batch_size, limit=n_examples),
            EpochFeed.from_files(data_dir, TEST_FILES, batch_size))


def inf_train_gen(train_gen):
    """Endless stream of training batches: epoch after epoch (TF/CT_gan_cifar_resnet.py:362-365)."""
    while True:
        yield from train_gen()


def prefetch_to_device(gen, device, depth=2):
    """(images uint8, labels) iterator -> (int32 [B,3072], int32 [B]) device tensors (the placeholder dtypes of
    TF/CT_gan_cifar_resnet.py:191-192), `depth` batches in flight.  The uint8 -> int32 widening is written straight into a ring of
    pinned host buffers (allocated once per batch shape), the H2D copies run on a copy stream; a consumer that takes several batches
    per step (the N_CRITIC critic batches of an iteration) should ask for a depth of two steps' worth."""
    import torch
    on_gpu = torch.device(device).type == 'cuda'
    stream = torch.cuda.Stream(device=device) if on_gpu else None
    inflight = collections.deque()
    ring, slot, slot_done = {}, [0], {}

    def pinned(shape, k):
        """Slot `slot` of the ring of pinned int32 buffers for operand k of this shape (depth + 2 slots: a slot is rewritten only after
        its copy has been waited for by the consumer's stream and `depth` newer ones were issued)."""
        key = (k, tuple(shape))
        if key not in ring:
            ring[key] = [torch.empty(tuple(shape), dtype=torch.int32).pin_memory() for _ in range(depth + 2)]
        return ring[key][slot[0] % (depth + 2)]

    def stage():
        item = next(gen, None)
        if item is None:
            return
        if not on_gpu:
            host = [torch.from_numpy(np.ascontiguousarray(a).astype(np.int32)) for a in item]
            inflight.append((host[0].to(device), host[1].to(device), None, None))
            return
        prev = slot_done.get(slot[0] % (depth + 2))
        if prev is not None:
            prev.synchronize()        # the copy that last read this slot's pinned buffers has run (long ago, normally)
        host = []
        for k, a in enumerate(item):
            h = pinned(np.shape(a), k)
            np.copyto(h.numpy(), a, casting='unsafe')        # uint8 / int -> int32, no intermediate array
            host.append(h)
        slot[0] += 1
        with torch.cuda.stream(stream):
            dev = [h.to(device, non_blocking=True) for h in host]
            done = torch.cuda.Event()
            done.record(stream)
        slot_done[(slot[0] - 1) % (depth + 2)] = done
        inflight.append((dev[0], dev[1], done, host))

    for _ in range(depth):
        stage()
    while inflight:
        images, labels, done, _ = inflight.popleft()
        if done is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(done)
            # the tensors were allocated on the copy stream: tell the caching allocator that the consumer's stream uses them too,
            # or a dropped batch's block can be handed to the next H2D copy while the consumer's (queued) read is still pending
            images.record_stream(cur)
            labels.record_stream(cur)
        yield images, labels
        stage()
