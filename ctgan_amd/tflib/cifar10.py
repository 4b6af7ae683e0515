"""tflib.cifar10 - the reference's generator-factory contract (TF/tflib/cifar10.py:8-70) on Python 3.

`load(batch_size, data_dir, n_examples)` -> (train_gen, dev_gen); calling a factory starts an epoch and
yields `(images uint8 [B,3072], labels [B])`.  Semantics kept: the training set is the FIRST `n_examples`
rows of data_batch_1..5 (:53-54), images and labels are shuffled in place with the same numpy RNG state
every epoch (:57-60), the remainder of an epoch is dropped (:62), the dev generator is the unrestricted
test batch (:69).  `prefetch_to_device` (build-only) double-buffers batches into pinned memory and onto
the GPU so the 0.79 MB/step feed stays off the critical path.
"""
import os
import pickle

import numpy as np


def unpickle(file):
    with open(file, 'rb') as fo:
        d = pickle.load(fo, encoding='latin1')
    return d['data'], d['labels']


def _generator(filenames, batch_size, data_dir, n_examples=None):
    all_data, all_labels = [], []
    for filename in filenames:
        data, labels = unpickle(os.path.join(data_dir, filename))
        all_data.append(data)
        all_labels.append(labels)
    images = np.concatenate(all_data, axis=0)
    labels = np.concatenate(all_labels, axis=0)
    if n_examples is not None:
        images = images[0:n_examples, :]
        labels = labels[0:n_examples]

    def get_epoch():
        rng_state = np.random.get_state()
        np.random.shuffle(images)
        np.random.set_state(rng_state)
        np.random.shuffle(labels)
        for i in range(len(images) // batch_size):
            yield (images[i * batch_size:(i + 1) * batch_size], labels[i * batch_size:(i + 1) * batch_size])

    return get_epoch


def cifar_generator(filenames, batch_size, data_dir):
    return _generator(filenames, batch_size, data_dir)


def cifar_generator2(filenames, batch_size, data_dir, n_examples):
    return _generator(filenames, batch_size, data_dir, n_examples)


def load(batch_size, data_dir, n_examples):
    return (
        cifar_generator2(['data_batch_1', 'data_batch_2', 'data_batch_3', 'data_batch_4', 'data_batch_5'],
                         batch_size, data_dir, n_examples),
        cifar_generator(['test_batch'], batch_size, data_dir),
    )


def inf_train_gen(train_gen):
    """`while True: for images, labels in train_gen(): yield ...` (TF/CT_gan_cifar_resnet.py:362-365)."""
    while True:
        for images, labels in train_gen():
            yield images, labels


def prefetch_to_device(gen, device, depth=2):
    """Wrap an (images uint8, labels) iterator: batches are converted to the int32 placeholder dtypes of
    the reference (:191-192), staged in pinned host buffers and copied asynchronously on a side stream,
    `depth` batches ahead of the consumer."""
    import collections

    import torch
    stream = torch.cuda.Stream(device=device) if torch.device(device).type == 'cuda' else None
    queue = collections.deque()

    def push():
        try:
            images, labels = next(gen)
        except StopIteration:
            return False
        hi = torch.from_numpy(np.ascontiguousarray(images).astype(np.int32))
        hl = torch.from_numpy(np.ascontiguousarray(labels).astype(np.int32))
        if stream is None:
            queue.append((hi.to(device), hl.to(device), None))
            return True
        hi, hl = hi.pin_memory(), hl.pin_memory()
        with torch.cuda.stream(stream):
            di = hi.to(device, non_blocking=True)
            dl = hl.to(device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        queue.append((di, dl, ev, hi, hl))
        return True

    for _ in range(depth):
        push()
    while queue:
        item = queue.popleft()
        if item[2] is not None:
            torch.cuda.current_stream().wait_event(item[2])
        yield item[0], item[1]
        push()
