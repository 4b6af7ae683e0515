"""tflib.mnist - MNIST batch feed with the reference's generator-factory contract (TF/tflib/mnist.py:8-104).

Contract kept (what `CT_gan_mnist.py:221-225` observes of `lib.mnist.load`): `load(batch_size, test_batch_size, n_examples=60000,
n_labelled=None)` returns `(train_gen, dev_gen, test_gen)`; calling a factory starts one epoch and yields `(images float32 [B,784] in
[0,1], targets [B])` - with `n_labelled` a third element, the 0/1 "labelled" vector.  The training set is the FIRST `n_examples`
rows of the pickle's training split (`mnist_generator2`, :48-51: the 1000-sample setting of the paper), dev and test are whole
(`mnist_generator`, :8-9).  Order of visits (:52-55, :63-71): ONE shuffle when the factory is built, then one shuffle per epoch, each
drawn from numpy's GLOBAL generator with images, targets (and the labelled vector) permuted by the same draw - the generator state
is saved before the image shuffle and restored before the others, so one epoch advances the global generator by exactly one
shuffle of `len(set)` items; the orders compound from epoch to epoch.  Batches are `reshape(-1, batch_size, 784)` (:73-74): a set
size that is not a multiple of the batch size raises ValueError, as numpy's reshape does in the reference.

Quirk kept: with `n_labelled` the third element of every batch is a copy of the WHOLE labelled vector (:80 yields `labelled`, not
`labelled_batches[i]`).

Built differently: the set is held once and each shuffle draws ONE index permutation (`np.random.shuffle(arange(n))` consumes the
same draws as an in-place shuffle of n rows and leaves the generator in the same state - tests/test_data_and_checkpoint.py checks
both against literal in-place shuffles).  No download (:94-96 fetches over HTTP; there is no network here and a training job should
not depend on one): the file is read from `filepath` (default: the reference's `/tmp/mnist.pkl.gz`) and a missing file is an error
that says so.  The pickle is the reference's Python-2 `mnist.pkl.gz` (three `(images [n,784] float32, targets [n] int64)` splits),
read with `encoding='latin1'`.
"""
import gzip
import os
import pickle

import numpy as np

DEFAULT_PATH = '/tmp/mnist.pkl.gz'


class MnistEpochs:
    """Callable epoch factory over (images [n,784], targets [n]) - `mnist_generator` / `mnist_generator2` of the reference."""

    def __init__(self, data, batch_size, n_labelled=None, n_examples=None, limit=None):
        images, targets = data
        images, targets = np.asarray(images), np.asarray(targets)
        if n_examples is not None:                      # mnist_generator2 (:50-51): truncate BEFORE the first shuffle
            images, targets = images[0:n_examples, :], targets[0:n_examples]
        assert len(images) == len(targets)
        self.batch_size = int(batch_size)
        self.images, self.targets = images, targets
        self.labelled = None
        self._shuffle()                                 # the pre-shuffle at construction (:10-13 / :52-55)
        if limit is not None:                           # (:14-17; `load` never passes it)
            self.images = self.images.astype('float32')[:limit]
            self.targets = self.targets.astype('int32')[:limit]
        if n_labelled is not None:
            self.labelled = np.zeros(len(self.images), dtype='int32')
            self.labelled[:n_labelled] = 1

    def _shuffle(self):
        order = np.arange(len(self.images))
        np.random.shuffle(order)                        # one draw sequence from the global generator for all arrays
        self.images, self.targets = self.images[order], self.targets[order]
        if self.labelled is not None:
            self.labelled = self.labelled[order]

    def __len__(self):
        return len(self.images) // self.batch_size

    def __call__(self):
        self._shuffle()
        B = self.batch_size
        image_batches = self.images.reshape(-1, B, 784)            # ValueError unless B divides the set, as in the reference
        target_batches = self.targets.reshape(-1, B)
        labelled = self.labelled
        if labelled is not None:
            labelled.reshape(-1, B)
        for i in range(len(image_batches)):
            if labelled is not None:
                yield np.copy(image_batches[i]), np.copy(target_batches[i]), np.copy(labelled)
            else:
                yield np.copy(image_batches[i]), np.copy(target_batches[i])


def mnist_generator(data, batch_size, n_labelled, limit=None):
    return MnistEpochs(data, batch_size, n_labelled, None, limit)


def mnist_generator2(data, batch_size, n_labelled, n_examples, limit=None):
    return MnistEpochs(data, batch_size, n_labelled, n_examples, limit)


def read_splits(filepath=DEFAULT_PATH):
    if not os.path.isfile(filepath):
        raise IOError("Couldn't find MNIST dataset at %s (mnist.pkl.gz; this library does not download it)" % filepath)
    with gzip.open(filepath, 'rb') as f:
        train_data, dev_data, test_data = pickle.load(f, encoding='latin1')
    return train_data, dev_data, test_data


def load(batch_size, test_batch_size, n_examples=60000, n_labelled=None, filepath=DEFAULT_PATH):
    train_data, dev_data, test_data = read_splits(filepath)
    return (mnist_generator2(train_data, batch_size, n_labelled, n_examples),
            mnist_generator(dev_data, test_batch_size, n_labelled),
            mnist_generator(test_data, test_batch_size, n_labelled))
