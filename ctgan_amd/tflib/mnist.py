"""tflib.mnist - generator-factory contract of TF/tflib/mnist.py:48-104 (Python 3, no download: the
reference fetches mnist.pkl.gz from a URL at :91-95, there is no network here - pass `filepath`).

`load(batch_size, test_batch_size, n_examples)` -> (train_gen, dev_gen, test_gen); batches are
`(float32 images [B,784] in [0,1], targets [B])`.  Kept: first-`n_examples` truncation followed by one
same-state shuffle at construction (:50-55), same-state shuffle per epoch (:65-68), reshape into whole
batches (so n_examples must be a multiple of batch_size, as in the reference)."""
import gzip
import pickle

import numpy


def mnist_generator(data, batch_size, n_labelled=None, limit=None):
    images, targets = data
    images, targets = numpy.array(images), numpy.array(targets)
    rng_state = numpy.random.get_state()
    numpy.random.shuffle(images)
    numpy.random.set_state(rng_state)
    numpy.random.shuffle(targets)
    if limit is not None:
        images = images.astype('float32')[:limit]
        targets = targets.astype('int32')[:limit]

    def get_epoch():
        rng_state = numpy.random.get_state()
        numpy.random.shuffle(images)
        numpy.random.set_state(rng_state)
        numpy.random.shuffle(targets)
        n = (len(images) // batch_size) * batch_size
        image_batches = images[:n].reshape(-1, batch_size, 784)
        target_batches = targets[:n].reshape(-1, batch_size)
        for i in range(len(image_batches)):
            yield (numpy.copy(image_batches[i]), numpy.copy(target_batches[i]))

    return get_epoch


def mnist_generator2(data, batch_size, n_labelled, n_examples, limit=None):
    images, targets = data
    return mnist_generator((numpy.array(images)[0:n_examples, :], numpy.array(targets)[0:n_examples]), batch_size,
                           n_labelled, limit)


def load(batch_size, test_batch_size, n_examples=60000, n_labelled=None, filepath='/tmp/mnist.pkl.gz'):
    with gzip.open(filepath, 'rb') as f:
        train_data, dev_data, test_data = pickle.load(f, encoding='latin1')
    return (
        mnist_generator2(train_data, batch_size, n_labelled, n_examples),
        mnist_generator(dev_data, test_batch_size, n_labelled),
        mnist_generator(test_data, test_batch_size, n_labelled),
    )
