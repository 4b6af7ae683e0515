"""the reference's data-generator contract and checkpoint/sample-dump plumbing (CPU)."""
import gzip
import os
import pickle

import numpy as np
import pytest
import torch


def _fake_cifar(d, n_per=40):
    rng = np.random.default_rng(0)
    for i, name in enumerate(['data_batch_%d' % k for k in range(1, 6)] + ['test_batch']):
        data = rng.integers(0, 256, (n_per, 3072), dtype=np.uint8)
        data[:, 0] = np.arange(i * n_per, (i + 1) * n_per) % 256          # row id in column 0
        labels = [int(x) for x in (np.arange(i * n_per, (i + 1) * n_per) % 10)]
        with open(os.path.join(d, name), 'wb') as f:
            pickle.dump({'data': data, 'labels': labels}, f, protocol=2)


def test_cifar10_generator_contract(tmp_path):
    from ctgan_amd.tflib import cifar10
    _fake_cifar(str(tmp_path))
    train_gen, dev_gen = cifar10.load(16, str(tmp_path), n_examples=100)
    np.random.seed(3)
    batches = list(train_gen())
    assert len(batches) == 100 // 16                                   # remainder dropped (TF/tflib/cifar10.py:62)
    ids = np.concatenate([b[0][:, 0] for b in batches]); labs = np.concatenate([b[1] for b in batches])
    assert batches[0][0].dtype == np.uint8 and batches[0][0].shape == (16, 3072)
    assert set(ids.tolist()) <= set(range(100))                          # FIRST n_examples only (:53-54)
    assert np.array_equal(ids % 10, labs)                                # images and labels shuffled identically (:57-60)
    ids2 = np.concatenate([b[0][:, 0] for b in train_gen()])
    assert not np.array_equal(ids, ids2)                                 # reshuffled every epoch
    assert len(list(dev_gen())) == 40 // 16                              # test batch, unrestricted
    it = cifar10.inf_train_gen(train_gen)
    assert len([next(it) for _ in range(20)]) == 20                      # endless
    dev = list(cifar10.prefetch_to_device(iter(batches), 'cpu'))
    assert len(dev) == len(batches) and dev[0][0].dtype == torch.int32 and dev[0][1].dtype == torch.int32
    assert torch.equal(dev[2][0], torch.from_numpy(batches[2][0].astype(np.int32)))


def test_cifar10_epoch_order_is_the_in_place_shuffle_order(tmp_path):
    """The reference shuffles images and labels in place under a saved / restored global generator state
    (TF/tflib/cifar10.py:57-60); EpochFeed draws one index permutation per epoch.  Same rows in the same order, epoch after
    epoch, and the same generator state afterwards."""
    from ctgan_amd.tflib import cifar10
    _fake_cifar(str(tmp_path))
    train_gen, _ = cifar10.load(10, str(tmp_path), n_examples=100)
    ids0 = train_gen.images[:, 0].copy()
    np.random.seed(11)
    got = [np.concatenate([b[0][:, 0] for b in train_gen()]) for _ in range(3)]
    state_after = np.random.get_state()[1].copy()
    np.random.seed(11)
    ref = ids0.copy()
    for e in range(3):
        st = np.random.get_state(); np.random.shuffle(ref); np.random.set_state(st)
        np.random.shuffle(np.arange(100))           # the labels' shuffle: advances the generator exactly as much
        assert np.array_equal(got[e], ref)
    assert np.array_equal(state_after, np.random.get_state()[1])


def _fake_mnist(path, n_train=120, n_dev=40, n_test=40):
    rng = np.random.default_rng(1)

    def split(n, base):
        x = rng.random((n, 784), dtype=np.float32)
        x[:, 0] = np.arange(base, base + n)                                # row id in column 0
        return x, (np.arange(base, base + n) % 10).astype(np.int64)
    with gzip.open(path, 'wb') as f:
        pickle.dump((split(n_train, 0), split(n_dev, 1000), split(n_test, 2000)), f, protocol=2)


def test_mnist_generator_contract(tmp_path):
    """TF/tflib/mnist.py:48-104: first-n_examples truncation, (images [B,784] float32, targets [B]) batches, images and targets
    permuted together, a fresh order every epoch, whole dev / test splits, reshape error when B does not divide the set, no download."""
    from ctgan_amd.tflib import mnist
    path = str(tmp_path / 'mnist.pkl.gz')
    with pytest.raises(IOError, match="Couldn't find MNIST dataset"):
        mnist.load(10, 10, filepath=path)
    _fake_mnist(path)
    np.random.seed(5)
    train_gen, dev_gen, test_gen = mnist.load(10, 20, n_examples=50, filepath=path)
    batches = list(train_gen())
    assert len(batches) == 5 and batches[0][0].shape == (10, 784) and batches[0][0].dtype == np.float32 and batches[0][1].shape == (10,)
    ids = np.concatenate([b[0][:, 0] for b in batches]).astype(int); labs = np.concatenate([b[1] for b in batches])
    assert sorted(ids.tolist()) == list(range(50))                         # FIRST n_examples only (:50-51), each once per epoch
    assert np.array_equal(ids % 10, labs)                                  # same permutation for images and targets (:52-55, :64-67)
    ids2 = np.concatenate([b[0][:, 0] for b in train_gen()]).astype(int)
    assert sorted(ids2.tolist()) == list(range(50)) and not np.array_equal(ids, ids2)
    assert sorted(np.concatenate([b[0][:, 0] for b in dev_gen()]).astype(int).tolist()) == list(range(1000, 1040))
    assert sorted(np.concatenate([b[0][:, 0] for b in test_gen()]).astype(int).tolist()) == list(range(2000, 2040))
    b0 = next(iter(train_gen())); b0[0][:] = -1                            # batches are copies (:85): the set is not written through them
    assert train_gen.images.min() >= 0
    bad, _, _ = mnist.load(16, 20, n_examples=50, filepath=path)           # 50 rows, batches of 16: numpy's reshape error (:73)
    with pytest.raises(ValueError):
        next(iter(bad()))
    lab_gen, _, _ = mnist.load(10, 20, n_examples=50, n_labelled=7, filepath=path)
    x, t, lab = next(iter(lab_gen()))
    assert lab.shape == (50,) and lab.sum() == 7 and lab.dtype == np.int32  # the whole labelled vector rides every batch (:80)


def test_mnist_order_is_the_reference_in_place_shuffle_order(tmp_path):
    """One shuffle when the factory is built, one per epoch, each under a saved / restored global generator state
    (TF/tflib/mnist.py:52-55, :64-71): the rows MnistEpochs yields against literal in-place shuffles of the same arrays, and the
    global generator's state afterwards."""
    from ctgan_amd.tflib import mnist
    path = str(tmp_path / 'mnist.pkl.gz')
    _fake_mnist(path)
    train, _, _ = mnist.read_splits(path)
    np.random.seed(21)
    gen = mnist.mnist_generator2(train, 10, None, 60)
    got = [np.concatenate([b[0][:, 0] for b in gen()]) for _ in range(3)]
    got_t = np.concatenate([b[1] for b in gen()])
    state_after = np.random.get_state()[1].copy()
    # literal restatement on copies
    np.random.seed(21)
    images, targets = train[0][0:60, :].copy(), train[1][0:60].copy()

    def both():
        st = np.random.get_state(); np.random.shuffle(images); np.random.set_state(st); np.random.shuffle(targets)
    both()
    for e in range(3):
        both()
        assert np.array_equal(got[e], images[:, 0])
    both()
    assert np.array_equal(got_t, targets)
    assert np.array_equal(state_after, np.random.get_state()[1])


def test_save_images_grid(tmp_path):
    from ctgan_amd.tflib import save_images
    X = np.zeros((6, 3, 4, 4), dtype=np.int32); X[3] = 200
    g = save_images.make_grid(X)
    assert g.shape == (2 * 4, 3 * 4, 3) and g[4:8, 0:4].min() == 200 and g[0:4].max() == 0   # sample 3 -> row 1, col 0
    assert save_images.make_grid(np.random.rand(128, 784).astype('float32')).shape == (8 * 28, 16 * 28)
    save_images.save_images(X, str(tmp_path / 's.png'))
    assert os.path.getsize(str(tmp_path / 's.png')) > 0


def test_checkpoint_roundtrip(cpu_kernels, tmp_path):
    import ctgan_amd.gan_cifar_resnet as R
    import ctgan_amd.tflib as lib
    from ctgan_amd import checkpoint
    lib.set_seed(2)
    R.configure(DIM_G=8, DIM_D=8, BATCH_SIZE=4)
    try:
        R.build_params('cpu')
        tr = R.Trainer(seed=5)
        g = torch.Generator().manual_seed(0)
        real = torch.randint(0, 256, (4, 3072), generator=g, dtype=torch.int32)
        labels = torch.randint(0, 10, (4,), generator=g, dtype=torch.int32)
        tr.train_iteration(1, lambda: (real, labels))
        path = str(tmp_path / 'ck.pt')
        checkpoint.save(path, tr, iteration=2)
        want = {n: p.detach().clone() for n, p in lib._params.items()}
        wm, wv, wt_, wst = tr.d_opt.m.clone(), tr.g_opt.v.clone(), tr.d_opt.t, tr.d_opt.state.clone()
        tr.train_iteration(2, lambda: (real, labels))                    # move away from the saved state
        assert not torch.equal(tr.d_opt.m, wm)
        it = checkpoint.load(path, tr)
        assert it == 2 and tr.d_opt.t == wt_ and torch.equal(tr.d_opt.m, wm) and torch.equal(tr.g_opt.v, wv)
        assert torch.equal(tr.d_opt.state, wst)
        for n, p in lib._params.items():
            assert torch.equal(p.detach(), want[n]), n
        assert tr.d_params[0].data_ptr() == tr.d_opt.theta.data_ptr()     # still views into the flat buffer
    finally:
        R.configure()


def test_inception_score_statistic():
    """tflib.inception_score: the score formula on known distributions - uniform predictions score 1, one-hot
    predictions spread evenly over k classes score k; the classifier is an argument (no Inception weights offline)."""
    from ctgan_amd.tflib import inception_score as isc
    rng = np.random.default_rng(0)
    n, k = 1000, 10
    m, s = isc.score_from_probabilities(np.full((n, k), 1.0 / k))
    assert abs(m - 1.0) < 1e-12 and s < 1e-12
    onehot = np.full((n, k), 1e-12); onehot[np.arange(n), np.arange(n) % k] = 1.0
    onehot /= onehot.sum(1, keepdims=True)
    m, s = isc.score_from_probabilities(onehot)
    assert abs(m - k) < 1e-6
    p = rng.dirichlet(np.ones(k), size=n)
    ref = []
    for i in range(10):                      # the statistic written out independently
        part = p[i * n // 10:(i + 1) * n // 10]
        py = part.mean(0)
        ref.append(np.exp(np.mean([(row * np.log(row / py)).sum() for row in part])))
    m, s = isc.score_from_probabilities(p, 10)
    assert abs(m - np.mean(ref)) < 1e-12 and abs(s - np.std(ref)) < 1e-12
    imgs = [rng.integers(0, 256, (32, 32, 3)).astype(np.float64) for _ in range(250)]
    m2, _ = isc.get_inception_score(imgs, splits=5, classifier=lambda b: np.full((b.shape[0], k), 1.0 / k))
    assert abs(m2 - 1.0) < 1e-12
    with pytest.raises(RuntimeError):
        isc.get_inception_score(imgs)
