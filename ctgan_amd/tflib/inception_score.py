"""tflib.inception_score - the score formula of TF/tflib/inception_score.py:25-53.

The reference scores samples with the 2015 Inception graph, which it downloads at import (`:96-97`); neither the weights nor a
network exist here, so the CLASSIFIER is an argument: any callable mapping a float32 batch [n, H, W, 3] with values in [0, 255] to
class probabilities [n, n_classes].  What is restated is the statistic itself: over `splits` consecutive slices of the
predictions, exp(mean_i KL(p(y|x_i) || p(y))) with p(y) the slice mean; returned as (mean, std) over the slices - the population
standard deviation, as numpy's default.
"""
import math

import numpy as np


def score_from_probabilities(preds, splits=10):
    """preds [n, n_classes], rows summing to 1 -> (mean, std) of exp(mean KL) over `splits` consecutive slices."""
    preds = np.asarray(preds, dtype=np.float64)
    n = preds.shape[0]
    scores = []
    for k in range(splits):
        part = preds[k * n // splits:(k + 1) * n // splits]
        marginal = part.mean(axis=0, keepdims=True)
        kl = (part * (np.log(part) - np.log(marginal))).sum(axis=1).mean()
        scores.append(math.exp(kl))
    return float(np.mean(scores)), float(np.std(scores))


def get_inception_score(images, splits=10, classifier=None, batch_size=100):
    """images: list of HxWx3 arrays with values in [0, 255] (same checks as the reference, :26-30)."""
    if classifier is None:
        raise RuntimeError('get_inception_score needs classifier=callable([n,H,W,3] float32 in [0,255]) -> probabilities: the '
                           'Inception-2015 graph the reference downloads is not available offline')
    assert type(images) == list and type(images[0]) == np.ndarray and images[0].ndim == 3
    assert np.max(images[0]) > 10 and np.min(images[0]) >= 0.0
    preds = []
    for i in range(0, len(images), batch_size):
        batch = np.stack([im.astype(np.float32) for im in images[i:i + batch_size]], 0)
        preds.append(np.asarray(classifier(batch)))
    return score_from_probabilities(np.concatenate(preds, 0), splits)
