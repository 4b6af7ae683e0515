"""tflib.save_images - image-grid writer with the layout of TF/tflib/save_images.py:9-37
(scipy.misc.imsave no longer exists; PNG is written with PIL)."""
import numpy as np


def make_grid(X):
    """Returns the uint8 grid image (H*nh, W*nw[, 3]) the reference would save."""
    X = np.asarray(X)
    if isinstance(X.flatten()[0], np.floating):
        X = (255.99 * X).astype('uint8')                      # [0,1] -> [0,255]
    n_samples = X.shape[0]
    rows = int(np.sqrt(n_samples))
    while n_samples % rows != 0:
        rows -= 1
    nh, nw = rows, n_samples // rows
    if X.ndim == 2:
        side = int(np.sqrt(X.shape[1]))
        X = np.reshape(X, (X.shape[0], side, side))
    if X.ndim == 4:
        X = X.transpose(0, 2, 3, 1)                           # BCHW -> BHWC
        h, w = X[0].shape[:2]
        img = np.zeros((h * nh, w * nw, 3), dtype=np.uint8)
    elif X.ndim == 3:
        h, w = X[0].shape[:2]
        img = np.zeros((h * nh, w * nw), dtype=np.uint8)
    else:
        raise ValueError('save_images expects [B,HW], [B,H,W] or [B,C,H,W]')
    for n, x in enumerate(X):
        j, i = n // nw, n % nw
        img[j * h:j * h + h, i * w:i * w + w] = np.clip(x, 0, 255).astype(np.uint8)
    return img


def save_images(X, save_path):
    from PIL import Image
    Image.fromarray(make_grid(X)).save(save_path)
