"""Seeded synthetic inputs for the parity tests and the benchmark."""
import numpy as np


def blobs(n, dim, n_blobs, seed_centres, seed_samples, sigma=0.1):
    """C1/C4 style: Gaussian blobs with centres U(-1,1)^dim."""
    rc = np.random.RandomState(seed_centres)
    centres = rc.uniform(-1.0, 1.0, size=(n_blobs, dim))
    rs = np.random.RandomState(seed_samples)
    which = rs.randint(0, n_blobs, size=n)
    x = centres[which] + sigma * rs.randn(n, dim)
    return x.astype(np.float32)


def mnist_like(n, seed=3, dim=784):
    """C2/C3 style: uint8-valued floats 0..255, ~19 % non-zero pixels inside a central
    20x20 window of a 28x28 image (un-normalised, as MnistDataLoader.cpp:73-75 yields).
    dim = side^2 + 10 appends the loader's one-hot label columns."""
    rs = np.random.RandomState(seed)
    side = int(round(np.sqrt(dim)))
    lside = int(round(np.sqrt(max(dim - 10, 0))))
    if side * side != dim and lside * lside == dim - 10 and dim > 10:
        # what MnistDataLoader really hands over: the image followed by the one-hot label
        # (MnistDataLoader.cpp:73-82, depth 784 + 10)
        img = mnist_like(n, seed, dim - 10)
        onehot = np.zeros((n, 10), np.float32)
        onehot[np.arange(n), rs.randint(0, 10, size=n)] = 1.0
        return np.concatenate([img, onehot], axis=1)
    if side * side != dim:
        x = rs.randint(0, 256, size=(n, dim)) * (rs.rand(n, dim) < 0.19)
        return x.astype(np.float32)
    img = np.zeros((n, side, side), np.float32)
    lo, hi = side // 2 - 10, side // 2 + 10
    lo, hi = max(lo, 0), min(hi, side)
    win = (hi - lo) * (hi - lo)
    frac = 0.19 * dim / win
    mask = rs.rand(n, hi - lo, hi - lo) < frac
    vals = rs.randint(1, 256, size=(n, hi - lo, hi - lo))
    img[:, lo:hi, lo:hi] = (mask * vals).astype(np.float32)
    return img.reshape(n, dim)


def correlated(n, dim, seed=5):
    """C5 style: features x_j = a*x_i + b + noise (linear relations between columns)."""
    rs = np.random.RandomState(seed)
    base = rs.randn(n, 1).astype(np.float64)
    a = rs.uniform(-2, 2, size=(1, dim))
    b = rs.uniform(-1, 1, size=(1, dim))
    x = a * base + b + 0.05 * rs.randn(n, dim)
    return x.astype(np.float32)


def random_map(n_nodes, depth, seed, scale=1.0):
    """Uniform initial map in [-scale, scale) with 3 decimals, the value set of
    Som::randomInitialize (Som.cpp:988) without depending on glibc rand()."""
    rs = np.random.RandomState(seed)
    ints = rs.randint(0, int(2000 * scale), size=(n_nodes, depth))
    return ((ints.astype(np.float32) - np.float32(1000.0 * scale)) / np.float32(1000.0)).astype(np.float32)
