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


def mnist_like_window(n, seed=3, dim=784):
    """Rounds 1-2 generator, kept for comparison: independent uniform pixels, ~19 % non-zero, inside the
    central 20x20 window -- exactly 400 of the 784 columns ever live, which real MNIST is not."""
    rs = np.random.RandomState(seed)
    side = int(round(np.sqrt(dim)))
    img = np.zeros((n, side, side), np.float32)
    lo, hi = max(side // 2 - 10, 0), min(side // 2 + 10, side)
    win = (hi - lo) * (hi - lo)
    mask = rs.rand(n, hi - lo, hi - lo) < 0.19 * dim / win
    vals = rs.randint(1, 256, size=(n, hi - lo, hi - lo))
    img[:, lo:hi, lo:hi] = (mask * vals).astype(np.float32)
    return img.reshape(n, dim)


def _pen_strokes(n, rs, side):
    """n images of a 3..5-segment pen polyline, built the way MNIST was: the glyph's bounding box scaled to
    fill a (side-8)-pixel box in its longer dimension, soft pen edge quantised to 0..255, then placed in the
    side x side frame by its centre of mass.  Vectorised (n x box^2 distance maps per segment)."""
    box, k = side - 8, 5
    pts = rs.uniform(0.0, 1.0, size=(n, k + 1, 2))
    nuse = rs.randint(3, k + 1, size=n)                       # segments drawn
    used = np.arange(k + 1)[None, :] <= nuse[:, None]
    big = np.where(used[..., None], pts, np.nan)
    lo, hi = np.nanmin(big, axis=1, keepdims=True), np.nanmax(big, axis=1, keepdims=True)
    span = np.maximum((hi - lo).max(axis=2, keepdims=True), 1e-3)
    pts = (pts - lo) / span * (box - 1)
    pts = pts + ((box - 1) - (hi - lo) / span * (box - 1)) / 2   # shorter dimension centred in the box
    th = rs.uniform(0.8, 1.5, size=(n, 1))                    # pen half-width
    yy, xx = np.mgrid[0:box, 0:box].astype(np.float64)
    P = np.stack([xx.ravel(), yy.ravel()], axis=1)
    img = np.zeros((n, box * box), np.float64)
    for s in range(k):
        a, b = pts[:, s, :][:, None, :], pts[:, s + 1, :][:, None, :]
        ab = b - a
        t = np.clip(((P[None] - a) * ab).sum(2) / np.maximum((ab * ab).sum(2), 1e-9), 0.0, 1.0)
        d = np.sqrt(((P[None] - (a + t[..., None] * ab)) ** 2).sum(2))
        v = np.clip((th - d) / 1.2 + 0.5, 0.0, 1.0)
        img = np.maximum(img, np.where((s < nuse)[:, None], v, 0.0))
    img = np.floor(img * 255.0 + 0.5 * (img > 0.02)).clip(0, 255).reshape(n, box, box)
    m = img.sum((1, 2)) + 1e-9
    cy, cx = (img * yy[None]).sum((1, 2)) / m, (img * xx[None]).sum((1, 2)) / m
    # centre-of-mass placement, damped: a random polyline is more lopsided than a digit (0.85 makes the union
    # of live pixels over 61440 images 725 -- MNIST's training set has 717 of 784 pixels ever non-zero)
    oy = np.clip(np.rint((side - box) / 2 + 0.85 * ((box - 1) / 2 - cy)).astype(int), 0, side - box)
    ox = np.clip(np.rint((side - box) / 2 + 0.85 * ((box - 1) / 2 - cx)).astype(int), 0, side - box)
    out = np.zeros((n, side, side), np.float32)
    rows = oy[:, None, None] + np.arange(box)[None, :, None]
    cols = ox[:, None, None] + np.arange(box)[None, None, :]
    out[np.arange(n)[:, None, None], rows, cols] = img
    return out.reshape(n, side * side)


def mnist_like(n, seed=3, dim=784):
    """C2/C3 style: uint8-valued floats 0..255, un-normalised, as MnistDataLoader.cpp:73-75 yields.
    Square dim: pen-stroke glyphs with MNIST's first-order statistics (28x28: 18.8 % of the pixels of an image
    non-zero, mean pixel 33.3, mean non-zero pixel 177 -- MNIST: 19.1 %, 33.3, 174 [recalled]) and its column
    occupancy: the border pixels are rarely or never inked, 661 of 784 columns are live in a 4096-row chunk,
    701 in 16384 rows, 725 in 61440 (MNIST's 60000 training images: 717 [recalled]).  Rounds 1-2 drew
    independent pixels in the central 20x20 window (400 live columns: mnist_like_window).
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
    if side * side != dim or side < 12:
        x = rs.randint(0, 256, size=(n, dim)) * (rs.rand(n, dim) < 0.19)
        return x.astype(np.float32)
    out = np.empty((n, dim), np.float32)
    for i0 in range(0, n, 4096):                               # bounded temporaries
        i1 = min(n, i0 + 4096)
        out[i0:i1] = _pen_strokes(i1 - i0, rs, side)
    return out


def float_sparse(n, seed=3, dim=784):
    """The MNIST-like pixels normalised to [0, 1] -- what a caller who divides by 255 hands over: float-valued (no
    uint8 shortcut of the search applies), same dead columns and zero quads as mnist_like."""
    return (mnist_like(n, seed=seed, dim=dim) / np.float32(255.0)).astype(np.float32)


def float_dense(n, seed=7, dim=784):
    """Signed dense float rows with structure (8 blobs, centres U(-1,1)^dim, unit sigma 0.5): no dead column, no
    all-zero quad, no uint8 shortcut -- none of the exact data-dependent shortcuts of the batch step applies."""
    return blobs(n, dim, 8, 1, seed, sigma=0.5)


def mnist_idx(directory, n, offset=0, dim=784):
    """n rows of the real MNIST training images (IDX3, big-endian header magic 0x803; file name as
    extern/mnistReader/mnist_reader.hpp:281), raw 0..255 floats as MnistDataLoader.cpp:73-75 yields them;
    dim = 794 appends the one-hot label columns (MnistDataLoader.cpp:77-82).  Returns None when the files are
    not there."""
    import os
    ip = os.path.join(directory, "train-images-idx3-ubyte")
    if not os.path.exists(ip):
        return None
    raw = np.fromfile(ip, dtype=np.uint8)
    magic, count, rows, cols = (int.from_bytes(raw[4 * k:4 * k + 4].tobytes(), "big") for k in range(4))
    if magic != 0x803 or rows * cols != 784:
        return None
    img = raw[16:16 + count * 784].reshape(count, 784)
    idx = (offset + np.arange(n)) % count
    x = img[idx].astype(np.float32)
    if dim == 794:
        lp = os.path.join(directory, "train-labels-idx1-ubyte")
        lab = np.fromfile(lp, dtype=np.uint8)[8:8 + count]
        onehot = np.zeros((n, 10), np.float32)
        onehot[np.arange(n), lab[idx]] = 1.0
        x = np.concatenate([x, onehot], axis=1)
    return x if x.shape[1] == dim else None


def column_occupancy(X):
    """(live columns, columns) of a chunk: a column is live when any row holds a non-zero there"""
    return int((X != 0).any(axis=0).sum()), int(X.shape[1])


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
