"""
Synthetic speckle stacks for tests, fixtures and bench.py.

This is this repository's own generator (specified in SURVEY.md section 8(d)); it is
not derived from the reference's wave-optics simulator.  A reference stack is
Gaussian-filtered white noise; the sample stack is the reference warped by a
smooth displacement field, attenuated (transmission T0), contrast-reduced
(dark-field D0) and with a little additive noise.

Sign convention (checked against the reference):
``sam[i, j] = ref[i + u_row, j + u_col]`` gives positive ``dy ~ u_row`` and
``dx ~ u_col`` in the result maps.
"""
import numpy as np

__all__ = ["displacement_field", "make_stack", "CONFIGS"]

# BASELINE.json configs (H, W, K, Nw, max_shift, dark-field)
CONFIGS = {
    "C1": dict(H=256, W=256, K=3, Nw=3, max_shift=2, df=False),
    "C2": dict(H=2048, W=2048, K=10, Nw=5, max_shift=5, df=True),
    "C3": dict(H=4096, W=4096, K=20, Nw=7, max_shift=8, df=True),
    "C4": dict(H=8192, W=8192, K=10, Nw=5, max_shift=5, df=True),
    "C5": dict(H=2048, W=2048, K=5, Nw=5, max_shift=5, df=True),
}


def displacement_field(H, W, A, rows=None, H_full=None):
    """Smooth displacement field (u_row, u_col) of amplitude ``A`` pixels.

    ``rows`` (optional) is an array of absolute row indices into a virtual image of
    height ``H_full``; it lets one rank generate only its own row slab of a larger
    image with the same field as the whole-image generator.
    """
    if rows is None:
        rows = np.arange(H)
    if H_full is None:
        H_full = H
    y = np.asarray(rows, dtype=np.float64)[:, None]
    x = np.arange(W, dtype=np.float64)[None, :]
    u_row = A * np.sin(2 * np.pi * y / H_full) * np.cos(np.pi * x / W)
    u_col = A * np.cos(2 * np.pi * x / W) * np.ones_like(y)
    return u_row, u_col


def _speckle(H, W, seed, sigma=1.5):
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    g = gaussian_filter(rng.standard_normal((H, W)), sigma)
    return 1.0 + 0.3 * g / g.std()


def make_stack(H, W, K, max_shift, df=True, seed=0, T0=0.8, D0=0.7, noise=0.005,
               amplitude=None, order=3):
    """Return ``(sam, ref, (u_row, u_col))``: float64 C-contiguous ``[K, H, W]`` stacks.

    ref[k] = 1 + 0.3 g/std(g),  g = gaussian_filter(N(0,1), 1.5 px), rng 1000+k+seed
    sam[k] = T0 (D0 (warp(ref[k]) - 1) + 1) + N(0, noise), rng 2000+k+seed
    """
    from scipy.ndimage import map_coordinates
    if amplitude is None:
        amplitude = max_shift - 2.5
    if not df:
        D0 = 1.0
    u_row, u_col = displacement_field(H, W, amplitude)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64),
                         np.arange(W, dtype=np.float64), indexing="ij")
    coords = np.stack([yy + u_row, xx + u_col])
    ref = np.empty((K, H, W), dtype=np.float64)
    sam = np.empty((K, H, W), dtype=np.float64)
    for k in range(K):
        ref[k] = _speckle(H, W, 1000 + k + seed)
        warped = map_coordinates(ref[k], coords, order=order, mode="reflect")
        rng = np.random.default_rng(2000 + k + seed)
        sam[k] = T0 * (D0 * (warped - 1.0) + 1.0) + noise * rng.standard_normal((H, W))
    return sam, ref, (u_row, u_col)


def make_stack_fast(H, W, K, max_shift, df=True, seed=0, T0=0.8, D0=0.7, noise=0.005,
                    amplitude=None):
    """Same statistics as :func:`make_stack` with bilinear warping (order=1).

    Used by bench.py for the large configurations where cubic interpolation of
    K x 4..67 Mpx on the host would dominate the wall time of the run.
    """
    return make_stack(H, W, K, max_shift, df=df, seed=seed, T0=T0, D0=D0, noise=noise,
                      amplitude=amplitude, order=1)


def make_block(rows, H_full, W, K, max_shift, df=True, seed=0, T0=0.8, D0=0.7, noise=0.005,
               amplitude=None, order=1):
    """Rows ``rows = (a, b)`` of a virtual ``H_full x W`` stack, for a rank that owns only that row block
    (bench.py's multi-GPU leg: BASELINE config C4 is 8 such blocks of 1024 rows).

    The displacement field is the whole image's; the speckle of a block is drawn from the block's own seed
    (with a margin so that the warp has rows to sample from), so the image is the concatenation of the blocks.
    Returns ``(sam, ref)``, float64 ``[K, b - a, W]``.
    """
    from scipy.ndimage import map_coordinates
    a, b = int(rows[0]), int(rows[1])
    if amplitude is None:
        amplitude = max_shift - 2.5
    if not df:
        D0 = 1.0
    m = int(np.ceil(abs(amplitude))) + 2                 # rows of speckle beyond the block on either side
    n = b - a
    u_row, u_col = displacement_field(n, W, amplitude, rows=np.arange(a, b), H_full=H_full)
    yy, xx = np.meshgrid(np.arange(n, dtype=np.float64) + m, np.arange(W, dtype=np.float64), indexing="ij")
    coords = np.stack([yy + u_row, xx + u_col])
    ref = np.empty((K, n, W), dtype=np.float64)
    sam = np.empty((K, n, W), dtype=np.float64)
    for k in range(K):
        big = _speckle(n + 2 * m, W, 1000 + k + seed)
        ref[k] = big[m:m + n]
        warped = map_coordinates(big, coords, order=order, mode="reflect")
        rng = np.random.default_rng(2000 + k + seed)
        sam[k] = T0 * (D0 * (warped - 1.0) + 1.0) + noise * rng.standard_normal((n, W))
    return sam, ref


def valley_stack(n, D, q, K=1):
    """A stack whose cost landscape is a long narrow valley along the shift diagonal: the walk zigzags down it for
    hundreds of evaluations (tests of the 500-call cap, Optim.cpp:14,267).  Every value is an exactly reproducible
    IEEE expression of small integers, so the fixture stores the parameters instead of the arrays."""
    m = n + D
    y = np.arange(m, dtype=np.float64)[:, None]
    x = np.arange(m, dtype=np.float64)[None, :]
    sam, ref = [], []
    for k in range(K):
        big = 1.0 + 0.5 * ((x - y + k) / q) ** 2 + ((x + y) / 4096.0) ** 2
        ref.append(np.ascontiguousarray(big[:n, :n]))
        sam.append(np.ascontiguousarray(0.75 * big[D:D + n, D:D + n]))
    return np.stack(sam), np.stack(ref)
