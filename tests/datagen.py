"""Deterministic synthetic inputs shared by tests, golden generation and bench (no numpy RNG state involved:
a splitmix64 counter hash -> uniforms -> Box-Muller, so any row range can be regenerated anywhere)."""
import hashlib
import itertools

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniforms(seed, n):
    """n float64 uniforms in (0,1), element i a pure function of (seed, i)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        h = _splitmix64(idx ^ _splitmix64(np.uint64(seed)))
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)


def normals(seed, shape):
    n = int(np.prod(shape))
    m = (n + 1) // 2
    u1 = uniforms(seed * 2 + 1, m)
    u2 = uniforms(seed * 2 + 2, m)
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])[:n]
    return z.reshape(shape)


def mixture(n, dim, seed, n_clusters=None, spread=0.3, normalize=False, intrinsic_dim=None, basis_seed=None,
            centre_scale=1.0):
    """Gaussian mixture: sqrt(n) centres ~ N(0, I), points = centre + spread * N(0, I).
    With `intrinsic_dim` the noise lives in a random intrinsic_dim-dimensional subspace (low intrinsic dimension, so
    that graph indexes reach high recall); `basis_seed` pins the subspace so data and queries share it."""
    k = n_clusters or max(2, int(np.sqrt(n)))
    centres = centre_scale * normals((seed if basis_seed is None else basis_seed) * 7 + 1, (k, dim))
    assign = (uniforms(seed * 7 + 2, n) * k).astype(np.int64)
    if intrinsic_dim:
        basis = normals((seed if basis_seed is None else basis_seed) * 7 + 4, (intrinsic_dim, dim)) / np.sqrt(dim)
        x = centres[assign] + spread * (normals(seed * 7 + 3, (n, intrinsic_dim)) @ basis)
    else:
        x = centres[assign] + spread * normals(seed * 7 + 3, (n, dim))
    x = x.astype(np.float32)
    if normalize:
        x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x)


def readme_grid():
    """The README / hnsw_result.test dataset: all [a,b,c] with a,b,c in 1..9 (reference README.md:12-14)."""
    return np.array(list(itertools.product(range(1, 10), repeat=3)), dtype=np.float32)


def sha(a):
    if isinstance(a, np.ndarray):
        a = np.ascontiguousarray(a).tobytes()
    return hashlib.sha256(a).hexdigest()
