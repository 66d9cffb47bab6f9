"""Seeded synthetic data shared by the parity tests (numpy PCG64; small sizes only)."""
import numpy as np


def mixture(n, d, n_comp=32, sigma=0.35, seed=1234):
    """Gaussian mixture like SURVEY.md §8d: means ~ N(0, I), rows = mean + sigma * N(0, I)."""
    rng = np.random.default_rng(seed)
    means = rng.standard_normal((n_comp, d)).astype(np.float32)
    comp = rng.integers(0, n_comp, n)
    x = means[comp] + np.float32(sigma) * rng.standard_normal((n, d)).astype(np.float32)
    return x.astype(np.float32)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)
