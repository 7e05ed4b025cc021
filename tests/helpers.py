"""Shared helpers of the parity tests: seeded synthetic data + small indexes."""
import numpy as np

from vectordb_amd.builder import build_index


def clustered(n, d, n_clusters, seed, sigma=0.2, integer=False):
    """Gaussian-mixture points in the style of the reference's generator
    (/root/reference/datasets/generate_datasets.rs:58-100): centres U(-1,1)^d, sigma noise,
    cluster = i mod n_clusters."""
    rng = np.random.default_rng(seed)
    cent = rng.uniform(-1, 1, (n_clusters, d)).astype(np.float32)
    x = cent[np.arange(n) % n_clusters] + sigma * rng.standard_normal((n, d)).astype(np.float32)
    if integer:  # SIFT-shaped: non-negative integer-valued f32 in the u8 range
        x = np.clip(np.rint((x + 1.5) * 60.0), 0, 255)
    return x.astype(np.float32)


def small_index(n=5000, d=32, n_clusters=50, branching=10, leaf=30, metric="l2", seed=1, code_kind="binary",
                integer=False):
    x = clustered(n, d, n_clusters, seed, integer=integer)
    fi = build_index(x, branching, leaf, metric, 10, seed=seed, code_kind=code_kind)
    return x, fi


def queries_like(x, nq, seed, noise=0.05):
    rng = np.random.default_rng(seed)
    q = x[rng.integers(0, x.shape[0], nq)] + noise * rng.standard_normal((nq, x.shape[1])).astype(np.float32)
    return q.astype(np.float32)
