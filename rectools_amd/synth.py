"""Shape-matched synthetic inputs — no MovieLens files exist offline.

`gen_interactions` draws per-user history lengths from a clipped log-normal, item ids from a Zipf-like
popularity over a random permutation, and monotone int64 timestamps (cumulative exponential gaps).
`ML_1M` / `ML_20M` give the user/item counts of the datasets BASELINE.json names.
"""
from __future__ import annotations

import typing as tp

import numpy as np

ML_1M = dict(n_users=6040, n_items=3706, mean_len=165.0, max_len=2314, min_len=20)
ML_20M = dict(n_users=138_493, n_items=26_744, mean_len=144.0, max_len=9254, min_len=20)


def gen_lengths(n_users: int, mean_len: float, min_len: int, max_len: int, rng: np.random.Generator) -> np.ndarray:
    sigma = 1.0
    mu = np.log(max(mean_len - min_len, 1.0)) - 0.5 * sigma * sigma
    lens = min_len + rng.lognormal(mu, sigma, size=n_users)
    return np.clip(lens.astype(np.int64), min_len, max_len)


def zipf_item_sampler(n_items: int, rng: np.random.Generator, alpha: float = 1.0) -> tp.Callable[[int], np.ndarray]:
    ranks = np.arange(1, n_items + 1, dtype=np.float64)
    p = ranks ** (-alpha)
    cdf = np.cumsum(p / p.sum())
    perm = rng.permutation(n_items)

    def sample(n: int) -> np.ndarray:
        return perm[np.minimum(np.searchsorted(cdf, rng.random(n)), n_items - 1)]

    return sample


def gen_interactions(
    n_users: int, n_items: int, mean_len: float = 144.0, min_len: int = 20, max_len: int = 9254, seed: int = 0,
    clip_len: tp.Optional[int] = None,
) -> tp.Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """-> (user_ids, item_ids, unix_ts) int64 flat arrays, grouped by user, time-ascending."""
    rng = np.random.default_rng(seed)
    lens = gen_lengths(n_users, mean_len, min_len, max_len, rng)
    if clip_len is not None:
        lens = np.minimum(lens, clip_len)
    total = int(lens.sum())
    users = np.repeat(np.arange(n_users, dtype=np.int64), lens)
    items = zipf_item_sampler(n_items, rng)(total).astype(np.int64)
    gaps = rng.exponential(86400.0, size=total)
    ts = np.cumsum(gaps)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    ts = ts - np.repeat(ts[starts], lens) + np.repeat(rng.integers(1_000_000_000, 1_500_000_000, n_users), lens)
    return users, items, ts.astype(np.int64)


def viewed_csr(users: np.ndarray, items: np.ndarray, n_users: int, n_items: int):
    """Binary user x item CSR of viewed pairs (the `filter_viewed=True` input of recommend())."""
    from scipy import sparse

    m = sparse.csr_matrix((np.ones(len(users), np.float32), (users, items)), shape=(n_users, n_items))
    m.sum_duplicates()
    m.data[:] = 1.0
    m.sort_indices()
    return m
