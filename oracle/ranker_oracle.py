"""Oracle (numpy, CPU) for the full-catalog top-k scorer.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates `TorchRanker.rank` — reference `rectools/models/rank/rank_torch.py:77-177` — and its scorers
`_dot_score` (:207-208), `_cosine_score` (:197-205), `_euclid_score` (:194-195), in plain numpy:

  1. `item_embs = objects_factors[sorted_object_whitelist]`                       (rank_torch.py:119-120)
  2. `scores = scorer(user_embs, item_embs)` in float32                           (rank_torch.py:133-136)
  3. pairs present in `filter_pairs_csr[:, whitelist]` (value != 0) get -inf       (rank_torch.py:138-144)
  4. `topk(k=min(k, n_whitelisted), sorted, largest=higher_is_better)`             (rank_torch.py:146-152)
  5. flatten, map positions back through the whitelist, and — only when a filter
     was given — drop entries whose score is -inf                                 (rank_torch.py:157-171)

`torch.topk` leaves the order of exactly-tied scores unspecified; this oracle (and the HIP kernel) fix
it: ties are broken towards the LOWER whitelist position.  Parity inputs must be tie-free wherever the
comparison is against torch itself.
"""
from __future__ import annotations

import typing as tp

import numpy as np
from scipy import sparse

DOT, COSINE, EUCLIDEAN = "dot", "cosine", "euclidean"
EPS_COSINE = np.float32(1e-8)  # rank_torch.py:58 `epsilon_cosine_dist`


def score_matrix(users: np.ndarray, items: np.ndarray, distance: str) -> np.ndarray:
    """float32 score matrix [n_users, n_items] (rank_torch.py:194-208)."""
    u = np.asarray(users, dtype=np.float32)
    i = np.asarray(items, dtype=np.float32)
    if distance == DOT:
        return u @ i.T
    if distance == COSINE:
        un = np.maximum(np.sqrt((u * u).sum(axis=1, dtype=np.float32, keepdims=True)), EPS_COSINE)
        inn = np.maximum(np.sqrt((i * i).sum(axis=1, dtype=np.float32, keepdims=True)), EPS_COSINE)
        return (u / un) @ (i / inn).T
    if distance == EUCLIDEAN:
        # torch.cdist(p=2): ||u - i||_2
        d2 = (
            (u * u).sum(axis=1, dtype=np.float32)[:, None]
            + (i * i).sum(axis=1, dtype=np.float32)[None, :]
            - np.float32(2.0) * (u @ i.T)
        )
        return np.sqrt(np.maximum(d2, np.float32(0.0)))
    raise NotImplementedError(f"distance {distance} is not supported")


def rank(
    subjects_factors: np.ndarray,
    objects_factors: np.ndarray,
    subject_ids: tp.Sequence[int],
    k: tp.Optional[int] = None,
    filter_pairs_csr: tp.Optional[sparse.csr_matrix] = None,
    sorted_object_whitelist: tp.Optional[np.ndarray] = None,
    distance: str = DOT,
    batch_size: int = 128,
) -> tp.Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Flat (subject_ids, object_ids, scores), grouped by subject, best first (rank_torch.py:77-177)."""
    subject_ids = np.asarray(subject_ids)
    if filter_pairs_csr is not None and filter_pairs_csr.shape[0] != len(subject_ids):
        # rank_torch.py:107-109
        raise ValueError("Number of rows in `filter_pairs_csr` must be equal to `len(sublect_ids)`")
    objects = np.asarray(objects_factors, dtype=np.float32)
    if sparse.issparse(subjects_factors):
        subjects = np.asarray(subjects_factors.toarray(), dtype=np.float32)
    else:
        subjects = np.asarray(subjects_factors, dtype=np.float32)
    if sorted_object_whitelist is None:
        sorted_object_whitelist = np.arange(objects.shape[0])
    whitelist = np.asarray(sorted_object_whitelist)
    if k is None:
        k = len(whitelist)
    higher_is_better = distance != EUCLIDEAN
    user_embs = subjects[subject_ids]
    item_embs = objects[whitelist]
    kk = min(k, item_embs.shape[0])

    out_s, out_i, out_u = [], [], []
    for start in range(0, user_embs.shape[0], batch_size):
        rows = np.arange(start, min(start + batch_size, user_embs.shape[0]))
        scores = score_matrix(user_embs[rows], item_embs, distance)
        if filter_pairs_csr is not None:
            mask = np.asarray(filter_pairs_csr[rows].toarray())[:, whitelist] != 0
            scores = np.where(mask, np.float32(-np.inf), scores)
        key = -scores if higher_is_better else scores
        # stable argsort => ties resolved towards the lower whitelist position
        order = np.argsort(key, axis=1, kind="stable")[:, :kk]
        out_s.append(np.take_along_axis(scores, order, axis=1))
        out_i.append(order)
        out_u.append(rows)
    top_scores = np.concatenate(out_s, axis=0) if out_s else np.zeros((0, kk), np.float32)
    top_inds = np.concatenate(out_i, axis=0) if out_i else np.zeros((0, kk), np.int64)
    target_inds = np.concatenate(out_u, axis=0) if out_u else np.zeros((0,), np.int64)

    all_scores = top_scores.reshape(-1)
    all_target_ids = subject_ids[target_inds].repeat(top_inds.shape[1])
    all_reco_ids = whitelist[top_inds].reshape(-1)
    if filter_pairs_csr is not None:
        keep = all_scores > -np.inf
        all_scores, all_target_ids, all_reco_ids = all_scores[keep], all_target_ids[keep], all_reco_ids[keep]
    return all_target_ids, all_reco_ids, all_scores.astype(np.float32)
