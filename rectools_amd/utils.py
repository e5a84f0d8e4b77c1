"""Helpers of the transformer path that belong to the reference's public surface
(`rectools/models/nn/transformers/utils.py`, `rectools/dataset/context.py`), restated on arrays."""
from __future__ import annotations

import typing as tp

import numpy as np
import pandas as pd

from .dataset import Columns


def leave_one_out_mask(interactions: pd.DataFrame, val_users: tp.Union[tp.Sequence[tp.Any], np.ndarray, int, None] = None) -> np.ndarray:
    """Validation mask for `get_val_mask_func`: True at each user's LAST interaction by time (utils.py:23-58).

    Interactions that share the latest timestamp: the one that comes last in the frame is the target (the reference ranks
    with method "first" and takes the maximum rank).  `val_users`: None = every user; an int = that many users drawn with
    `np.random.choice(..., replace=False)` over the users in order of appearance (the reference's RNG use, so a seeded
    run picks the same users); otherwise the explicit user ids."""
    n = len(interactions)
    user_col = interactions[Columns.User].values
    by_time = np.argsort(interactions[Columns.Datetime].values, kind="stable")     # ties keep their frame order
    codes, distinct = pd.factorize(user_col[by_time])
    last_seen = np.full(len(distinct), -1, dtype=np.int64)
    last_seen[codes] = np.arange(n)                  # repeated index: the last assignment stays = last row in time order
    mask = np.zeros(n, dtype=bool)
    mask[by_time[last_seen]] = True
    if val_users is None:
        return mask
    if isinstance(val_users, (int, np.integer)):
        val_users = np.random.choice(pd.unique(user_col), size=int(val_users), replace=False)
    return mask & np.isin(user_col, np.asarray(val_users))


def get_context(df: pd.DataFrame) -> pd.DataFrame:
    """One row per user — the user's EARLIEST row by datetime (the first such row on ties), users ascending — to be passed as
    `context` to `recommend()` of models that rank "as of" a request time (HSTU with relative time attention).
    `rectools/dataset/context.py:22-51`: a missing weight column becomes 1.0, the item column is not part of a context."""
    out = df.drop(columns=[Columns.Item], errors="ignore").copy()
    if Columns.Weight in out.columns:
        out[Columns.Weight] = out[Columns.Weight].astype(float)
    else:
        out[Columns.Weight] = 1.0
    out[Columns.Datetime] = pd.to_datetime(out[Columns.Datetime])
    earliest_first = out.iloc[np.argsort(out[Columns.Datetime].values, kind="stable")]
    first_rows = earliest_first[~earliest_first.duplicated(subset=Columns.User, keep="first")]
    return first_rows.iloc[np.argsort(first_rows[Columns.User].values, kind="stable")]
