"""Helpers of the transformer path that are part of the reference's public surface (`rectools/models/nn/transformers/utils.py`)."""
from __future__ import annotations

import typing as tp

import numpy as np
import pandas as pd

from .dataset import Columns


def leave_one_out_mask(interactions: pd.DataFrame, val_users: tp.Union[tp.Sequence[tp.Any], np.ndarray, int, None] = None) -> np.ndarray:
    """Validation mask for `get_val_mask_func`: True at each user's LAST interaction by time (utils.py:23-58).

    Ties in time go to the row that comes last in the frame (rank method "first", ascending, then the maximum rank).
    `val_users`: None = every user, an int = that many users drawn with `np.random.choice` without replacement (the
    reference's RNG use), otherwise the explicit user ids."""
    groups = interactions.groupby(Columns.User)
    time_order = groups[Columns.Datetime].rank(method="first", ascending=True).astype(int)
    n_interactions = groups[Columns.Datetime].transform("size").astype(int)
    last = (n_interactions - time_order) == 0
    if isinstance(val_users, (int, np.integer)):
        val_users = np.random.choice(interactions[Columns.User].unique(), size=int(val_users), replace=False)
    elif val_users is None:
        return last.values
    return (interactions[Columns.User].isin(val_users) & last).values


def get_context(df: pd.DataFrame) -> pd.DataFrame:
    """One row per user — the user's EARLIEST row by datetime — to be passed as `context` to `recommend()` of models that
    need the time of the recommendation request (HSTU with relative time attention).  `rectools/dataset/context.py:22-51`:
    a missing weight column becomes 1.0, datetimes are parsed, the item column is dropped."""
    df = df.copy()
    if Columns.Weight not in df.columns:
        df[Columns.Weight] = 1.0
    df[Columns.Weight] = df[Columns.Weight].astype(float)
    df[Columns.Datetime] = pd.to_datetime(df[Columns.Datetime])
    context = df.loc[df.groupby(Columns.User)[Columns.Datetime].idxmin()]
    return context.drop(columns=[Columns.Item]) if Columns.Item in context else context
