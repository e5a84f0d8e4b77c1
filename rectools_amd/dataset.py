"""Minimal host-side mirror of `rectools.dataset` — only what the transformer fit()/recommend() path touches.

The engine accepts the reference's own `rectools.dataset.Dataset` unchanged (duck typing: `.user_id_map`,
`.item_id_map`, `.interactions.df`, `.get_raw_interactions()`, `.get_user_item_matrix()`); this module exists so that
the package is usable (tests, bench, smoke) where `rectools` is not installed.  Semantics follow
rectools/dataset/identifiers.py:29-242, interactions.py:30-201 and dataset.py:108-348; no code is shared.
"""
from __future__ import annotations

import typing as tp

import numpy as np
import pandas as pd
from scipy import sparse


class Columns:
    """Fixed column names (rectools/columns.py:18-35)."""

    User = "user_id"
    Item = "item_id"
    Weight = "weight"
    Datetime = "datetime"
    Rank = "rank"
    Score = "score"
    TargetItem = "target_item_id"
    UserItem = [User, Item]
    Interactions = [User, Item, Weight, Datetime]
    Recommendations = [User, Item, Score, Rank]
    RecommendationsI2I = [TargetItem, Item, Score, Rank]


class IdMap:
    """External <-> internal id mapping; internal ids are positions in `external_ids`."""

    def __init__(self, external_ids: np.ndarray) -> None:
        self.external_ids = np.asarray(external_ids)
        if pd.Series(self.external_ids).duplicated().any():
            raise ValueError("external ids must be unique")
        self._to_internal: tp.Optional[pd.Series] = None

    @classmethod
    def from_values(cls, values: tp.Any) -> "IdMap":
        return cls(pd.unique(np.asarray(values) if not isinstance(values, pd.Series) else values.values))

    @property
    def size(self) -> int:
        return len(self.external_ids)

    @property
    def to_internal(self) -> pd.Series:
        if self._to_internal is None:
            self._to_internal = pd.Series(np.arange(self.size), index=self.external_ids)
        return self._to_internal

    def get_sorted_internal(self) -> np.ndarray:
        return np.arange(self.size)

    def get_external_sorted_by_internal(self) -> np.ndarray:
        return self.external_ids

    def convert_to_internal(self, external: tp.Any, strict: bool = True, return_missing: bool = False) -> tp.Any:
        vals = pd.Series(np.asarray(external)).map(self.to_internal)
        missing_mask = vals.isna().values
        if strict and missing_mask.any():
            raise KeyError("Some ids are missing from the mapping")
        internal = vals[~missing_mask].astype(np.int64).values
        if return_missing:
            return internal, np.asarray(external)[missing_mask]
        return internal

    def convert_to_external(self, internal: tp.Any, strict: bool = True) -> np.ndarray:
        internal = np.asarray(internal)
        if strict and ((internal < 0) | (internal >= self.size)).any():
            raise KeyError("Some internal ids are out of range")
        return self.external_ids[internal]

    def add_ids(self, values: tp.Any, raise_if_already_present: bool = False) -> "IdMap":
        new = pd.unique(np.asarray(values) if not isinstance(values, pd.Series) else values.values)
        known = pd.Series(new).isin(self.external_ids).values
        if raise_if_already_present and known.any():
            raise ValueError("Some ids are already present in the map")
        ext = self.external_ids
        fresh = new[~known]
        if len(fresh):
            # object dtype keeps heterogeneous keys ("PAD" next to integer ids) intact
            ext = np.concatenate([ext.astype(object), fresh.astype(object)]) if ext.dtype != fresh.dtype else np.concatenate([ext, fresh])
        return IdMap(ext)


class Interactions:
    """Interactions table in internal ids (columns user_id, item_id, weight, datetime [+ extras])."""

    def __init__(self, df: pd.DataFrame) -> None:
        for c in Columns.Interactions:
            if c not in df.columns:
                raise KeyError(f"Column '{c}' must be present in interactions")
        self.df = df

    @classmethod
    def from_raw(cls, interactions: pd.DataFrame, user_id_map: IdMap, item_id_map: IdMap, keep_extra_cols: bool = False) -> "Interactions":
        df = pd.DataFrame({
            Columns.User: interactions[Columns.User].map(user_id_map.to_internal).astype(np.int64).values,
            Columns.Item: interactions[Columns.Item].map(item_id_map.to_internal).astype(np.int64).values,
            Columns.Weight: interactions[Columns.Weight].astype(float).values,
            Columns.Datetime: pd.to_datetime(interactions[Columns.Datetime]).values,
        })
        if keep_extra_cols:
            for c in interactions.columns:
                if c not in Columns.Interactions:
                    df[c] = interactions[c].values
        return cls(df)

    def to_external(self, user_id_map: IdMap, item_id_map: IdMap, include_extra_cols: bool = True) -> pd.DataFrame:
        out = pd.DataFrame({
            Columns.User: user_id_map.convert_to_external(self.df[Columns.User].values),
            Columns.Item: item_id_map.convert_to_external(self.df[Columns.Item].values),
            Columns.Weight: self.df[Columns.Weight].values,
            Columns.Datetime: self.df[Columns.Datetime].values,
        })
        if include_extra_cols:
            for c in self.df.columns:
                if c not in Columns.Interactions and (include_extra_cols is True or c in include_extra_cols):
                    out[c] = self.df[c].values
        return out

    def get_user_item_matrix(self, include_weights: bool, n_users: int, n_items: int) -> sparse.csr_matrix:
        data = self.df[Columns.Weight].values.astype(np.float32) if include_weights else np.ones(len(self.df), np.float32)
        m = sparse.csr_matrix((data, (self.df[Columns.User].values, self.df[Columns.Item].values)), shape=(n_users, n_items))
        return m


class Dataset:
    """Container of id maps and interactions (rectools/dataset/dataset.py:108-348, interactions-only subset)."""

    def __init__(self, user_id_map: IdMap, item_id_map: IdMap, interactions: Interactions, user_features: tp.Any = None,
                 item_features: tp.Any = None) -> None:
        self.user_id_map, self.item_id_map, self.interactions = user_id_map, item_id_map, interactions
        self.user_features, self.item_features = user_features, item_features

    @classmethod
    def construct(cls, interactions_df: pd.DataFrame, keep_extra_cols: bool = False, **kwargs: tp.Any) -> "Dataset":
        if kwargs.get("user_features_df") is not None or kwargs.get("item_features_df") is not None:
            raise NotImplementedError("features are outside the accelerated path (SURVEY.md §2.1)")
        user_id_map = IdMap.from_values(interactions_df[Columns.User].values)
        item_id_map = IdMap.from_values(interactions_df[Columns.Item].values)
        return cls(user_id_map, item_id_map, Interactions.from_raw(interactions_df, user_id_map, item_id_map, keep_extra_cols))

    @property
    def n_hot_users(self) -> int:
        return self.user_id_map.size

    @property
    def n_hot_items(self) -> int:
        return self.item_id_map.size

    def get_hot_item_features(self) -> None:
        return None

    def get_user_item_matrix(self, include_weights: bool = True, include_warm_users: bool = False,
                             include_warm_items: bool = False, dtype: tp.Any = np.float32) -> sparse.csr_matrix:
        return self.interactions.get_user_item_matrix(include_weights, self.user_id_map.size, self.item_id_map.size).astype(dtype)

    def get_raw_interactions(self, include_weight: bool = True, include_datetime: bool = True,
                             include_extra_cols: tp.Any = True) -> pd.DataFrame:
        return self.interactions.to_external(self.user_id_map, self.item_id_map, include_extra_cols)
