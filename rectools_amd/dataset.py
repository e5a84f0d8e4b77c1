"""Minimal host-side mirror of `rectools.dataset` — only what the transformer fit()/recommend() path touches
(interactions, id maps, and sparse item features for `CatFeaturesItemNet`).

The engine accepts the reference's own `rectools.dataset.Dataset` unchanged (duck typing: `.user_id_map`,
`.item_id_map`, `.interactions.df`, `.get_raw_interactions()`, `.get_user_item_matrix()`); this module exists so that
the package is usable (tests, bench, smoke) where `rectools` is not installed.  Semantics follow
rectools/dataset/identifiers.py:29-242, interactions.py:30-201, features.py:172-468 and dataset.py:108-348; no code
is shared.
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
    def external_dtype(self) -> tp.Any:
        return self.external_ids.dtype

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


DIRECT_FEATURE_VALUE = "__is_direct_feature"   # second element of a direct (non-categorical) feature's name


class SparseFeatures:
    """Sparse feature matrix [n_objects, n_columns] + column names `(feature, value)` (features.py:172-468).

    Direct features keep their numeric value in one column named `(feature, DIRECT_FEATURE_VALUE)`; a categorical
    feature becomes one column per distinct value (one-hot, duplicates counted, times weight)."""

    def __init__(self, values: sparse.csr_matrix, names: tp.Sequence[tp.Tuple[tp.Any, tp.Any]]) -> None:
        names = tuple(names)
        if values.shape[1] != len(names):
            raise ValueError("Number of feature names must be equal to the number of columns of the values matrix")
        self.values, self.names = values, names

    @classmethod
    def from_iterables(cls, values: sparse.csr_matrix, names: tp.Iterable[tp.Tuple[tp.Any, tp.Any]]) -> "SparseFeatures":
        return cls(sparse.csr_matrix(values).astype(np.float32), tuple(names))

    @classmethod
    def from_flatten(cls, df: pd.DataFrame, id_map: IdMap, cat_features: tp.Iterable[tp.Any] = (), id_col: str = "id",
                     feature_col: str = "feature", value_col: str = "value", weight_col: str = "weight") -> "SparseFeatures":
        """features.py:254-376.  Column order: direct features (first-appearance order), then for every categorical
        feature in `cat_features` order its values in first-appearance order."""
        missing = {id_col, feature_col, value_col} - set(df.columns)
        if missing:
            raise KeyError(f"Missed columns {missing}")
        try:
            ids = id_map.convert_to_internal(df[id_col].values)
        except KeyError as e:
            raise KeyError("All ids in `df` must be present in `id_map`") from e
        try:
            weights = df[weight_col].values.astype(float) if weight_col in df else np.ones(len(df))
        except ValueError as e:
            raise TypeError("Weights must be numeric") from e
        feats, vals = df[feature_col].values, df[value_col].values
        cat_features = list(cat_features)
        n = id_map.size
        is_cat = pd.Series(feats).isin(cat_features).values
        blocks, names = [], []
        direct = ~is_cat
        direct_names = pd.unique(feats[direct])
        col = pd.Series(np.arange(len(direct_names)), index=direct_names)
        try:
            dvals = vals[direct].astype(np.float32) if direct.any() else np.zeros(0, np.float32)
        except ValueError as e:
            raise TypeError("Values of direct features must be numeric") from e
        blocks.append(sparse.csr_matrix((dvals * weights[direct], (ids[direct], col.reindex(feats[direct]).values.astype(np.int64)
                                                                if direct.any() else np.zeros(0, np.int64))),
                                        shape=(n, len(direct_names))))
        names.extend((f, DIRECT_FEATURE_VALUE) for f in direct_names)
        for cf in cat_features:
            sel = feats == cf
            uniq = pd.unique(vals[sel])
            vcol = pd.Series(np.arange(len(uniq)), index=uniq)
            cols = vcol.reindex(vals[sel]).values.astype(np.int64) if sel.any() else np.zeros(0, np.int64)
            blocks.append(sparse.csr_matrix((weights[sel], (ids[sel], cols)), shape=(n, len(uniq))))
            names.extend((cf, v) for v in uniq)
        csr = sparse.hstack(blocks, format="csr")
        csr.sum_duplicates()
        return cls.from_iterables(csr, names)

    def get_sparse(self) -> sparse.csr_matrix:
        return self.values

    def take(self, ids: tp.Any) -> "SparseFeatures":
        return SparseFeatures(self.values[np.asarray(ids)], self.names)

    def __len__(self) -> int:
        return self.values.shape[0]

    @property
    def cat_col_mask(self) -> np.ndarray:
        return np.array([name[1] != DIRECT_FEATURE_VALUE for name in self.names], dtype=bool)

    @property
    def cat_feature_indices(self) -> np.ndarray:
        return np.arange(len(self.names))[self.cat_col_mask]

    def get_cat_features(self) -> "SparseFeatures":
        idx = self.cat_feature_indices
        return SparseFeatures(self.values[:, idx], tuple(self.names[i] for i in idx))


class Dataset:
    """Container of id maps, interactions and (optionally) sparse item features (rectools/dataset/dataset.py:108-348)."""

    def __init__(self, user_id_map: IdMap, item_id_map: IdMap, interactions: Interactions, user_features: tp.Any = None,
                 item_features: tp.Any = None) -> None:
        self.user_id_map, self.item_id_map, self.interactions = user_id_map, item_id_map, interactions
        self.user_features, self.item_features = user_features, item_features

    @classmethod
    def construct(cls, interactions_df: pd.DataFrame, keep_extra_cols: bool = False, item_features_df: tp.Optional[pd.DataFrame] = None,
                  cat_item_features: tp.Iterable[tp.Any] = (), make_dense_item_features: bool = False,
                  **kwargs: tp.Any) -> "Dataset":
        """dataset.py:208-281.  Items that appear only in `item_features_df` are appended to the item id map ("warm"
        items, dataset.py:295).  User features are not used by the transformer models (data_preparator.py:261) and dense
        item features have no categorical columns (item_net.py:138-143): both are rejected here."""
        if kwargs.get("user_features_df") is not None or make_dense_item_features:
            raise NotImplementedError("user features / dense item features are outside the accelerated path (SURVEY.md §2.1)")
        # one factorize per id column gives the id map (distinct values in order of appearance, what IdMap.from_values
        # keeps) AND the internal ids, instead of a unique pass followed by a hash-map lookup per row
        u_codes, u_uniques = pd.factorize(interactions_df[Columns.User].values)
        i_codes, i_uniques = pd.factorize(interactions_df[Columns.Item].values)
        if (u_codes < 0).any() or (i_codes < 0).any():
            raise ValueError("user / item ids must not be missing values")
        user_id_map, item_id_map = IdMap(u_uniques), IdMap(i_uniques)
        frame = {Columns.User: u_codes.astype(np.int64, copy=False), Columns.Item: i_codes.astype(np.int64, copy=False),
                 Columns.Weight: interactions_df[Columns.Weight].astype(float).values,
                 Columns.Datetime: pd.to_datetime(interactions_df[Columns.Datetime]).values}
        if keep_extra_cols:
            for c in interactions_df.columns:
                if c not in Columns.Interactions:
                    frame[c] = interactions_df[c].values
        interactions = Interactions(pd.DataFrame(frame, copy=False))
        item_features = None
        if item_features_df is not None:
            id_col = Columns.Item if Columns.Item in item_features_df else "id"
            item_id_map = item_id_map.add_ids(item_features_df[id_col].values)
            item_features = SparseFeatures.from_flatten(item_features_df, item_id_map, cat_item_features, id_col=id_col)
        return cls(user_id_map, item_id_map, interactions, item_features=item_features)

    def get_schema(self) -> tp.Dict[str, tp.Any]:
        """Dataset statistics in the reference's JSON-able layout (dataset.py:139-174): what a checkpoint needs to rebuild
        the item net without the dataset (`items.n_hot`, the sparse feature names / categorical columns / stored values)."""
        def entity(n_hot: int, id_map: IdMap, features: tp.Any) -> tp.Dict[str, tp.Any]:
            fs = None
            if features is not None:
                plain = lambda v: v.item() if hasattr(v, "item") and not isinstance(v, (str, bytes)) else v  # noqa: E731
                fs = {"names": [[plain(a), plain(b)] for a, b in features.names], "kind": "sparse",
                      "cat_feature_indices": features.cat_feature_indices.tolist(),
                      "cat_n_stored_values": int(features.get_cat_features().values.nnz)}
            return {"n_hot": int(n_hot), "id_map": {"size": int(id_map.size), "dtype": np.dtype(id_map.external_dtype).str},
                    "features": fs}

        return {"n_interactions": int(self.interactions.df.shape[0]),
                "users": entity(self.n_hot_users, self.user_id_map, None),
                "items": entity(self.n_hot_items, self.item_id_map, self.item_features)}

    @property
    def n_hot_users(self) -> int:
        """Users with interactions (dataset.py:176-184)."""
        return int(self.interactions.df[Columns.User].max()) + 1 if len(self.interactions.df) else 0

    @property
    def n_hot_items(self) -> int:
        """Items with interactions (dataset.py:188-199); items known only from features come after them."""
        return int(self.interactions.df[Columns.Item].max()) + 1 if len(self.interactions.df) else 0

    def get_hot_item_features(self) -> tp.Any:
        return None if self.item_features is None else self.item_features.take(np.arange(self.n_hot_items))

    def get_user_item_matrix(self, include_weights: bool = True, include_warm_users: bool = False,
                             include_warm_items: bool = False, dtype: tp.Any = np.float32) -> sparse.csr_matrix:
        return self.interactions.get_user_item_matrix(include_weights, self.user_id_map.size, self.item_id_map.size).astype(dtype)

    def get_raw_interactions(self, include_weight: bool = True, include_datetime: bool = True,
                             include_extra_cols: tp.Any = True) -> pd.DataFrame:
        return self.interactions.to_external(self.user_id_map, self.item_id_map, include_extra_cols)
