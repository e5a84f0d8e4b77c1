"""Host data path of the transformer models: dataset processing, CSR sequence store, collates.

Mirrors `TransformerDataPreparatorBase` / `SASRecDataPreparator` / `BERT4RecDataPreparator`
(rectools/models/nn/transformers/data_preparator.py:102-469, sasrec.py:51-166, bert4rec.py:51-193), with one
structural change: user sessions are NOT Python `List[List[int]]` (data_preparator.py:73-99, infeasible at 10 M
users) but a CSR store — `offsets[n_users+1]` + flat `items / weights / unix_ts` arrays — and the collates are
vectorised numpy over a batch of CSR slices, producing exactly the tensors the reference's collates produce
(pinned by tests/golden/collate_golden.npz).  Negatives are drawn on the device (uniform over real items,
negative_sampler.py:58-73; parity is distributional, as the reference itself resamples every batch).
"""
from __future__ import annotations

import os
import typing as tp
import warnings

import numpy as np
import pandas as pd
import torch
from scipy import sparse

from .dataset import Columns, Dataset, IdMap, Interactions, SparseFeatures

PADDING_VALUE = "PAD"
MASKING_VALUE = "MASK"


class SequenceStore:
    """CSR store of user sessions, ordered by user (in order of first appearance or sorted) then by time."""

    def __init__(self, offsets: np.ndarray, items: np.ndarray, weights: np.ndarray, unix_ts: tp.Optional[np.ndarray],
                 users: np.ndarray) -> None:
        self.offsets, self.items, self.weights, self.unix_ts, self.users = offsets, items, weights, unix_ts, users

    def __len__(self) -> int:
        return len(self.offsets) - 1

    @classmethod
    def from_interactions(cls, df: pd.DataFrame, sort_users: bool = False) -> "SequenceStore":
        """Same grouping as SequenceDataset.from_interactions (data_preparator.py:73-99): stable sort by datetime,
        group by user (in order of appearance unless `sort_users`)."""
        d = df.sort_values(Columns.Datetime, kind="stable")
        users = d[Columns.User].values
        if sort_users:
            order = np.argsort(users, kind="stable")
        else:
            codes, _ = pd.factorize(users)                   # codes count users in order of first appearance
            order = np.argsort(codes, kind="stable")
        users = users[order]
        items = d[Columns.Item].values[order].astype(np.int64)
        weights = d[Columns.Weight].values[order].astype(np.float32)
        ts = d["unix_ts"].values[order].astype(np.int64) if "unix_ts" in d.columns else None
        change = np.flatnonzero(np.r_[True, users[1:] != users[:-1]])
        offsets = np.r_[change, len(users)].astype(np.int64)
        return cls(offsets, items, weights, ts, users[change])

    def session(self, i: int) -> tp.Tuple[np.ndarray, np.ndarray]:
        return self.items[self.offsets[i]:self.offsets[i + 1]], self.weights[self.offsets[i]:self.offsets[i + 1]]


class DeviceSequenceStore:
    """`SequenceStore` resident in HBM (int64 offsets / items / timestamps, fp32 weights): `rt_collate` cuts batches out
    of it on the device — no per-step host gather, no H2D copy.  No CPU fallback."""

    def __init__(self, store: SequenceStore, device: tp.Any) -> None:
        dev = torch.device(device)
        if dev.type != "cuda":
            from . import _lib

            raise _lib.HipLibraryError("DeviceSequenceStore needs a HIP device (no CPU fallback)")
        self.offsets = torch.from_numpy(np.ascontiguousarray(store.offsets, dtype=np.int64)).to(dev)
        self.items = torch.from_numpy(np.ascontiguousarray(store.items, dtype=np.int64)).to(dev)
        self.weights = torch.from_numpy(np.ascontiguousarray(store.weights, dtype=np.float32)).to(dev)
        self.unix_ts = None if store.unix_ts is None else torch.from_numpy(np.ascontiguousarray(store.unix_ts, dtype=np.int64)).to(dev)
        self.device, self.n = dev, len(store)

    def __len__(self) -> int:
        return self.n

    @classmethod
    def from_device(cls, offsets: torch.Tensor, items: torch.Tensor, weights: torch.Tensor,
                    unix_ts: tp.Optional[torch.Tensor]) -> "DeviceSequenceStore":
        """Wrap arrays that were built on the device (recommend(): sessions sorted and grouped there)."""
        self = cls.__new__(cls)
        self.offsets, self.items, self.weights, self.unix_ts = offsets, items, weights, unix_ts
        self.device, self.n = offsets.device, int(offsets.numel()) - 1
        return self


def _device_collate(dstore: DeviceSequenceStore, idx: torch.Tensor, L: int, mode: int, with_ts: bool, probs=None,
                    rand_ids=None, mask_prob: float = 0.0, mask_id: int = 0) -> tp.Dict[str, torch.Tensor]:
    from . import ops

    B = int(idx.numel())
    dev = dstore.device
    train = mode in (0, 3)
    x = torch.empty((B, L), dtype=torch.int64, device=dev)
    y = torch.empty((B, L), dtype=torch.int64, device=dev) if train else None
    yw = torch.empty((B, L), dtype=torch.float32, device=dev) if train else None
    ts = torch.empty((B, L + 1), dtype=torch.int64, device=dev) if with_ts else None
    ops._c("rt_collate", dstore.offsets, dstore.items, dstore.weights, dstore.unix_ts if with_ts else None, idx, B, L, mode,
           probs, rand_ids, float(mask_prob), int(mask_id), x, y, yw, ts)
    out = {"x": x}
    if train:
        out["y"], out["yw"] = y, yw
    if with_ts:
        out["unix_ts"] = ts
    return out


def _tail_layout(lengths: np.ndarray, keep: int) -> tp.Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """For sessions truncated to their last `keep` elements: (row index, position inside the kept tail, kept length)."""
    kept = np.minimum(lengths, keep)
    rows = np.repeat(np.arange(len(lengths)), kept)
    starts = np.repeat(np.cumsum(kept) - kept, kept)
    pos = np.arange(int(kept.sum())) - starts
    return rows, pos, kept


class TransformerNegativeSamplerBase:
    """Plug-in seam of the reference (negative_sampler.py:23-46): subclass and pass as `negative_sampler_type`."""

    def __init__(self, n_negatives: int, **kwargs: tp.Any) -> None:
        self.n_negatives = n_negatives

    def get_negatives(self, batch_dict: tp.Dict[str, torch.Tensor], lowest_id: int, highest_id: int,
                      session_len_limit: tp.Optional[int] = None, **kwargs: tp.Any) -> torch.Tensor:
        raise NotImplementedError()


class CatalogUniformSampler(TransformerNegativeSamplerBase):
    """Negatives drawn uniformly from the real items [lowest_id, highest_id), without rejecting positives
    (negative_sampler.py:49-73).  The reference draws with `torch.randint` on the host inside its collate function and
    ships 26 MB per C2 batch over PCIe; here `rt_sample_negatives` (Philox4x32-10, csrc/rt_collate.hip) fills the
    [B, L | 1, N] tensor on the device the batch lives on.  Batch number c of a sampler seeded s is a pure function of
    (s, c): reproducible, and different for every batch.  Parity with the reference is distributional (it resamples every
    batch from torch's global generator)."""

    def __init__(self, n_negatives: int, seed: int = 0, **kwargs: tp.Any) -> None:
        super().__init__(n_negatives)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.calls = 0

    def get_negatives(self, batch_dict: tp.Dict[str, torch.Tensor], lowest_id: int, highest_id: int,
                      session_len_limit: tp.Optional[int] = None, **kwargs: tp.Any) -> torch.Tensor:
        from . import _lib, ops

        x = batch_dict["x"]
        if not x.is_cuda:
            raise _lib.HipLibraryError("CatalogUniformSampler draws on the HIP device of the batch (no CPU fallback)")
        session_len = session_len_limit if session_len_limit is not None else x.shape[1]
        out = torch.empty((x.shape[0], session_len, self.n_negatives), dtype=torch.int64, device=x.device)
        self.calls += 1
        ops._c("rt_sample_negatives", int(lowest_id), int(highest_id), out.numel(), self.seed, self.calls, out)
        return out


class TransformerDataPreparatorBase:
    train_session_max_len_addition: int = 0
    item_extra_tokens: tp.Sequence[tp.Hashable] = (PADDING_VALUE,)

    def __init__(self, session_max_len: int, batch_size: int, dataloader_num_workers: int = 0,
                 train_min_user_interactions: int = 2, get_val_mask_func: tp.Optional[tp.Callable] = None,
                 shuffle_train: bool = True, n_negatives: tp.Optional[int] = None, negative_sampler: tp.Any = None,
                 get_val_mask_func_kwargs: tp.Optional[dict] = None, extra_cols: tp.Optional[tp.List[str]] = None,
                 add_unix_ts: bool = False, **kwargs: tp.Any) -> None:
        self.session_max_len = session_max_len
        self.batch_size = batch_size
        self.dataloader_num_workers = dataloader_num_workers
        self.train_min_user_interactions = train_min_user_interactions
        self.get_val_mask_func = get_val_mask_func
        self.get_val_mask_func_kwargs = get_val_mask_func_kwargs or {}
        self.shuffle_train = shuffle_train
        self.n_negatives = n_negatives
        self.negative_sampler = negative_sampler
        self.extra_cols = extra_cols
        self.add_unix_ts = add_unix_ts
        self.item_id_map: IdMap
        self.train_dataset: Dataset
        self.val_interactions: tp.Optional[pd.DataFrame] = None
        self.extra_token_ids: tp.Dict[tp.Hashable, int] = {}

    # ---- id bookkeeping -------------------------------------------------------------------------------
    @property
    def n_item_extra_tokens(self) -> int:
        return len(self.item_extra_tokens)

    def get_known_items_sorted_internal_ids(self) -> np.ndarray:
        return self.item_id_map.get_sorted_internal()[self.n_item_extra_tokens:]

    def get_known_item_ids(self) -> np.ndarray:
        return self.item_id_map.get_external_sorted_by_internal()[self.n_item_extra_tokens:]

    @staticmethod
    def _to_unix_ts(datetime: pd.Series) -> np.ndarray:
        return (pd.to_datetime(datetime).values.astype("datetime64[ns]").astype("int64") / 10**9).astype("int64")

    # ---- train dataset --------------------------------------------------------------------------------
    def _filter_train_interactions(self, df: pd.DataFrame) -> pd.DataFrame:
        stats = df[Columns.User].value_counts()
        users = stats[stats >= self.train_min_user_interactions].index
        df = df[df[Columns.User].isin(users)]
        return (df.sort_values(Columns.Datetime, kind="stable").groupby(Columns.User, sort=False)
                .tail(self.session_max_len + self.train_session_max_len_addition))

    # ---- array fast path of process_dataset_train ------------------------------------------------------
    def _fast_path_applies(self, dataset: tp.Any) -> bool:
        if (self.extra_cols or []) or os.environ.get("RT_PREP", "arrays") == "pandas":
            return False   # extra interaction columns ride along in pandas
        df = getattr(getattr(dataset, "interactions", None), "df", None)
        if df is None or len(df) == 0:
            return False
        return (pd.api.types.is_integer_dtype(df[Columns.User].dtype) and pd.api.types.is_integer_dtype(df[Columns.Item].dtype)
                and pd.api.types.is_datetime64_any_dtype(df[Columns.Datetime].dtype))

    @staticmethod
    def _first_appearance(x: torch.Tensor, n_values: int) -> tp.Tuple[torch.Tensor, torch.Tensor]:
        """Distinct values of `x` in order of first appearance (what `pd.unique` returns) and the value -> rank lookup."""
        big = x.numel()
        first = torch.full((n_values,), big, dtype=torch.int64, device=x.device)
        first.scatter_reduce_(0, x, torch.arange(big, dtype=torch.int64, device=x.device), reduce="amin", include_self=True)
        vals = torch.nonzero(first < big).reshape(-1)
        uniq = vals[torch.sort(first[vals]).indices]
        lookup = torch.full((n_values,), -1, dtype=torch.int64, device=x.device)
        lookup[uniq] = torch.arange(uniq.numel(), dtype=torch.int64, device=x.device)
        return uniq, lookup

    prep_device: tp.Optional[tp.Union[str, torch.device]] = None      # set by the model: the device ITS parameters will live on

    def _prep_device(self) -> torch.device:
        """Where the sorts of `process_dataset_train` run: RT_PREP_DEVICE (override) > the device the model handed over (`prep_device`:
        its own `recommend_torch_device` / LOCAL_RANK's GPU — under torchrun every rank sorts on ITS GPU, not all of them on GPU 0) > the
        current HIP device of a stand-alone preparator > the host."""
        env = os.environ.get("RT_PREP_DEVICE")
        if env:
            return torch.device(env)
        if self.prep_device is not None:
            return torch.device(self.prep_device)
        return torch.device("cuda" if torch.cuda.is_available() else "cpu")

    def _process_dataset_train_arrays(self, dataset: tp.Any, train_rows: tp.Optional[np.ndarray] = None) -> None:
        """The same result as the frame-based path below (data_preparator.py:214-284), computed with sorts / scans over the
        interaction columns instead of pandas groupby: at ML-20M scale the frame path costs several epochs of GPU training.
        Works on the dataset's INTERNAL ids (integers whatever the external id type) and translates only the distinct ids.
        The ops are torch ops, identical on host and device (stable sorts: same result): they run on `_prep_device()` (2 M users x 1 M
        items, 78.8 M interactions on the MI355X box: 4.8 s against 14.0 s on its 128 host threads — profiles/r4_c4_scale.json) and are
        repeated on the host when the device cannot serve them — out of memory beside another process's tables, or
        `torch.use_deterministic_algorithms(True)`, under which the device's `bincount` / `scatter_reduce_` refuse to run."""
        dev = self._prep_device()
        if dev.type != "cpu":
            try:
                self._process_dataset_train_arrays_on(dev, dataset, train_rows)
                return
            except torch.cuda.OutOfMemoryError:
                warnings.warn(f"process_dataset_train: out of memory on {dev}; the dataset is processed on the host instead")
                torch.cuda.empty_cache()
            except RuntimeError as err:
                if "deterministic" not in str(err):
                    raise
                warnings.warn(f"process_dataset_train: {dev} has no deterministic implementation of a step "
                              f"(torch.use_deterministic_algorithms is on); the dataset is processed on the host instead")
            dev = torch.device("cpu")
        self._process_dataset_train_arrays_on(dev, dataset, train_rows)

    def _process_dataset_train_arrays_on(self, dev: torch.device, dataset: tp.Any, train_rows: tp.Optional[np.ndarray] = None) -> None:
        df = dataset.interactions.df
        as_t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
        u = as_t(df[Columns.User].values.astype(np.int64, copy=False))
        it = as_t(df[Columns.Item].values.astype(np.int64, copy=False))
        t_ns = df[Columns.Datetime].values.astype("datetime64[ns]").view(np.int64)
        t = as_t(t_ns)
        n_src_users, n_src_items = int(u.max()) + 1, int(it.max()) + 1
        # users with enough interactions (value_counts over ALL their interactions)
        if train_rows is not None:      # validation targets are taken out first (data_preparator.py:236-242)
            src = as_t(np.flatnonzero(train_rows))
            cnt = torch.bincount(u[src], minlength=n_src_users)
            idx0 = src[cnt[u[src]] >= self.train_min_user_interactions]
        else:
            cnt = torch.bincount(u, minlength=n_src_users)
            idx0 = torch.nonzero(cnt[u] >= self.train_min_user_interactions).reshape(-1)
        # stable sort by time, then the last (L + addition) rows of every user — in time order, as groupby(...).tail keeps them
        rows_time = idx0[torch.sort(t[idx0], stable=True).indices]
        u1 = u[rows_time]
        o2 = torch.sort(u1, stable=True).indices
        ends = torch.cumsum(torch.bincount(u1, minlength=n_src_users), 0)
        pos = torch.arange(o2.numel(), dtype=torch.int64, device=dev)
        keep_grouped = (ends[u1[o2]] - 1 - pos) < (self.session_max_len + self.train_session_max_len_addition)
        keep_time = torch.zeros(o2.numel(), dtype=torch.bool, device=dev)
        keep_time[o2[keep_grouped]] = True
        rows = rows_time[keep_time]
        uf, itf = u[rows], it[rows]
        uniq_u, look_u = self._first_appearance(uf, n_src_users)
        uniq_i, look_i = self._first_appearance(itf, n_src_items)
        ext_users = np.asarray(dataset.user_id_map.external_ids)[uniq_u.cpu().numpy()]
        ext_items = np.asarray(dataset.item_id_map.external_ids)[uniq_i.cpu().numpy()]
        user_id_map = IdMap.from_values(ext_users)
        item_id_map = IdMap.from_values(np.array(list(self.item_extra_tokens), dtype=object)).add_ids(ext_items)
        if item_id_map.size != self.n_item_extra_tokens + len(ext_items):   # an item shares its id with an extra token
            raise ValueError("item ids collide with the extra tokens of the model")
        new_u, new_i = look_u[uf], look_i[itf] + self.n_item_extra_tokens
        rows_np = rows.cpu().numpy()
        frame = {Columns.User: new_u.cpu().numpy(), Columns.Item: new_i.cpu().numpy(),
                 Columns.Weight: df[Columns.Weight].values[rows_np].astype(float),
                 Columns.Datetime: t_ns[rows_np].view("datetime64[ns]")}
        if self.add_unix_ts:
            frame["unix_ts"] = (t_ns[rows_np] / 10**9).astype("int64")      # the frame path's arithmetic (_to_unix_ts)
        final = Interactions(pd.DataFrame(frame, copy=False))
        item_features = None
        if getattr(dataset, "item_features", None) is not None:
            item_features = self._process_features_for_id_map(dataset.item_features, dataset.item_id_map, item_id_map,
                                                              self.n_item_extra_tokens)
        self.train_dataset = Dataset(user_id_map, item_id_map, final, item_features=item_features)
        self.item_id_map = item_id_map
        self.extra_token_ids = dict(zip(self.item_extra_tokens, item_id_map.convert_to_internal(list(self.item_extra_tokens))))
        self.val_interactions = None
        # the session store falls out of one more stable sort (users in id-map order = order of first appearance in time)
        o3 = torch.sort(new_u, stable=True).indices.cpu().numpy()
        users_sorted = frame[Columns.User][o3]
        change = np.flatnonzero(np.r_[True, users_sorted[1:] != users_sorted[:-1]])
        self._train_store = SequenceStore(np.r_[change, len(users_sorted)].astype(np.int64), frame[Columns.Item][o3].astype(np.int64),
                                          frame[Columns.Weight][o3].astype(np.float32),
                                          frame["unix_ts"][o3].astype(np.int64) if self.add_unix_ts else None, users_sorted[change])

    def _process_with_val_mask_arrays(self, dataset: tp.Any) -> None:
        """Validation mask (a user callable over the RAW frame, so that frame is still produced) + the array path for the
        training rows; the validation sessions are cut from the processed training frame (data_preparator.py:236-284)."""
        raw = dataset.get_raw_interactions()
        val_mask = np.asarray(self.get_val_mask_func(raw, **self.get_val_mask_func_kwargs), dtype=bool)
        self._process_dataset_train_arrays(dataset, train_rows=~val_mask)
        user_id_map, item_id_map = self.train_dataset.user_id_map, self.item_id_map
        val_targets = raw[val_mask]
        if self.add_unix_ts:
            val_targets = val_targets.assign(unix_ts=self._to_unix_ts(val_targets[Columns.Datetime]))
        val_targets = val_targets[val_targets[Columns.User].isin(user_id_map.external_ids)
                                  & val_targets[Columns.Item].isin(item_id_map.external_ids)]
        train_df = self.train_dataset.interactions.df
        val_users = user_id_map.convert_to_internal(val_targets[Columns.User].unique())
        history = train_df[np.isin(train_df[Columns.User].values, val_users)].copy()
        history[Columns.Weight] = 0.0
        targets = Interactions.from_raw(val_targets, user_id_map, item_id_map, keep_extra_cols=True).df
        self.val_interactions = pd.concat([history, targets[history.columns]], axis=0).reset_index(drop=True)

    def process_dataset_train(self, dataset: tp.Any) -> None:
        """data_preparator.py:229-284 — PAD (and MASK) first in the item id map so that PAD == 0."""
        self._train_store = None
        if dataset is not None and self._fast_path_applies(dataset):
            if self.get_val_mask_func is None:
                self._process_dataset_train_arrays(dataset)
            else:
                self._process_with_val_mask_arrays(dataset)
            return
        raw = dataset.get_raw_interactions()
        if self.add_unix_ts:
            raw["unix_ts"] = self._to_unix_ts(raw[Columns.Datetime])
        interactions = raw
        val_mask = None
        if self.get_val_mask_func is not None:
            val_mask = self.get_val_mask_func(raw, **self.get_val_mask_func_kwargs)
            interactions = raw[~val_mask].reset_index(drop=True)
        interactions = self._filter_train_interactions(interactions)
        user_id_map = IdMap.from_values(interactions[Columns.User].values)
        item_id_map = IdMap.from_values(np.array(list(self.item_extra_tokens), dtype=object))
        item_id_map = item_id_map.add_ids(interactions[Columns.Item])
        final = Interactions.from_raw(interactions, user_id_map, item_id_map, keep_extra_cols=True)
        item_features = None
        if getattr(dataset, "item_features", None) is not None:
            item_features = self._process_features_for_id_map(dataset.item_features, dataset.item_id_map, item_id_map,
                                                              self.n_item_extra_tokens)
        # user features are dropped: the models do not use them (data_preparator.py:261)
        self.train_dataset = Dataset(user_id_map, item_id_map, final, item_features=item_features)
        self.item_id_map = item_id_map
        self.extra_token_ids = dict(zip(self.item_extra_tokens, item_id_map.convert_to_internal(list(self.item_extra_tokens))))
        self.val_interactions = None
        if val_mask is not None:
            val_targets = raw[val_mask]
            val_targets = val_targets[val_targets[Columns.User].isin(user_id_map.external_ids)
                                      & val_targets[Columns.Item].isin(item_id_map.external_ids)]
            val_inter = interactions[interactions[Columns.User].isin(val_targets[Columns.User].unique())].copy()
            val_inter[Columns.Weight] = 0
            val_inter = pd.concat([val_inter, val_targets], axis=0)
            self.val_interactions = Interactions.from_raw(val_inter, user_id_map, item_id_map, keep_extra_cols=True).df

    @staticmethod
    def _process_features_for_id_map(raw_features: tp.Any, raw_id_map: tp.Any, id_map: IdMap, n_extra_tokens: int) -> tp.Any:
        """Item features re-indexed to the model's item ids, with empty rows for the extra tokens
        (data_preparator.py:194-212).  Sparse features only (the only kind `CatFeaturesItemNet` reads)."""
        if not hasattr(raw_features, "get_cat_features"):
            return None   # dense features: no categorical columns, the feature block is skipped (item_net.py:138-143)
        raw_internal = raw_id_map.convert_to_internal(id_map.get_external_sorted_by_internal()[n_extra_tokens:])
        taken = raw_features.take(raw_internal)
        vals = sparse.csr_matrix(taken.values)
        full = sparse.vstack([sparse.csr_matrix((n_extra_tokens, vals.shape[1]), dtype=vals.dtype), vals], format="csr")
        return SparseFeatures.from_iterables(full, taken.names)

    def train_store(self) -> SequenceStore:
        if getattr(self, "_train_store", None) is not None:
            return self._train_store
        return SequenceStore.from_interactions(self.train_dataset.interactions.df)

    def val_store(self) -> tp.Optional[SequenceStore]:
        return None if self.val_interactions is None else SequenceStore.from_interactions(self.val_interactions)

    # ---- recommend datasets ---------------------------------------------------------------------------
    def transform_dataset_u2i(self, dataset: tp.Any, users: tp.Any, context: tp.Optional[pd.DataFrame] = None) -> Dataset:
        """data_preparator.py:354-424"""
        df = dataset.interactions.df
        cols = Columns.Interactions + [c for c in (self.extra_cols or []) if c in df.columns]
        interactions = df[cols]
        users_internal = dataset.user_id_map.convert_to_internal(users, strict=False)
        items_internal = dataset.item_id_map.convert_to_internal(self.get_known_item_ids(), strict=False)
        interactions = interactions[interactions[Columns.User].isin(users_internal)]
        interactions = interactions[interactions[Columns.Item].isin(items_internal)].copy()
        interactions[Columns.Item] = dataset.item_id_map.convert_to_external(interactions[Columns.Item].values)
        interactions[Columns.User] = dataset.user_id_map.convert_to_external(interactions[Columns.User].values)
        rec_user_id_map = IdMap.from_values(interactions[Columns.User].values)
        if context is not None:
            if not pd.Series(np.asarray(users)).isin(context[Columns.User].unique()).all():
                raise ValueError("No context for some target users")
            if context.duplicated(subset=Columns.User).any():
                raise ValueError("Duplicated user entries found in context. Each user must have exactly one context row.")
            context = context.copy()
            context[Columns.Item] = PADDING_VALUE
            if Columns.Weight not in context.columns:
                context[Columns.Weight] = 0.0
            context = context[context[Columns.User].isin(interactions[Columns.User].unique())]
            interactions = pd.concat([interactions, context[[c for c in interactions.columns if c in context.columns]]])
        if self.add_unix_ts:
            interactions["unix_ts"] = self._to_unix_ts(interactions[Columns.Datetime])
        n_filtered = len(np.asarray(users)) - rec_user_id_map.size
        if n_filtered > 0:
            warnings.warn(f"{n_filtered} target users were considered cold because of missing known items")
        filtered = Interactions.from_raw(interactions, rec_user_id_map, self.item_id_map, keep_extra_cols=True)
        return Dataset(rec_user_id_map, self.item_id_map, filtered)

    def transform_dataset_i2i(self, dataset: tp.Any) -> Dataset:
        """data_preparator.py:426-451"""
        raw = dataset.get_raw_interactions()
        raw = raw[raw[Columns.Item].isin(self.get_known_item_ids())]
        return Dataset(dataset.user_id_map, self.item_id_map, Interactions.from_raw(raw, dataset.user_id_map, self.item_id_map, True))

    # ---- batches ----------------------------------------------------------------------------------------
    def add_negatives(self, batch: tp.Dict[str, torch.Tensor], validation: bool = False) -> tp.Dict[str, torch.Tensor]:
        """What the reference's collates do after building x / y / yw (sasrec.py:99-103,141-146; bert4rec.py:150-156,
        175-181): `negatives` [B, L, N] for a training batch, [B, 1, N] for a validation batch, from the plugged sampler."""
        if self.negative_sampler is not None:
            batch["negatives"] = self.negative_sampler.get_negatives(
                batch, lowest_id=self.n_item_extra_tokens, highest_id=self.item_id_map.size,
                session_len_limit=1 if validation else None)
        return batch

    def collate_train(self, store: SequenceStore, idx: np.ndarray) -> tp.Dict[str, np.ndarray]:
        raise NotImplementedError()

    def collate_val(self, store: SequenceStore, idx: np.ndarray) -> tp.Dict[str, np.ndarray]:
        raise NotImplementedError()

    def collate_recommend(self, store: SequenceStore, idx: np.ndarray) -> tp.Dict[str, np.ndarray]:
        raise NotImplementedError()

    # device-side twins of the two hot collates (same outputs as the host functions above, as device tensors)
    def collate_train_device(self, dstore: "DeviceSequenceStore", idx: torch.Tensor) -> tp.Dict[str, torch.Tensor]:
        raise NotImplementedError()

    def collate_recommend_device(self, dstore: "DeviceSequenceStore", idx: torch.Tensor) -> tp.Dict[str, torch.Tensor]:
        raise NotImplementedError()


def _gather_tails(store: SequenceStore, idx: np.ndarray, keep: int):
    """Flat views of the last `keep` elements of the selected sessions."""
    lo, hi = store.offsets[idx], store.offsets[idx + 1]
    lengths = hi - lo
    rows, pos, kept = _tail_layout(lengths, keep)
    src = np.repeat(hi - kept, kept) + pos
    return rows, pos, kept, src


class SASRecDataPreparator(TransformerDataPreparatorBase):
    """Shifted-sequence objective: x = session[:-1], y = session[1:] (sasrec.py:51-166)."""

    train_session_max_len_addition: int = 1

    def collate_train(self, store: SequenceStore, idx: np.ndarray) -> tp.Dict[str, np.ndarray]:
        B, L = len(idx), self.session_max_len
        rows, pos, kept, src = _gather_tails(store, idx, L + 1)
        n = np.repeat(kept, kept)
        x = np.zeros((B, L), np.int64); y = np.zeros((B, L), np.int64); yw = np.zeros((B, L), np.float32)
        col = L - (n - 1) + pos              # column of element `pos` when used as an input
        m_in = pos < n - 1
        x[rows[m_in], col[m_in]] = store.items[src[m_in]]
        m_out = pos >= 1
        y[rows[m_out], col[m_out] - 1] = store.items[src[m_out]]
        yw[rows[m_out], col[m_out] - 1] = store.weights[src[m_out]]
        out = {"x": x, "y": y, "yw": yw}
        if self.add_unix_ts:
            t = np.zeros((B, L + 1), np.int64)
            t[rows, (L + 1) - n + pos] = store.unix_ts[src]
            first = t[np.arange(B), (L + 1) - kept]
            pad = np.arange(L + 1)[None, :] < ((L + 1) - kept)[:, None]
            t = np.where(pad, first[:, None], t)
            out["unix_ts"] = t
        return out

    def collate_train_device(self, dstore: DeviceSequenceStore, idx: torch.Tensor) -> tp.Dict[str, torch.Tensor]:
        return _device_collate(dstore, idx, self.session_max_len, 0, self.add_unix_ts)

    def collate_recommend_device(self, dstore: DeviceSequenceStore, idx: torch.Tensor) -> tp.Dict[str, torch.Tensor]:
        return _device_collate(dstore, idx, self.session_max_len, 2 if self.add_unix_ts else 1, self.add_unix_ts)

    def collate_val(self, store: SequenceStore, idx: np.ndarray) -> tp.Dict[str, np.ndarray]:
        """sasrec.py:118-147: inputs are the zero-weight interactions, target the first non-zero-weight one."""
        B, L = len(idx), self.session_max_len
        x = np.zeros((B, L), np.int64); y = np.zeros((B, 1), np.int64); yw = np.zeros((B, 1), np.float32)
        t = np.zeros((B, L + 1), np.int64)
        for i, u in enumerate(idx):
            ses, w = store.session(int(u))
            inp = ses[w == 0][-L:]
            tgt = int(np.flatnonzero(w != 0)[0])
            x[i, L - len(inp):] = inp
            y[i, 0], yw[i, 0] = ses[tgt], w[tgt]
            if self.add_unix_ts:
                ts = store.unix_ts[store.offsets[u]:store.offsets[u + 1]]
                t[i, L + 1 - (len(ses) - 1):] = ts[1:][-(L + 1):] if len(ses) - 1 <= L + 1 else ts[-(L + 1):]
                n_pad = L + 2 - len(ses)
                if n_pad > 0:
                    t[i, :n_pad] = t[i, n_pad]
        out = {"x": x, "y": y, "yw": yw}
        if self.add_unix_ts:
            out["unix_ts"] = t
        return out

    def collate_recommend(self, store: SequenceStore, idx: np.ndarray) -> tp.Dict[str, np.ndarray]:
        B, L = len(idx), self.session_max_len
        x = np.zeros((B, L), np.int64)
        if self.add_unix_ts:  # last element of every session is the dummy context row (sasrec.py:154-163)
            rows, pos, kept, src = _gather_tails(store, idx, L + 1)
            n = np.repeat(kept, kept)
            m_in = pos < n - 1
            x[rows[m_in], (L - (n - 1) + pos)[m_in]] = store.items[src[m_in]]
            t = np.zeros((B, L + 1), np.int64)
            t[rows, (L + 1) - n + pos] = store.unix_ts[src]
            first = t[np.arange(B), (L + 1) - kept]
            pad = np.arange(L + 1)[None, :] < ((L + 1) - kept)[:, None]
            return {"x": x, "unix_ts": np.where(pad, first[:, None], t)}
        rows, pos, kept, src = _gather_tails(store, idx, L)
        n = np.repeat(kept, kept)
        x[rows, L - n + pos] = store.items[src]
        return {"x": x}


class BERT4RecDataPreparator(TransformerDataPreparatorBase):
    """Masked-item objective (bert4rec.py:51-193).  MASK is the second extra token (id 1)."""

    train_session_max_len_addition: int = 0
    item_extra_tokens: tp.Sequence[tp.Hashable] = (PADDING_VALUE, MASKING_VALUE)

    def __init__(self, *args: tp.Any, mask_prob: float = 0.15, **kwargs: tp.Any) -> None:
        super().__init__(*args, **kwargs)
        self.mask_prob = mask_prob

    def _mask_session(self, ses: np.ndarray, first_border: float = 0.8, second_border: float = 0.9):
        """bert4rec.py:109-127 with the reference's exact np.random call order (rand per session, randint per
        randomly replaced element), so that a seeded run masks the same positions."""
        masked, target = ses.copy(), ses.copy()
        probs = np.random.rand(len(ses))
        for j in range(len(ses)):
            if probs[j] < self.mask_prob:
                pj = probs[j] / self.mask_prob
                if pj < first_border:
                    masked[j] = self.extra_token_ids[MASKING_VALUE]
                elif pj < second_border:
                    masked[j] = np.random.randint(low=self.n_item_extra_tokens, high=self.item_id_map.size)
            else:
                target[j] = 0
        return masked, target

    def collate_train_with_draws(self, store: SequenceStore, idx: np.ndarray, probs: np.ndarray, rand_ids: np.ndarray,
                                 first_border: float = 0.8, second_border: float = 0.9) -> tp.Dict[str, np.ndarray]:
        """`collate_train` as a pure function of the random draws (probs [B,L] uniform, rand_ids [B,L] item ids, both
        indexed by output column): the statement the device kernel (`rt_collate` mode 3) is checked against."""
        B, L = len(idx), self.session_max_len
        x = np.zeros((B, L), np.int64); y = np.zeros((B, L), np.int64); yw = np.zeros((B, L), np.float32)
        mask_id = self.extra_token_ids[MASKING_VALUE]
        mp = np.float32(self.mask_prob)
        for i, u in enumerate(idx):
            ses, w = store.session(int(u))
            ses, w = ses[-L:], w[-L:]
            c0 = L - len(ses)
            pr = probs[i, c0:].astype(np.float32)
            hit = pr < mp
            pj = pr / mp
            masked = np.where(hit & (pj < np.float32(first_border)), mask_id,
                              np.where(hit & (pj < np.float32(second_border)), rand_ids[i, c0:], ses))
            x[i, c0:] = masked
            y[i, c0:] = np.where(hit, ses, 0)
            yw[i, c0:] = w
        return {"x": x, "y": y, "yw": yw}

    def collate_train_device(self, dstore: DeviceSequenceStore, idx: torch.Tensor) -> tp.Dict[str, torch.Tensor]:
        B, L = int(idx.numel()), self.session_max_len
        probs = torch.rand((B, L), dtype=torch.float32).to(dstore.device)      # host draws, as the reference's collate (bert4rec.py:109-127)
        rand_ids = torch.randint(self.n_item_extra_tokens, self.item_id_map.size, (B, L), dtype=torch.int64, device=dstore.device)
        return _device_collate(dstore, idx, L, 3, False, probs, rand_ids, self.mask_prob, self.extra_token_ids[MASKING_VALUE])

    def collate_recommend_device(self, dstore: DeviceSequenceStore, idx: torch.Tensor) -> tp.Dict[str, torch.Tensor]:
        return _device_collate(dstore, idx, self.session_max_len, 4, False, mask_id=self.extra_token_ids[MASKING_VALUE])

    def collate_train(self, store: SequenceStore, idx: np.ndarray) -> tp.Dict[str, np.ndarray]:
        B, L = len(idx), self.session_max_len
        x = np.zeros((B, L), np.int64); y = np.zeros((B, L), np.int64); yw = np.zeros((B, L), np.float32)
        for i, u in enumerate(idx):
            ses, w = store.session(int(u))
            ses, w = ses[-L:], w[-L:]   # train sessions already hold at most L items (addition = 0)
            masked, target = self._mask_session(ses)
            x[i, L - len(ses):] = masked
            y[i, L - len(ses):] = target
            yw[i, L - len(ses):] = w
        return {"x": x, "y": y, "yw": yw}

    def collate_val(self, store: SequenceStore, idx: np.ndarray) -> tp.Dict[str, np.ndarray]:
        B, L = len(idx), self.session_max_len
        x = np.zeros((B, L), np.int64); y = np.zeros((B, 1), np.int64); yw = np.zeros((B, 1), np.float32)
        for i, u in enumerate(idx):
            ses, w = store.session(int(u))
            inp = np.r_[ses[w == 0], self.extra_token_ids[MASKING_VALUE]][-L:]
            tgt = int(np.flatnonzero(w != 0)[0])
            x[i, L - len(inp):] = inp
            y[i, 0], yw[i, 0] = ses[tgt], w[tgt]
        return {"x": x, "y": y, "yw": yw}

    def collate_recommend(self, store: SequenceStore, idx: np.ndarray) -> tp.Dict[str, np.ndarray]:
        """history[-(L-1):] + [MASK], left padded (bert4rec.py:182-193)."""
        B, L = len(idx), self.session_max_len
        x = np.zeros((B, L), np.int64)
        rows, pos, kept, src = _gather_tails(store, idx, L - 1)
        n = np.repeat(kept, kept)
        x[rows, (L - 1) - n + pos] = store.items[src]
        x[:, L - 1] = self.extra_token_ids[MASKING_VALUE]
        return {"x": x}


def epoch_permutation(n: int, epoch: int, seed: int, shuffle: bool) -> np.ndarray:
    """Sample order of one epoch; identical on every rank (then sharded as perm[rank::world], DistributedSampler-style)."""
    if not shuffle:
        return np.arange(n)
    return np.random.default_rng(seed + 1000003 * epoch).permutation(n)


def shard_indices(perm: np.ndarray, rank: int, world: int) -> np.ndarray:
    """DistributedSampler semantics: pad to a multiple of `world` by wrapping, then take every world-th sample."""
    if world <= 1:
        return perm
    total = -(-len(perm) // world) * world
    if total > len(perm):
        perm = np.r_[perm, perm[: total - len(perm)]]
    return perm[rank::world]
