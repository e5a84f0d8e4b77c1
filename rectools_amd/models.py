"""`SASRecModel`, `BERT4RecModel`, `HSTUModel` — the reference's public model API on the MI355X engine.

Same constructor arguments, `fit` / `fit_partial` / `recommend` / `recommend_to_items` / `get_config` / `from_config` /
`get_params` / `save` / `load` contract and result frames as `rectools.models.SASRecModel` etc.
(rectools/models/nn/transformers/base.py:241-724, sasrec.py:315-541, bert4rec.py:204-452, hstu.py:412-729,
rectools/models/base.py:326-519), with PyTorch-Lightning replaced by a native loop (`_run_epochs`): per batch
vectorised collate -> H2D -> HIP forward/backward (rectools_amd.ops) -> one fused Adam kernel, and — under
torch.distributed — DistributedSampler-style sharding plus ONE RCCL all-reduce of the flat gradient per step.
recommend() keeps user and item embeddings on the device and ranks them with the exact top-k HIP kernel.
"""
from __future__ import annotations

import os

import importlib
import io
import pickle
import typing as tp
import warnings

import numpy as np
import pandas as pd
import torch

from . import checkpoint as ckpt
from . import lightning as hl
from . import nn as hnn
from . import ops
from .data_preparator import (MASKING_VALUE, BERT4RecDataPreparator, CatalogUniformSampler, DeviceSequenceStore, SASRecDataPreparator,
                              SequenceStore, TransformerDataPreparatorBase, TransformerNegativeSamplerBase, epoch_permutation,
                              shard_indices)
from .dataset import Columns
from .rank import DeviceCSR, Distance, HipRanker


class NotFittedError(Exception):
    """Raised when `recommend` is called before `fit` (rectools/exceptions.py)."""

    def __init__(self, model_name: str) -> None:
        super().__init__(f"{model_name} isn't fitted, call method `fit` first.")


def _full_path(obj: tp.Any) -> str:
    return f"{obj.__module__}.{obj.__qualname__}"


def _import_object(path: tp.Union[str, tp.Any]) -> tp.Any:
    if not isinstance(path, str):
        return path
    if "." not in path:      # the reference writes bare model class names into its configs ("SASRecModel")
        return getattr(importlib.import_module(__name__), path)
    module, name = path.rsplit(".", 1)
    return getattr(importlib.import_module(module), name)


def _dist_info() -> tp.Tuple[int, int]:
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class _AttrView:
    """A checkpoint's `dataset_schema` dict with attribute access, nested: the reference hands `from_dataset_schema` a `DatasetSchema`
    object and its blocks read `dataset_schema.items.n_hot`, `.items.features.cat_feature_indices` ... (item_net.py:44-52, 193-228;
    ADVICE r5: a raw dict raised AttributeError there).  Attribute access wins over dict methods (`.items` is the ITEM schema, as on the
    reference's object); subscripting, `in`, `len`, iteration, `.keys()`, `.get()` and `dict(view)` still read the mapping."""

    def __init__(self, data: tp.Dict[str, tp.Any]) -> None:
        object.__setattr__(self, "_data", data)

    def __getattr__(self, name: str) -> tp.Any:
        data = object.__getattribute__(self, "_data")
        if name in data:
            return _schema_view(data[name])
        raise AttributeError(name)

    def __getitem__(self, key: str) -> tp.Any:
        return _schema_view(self._data[key])

    def __contains__(self, key: object) -> bool:
        return key in self._data

    def __iter__(self) -> tp.Iterator[str]:
        return iter(self._data)

    def __len__(self) -> int:
        return len(self._data)

    def keys(self) -> tp.Any:
        return self._data.keys()

    def get(self, key: str, default: tp.Any = None) -> tp.Any:
        return _schema_view(self._data.get(key, default))

    def __repr__(self) -> str:
        return f"_AttrView({self._data!r})"


def _schema_view(obj: tp.Any) -> tp.Any:
    if isinstance(obj, dict) and not isinstance(obj, _AttrView):
        return _AttrView(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_schema_view(v) for v in obj)
    return obj


# the data preparators whose batches have a packed (padding-free) twin on the device: `rt_collate_packed` / `rt_collate_packed_bert`
_PACKED_PREPARATORS = ("SASRecDataPreparator", "BERT4RecDataPreparator")


def _packed_hooks_are_stock(lm: tp.Any) -> bool:
    """The packed training step calls `training_loss_packed` -> `encode_packed_train` instead of the reference-shaped hooks
    `training_loss` -> `_encode` -> `encode_sessions`.  A plugged `lightning_module_type` / `backbone_type` that overrides one of
    those hooks (and does not bring its own packed twin) must therefore keep the padded path — the one its override runs on."""
    tl, tm = type(lm), type(lm.torch_model)
    base_l, base_b = hl.TransformerLossModule, hnn.TransformerTorchBackbone
    loss_ok = (tl.training_loss is base_l.training_loss and tl._encode is base_l._encode) \
        or tl.training_loss_packed is not base_l.training_loss_packed      # pylint: disable=protected-access
    enc_ok = tm.encode_sessions is base_b.encode_sessions or tm.encode_packed_train is not base_b.encode_packed_train
    # a plugged similarity module is called with the reference's shapes ([B, L, d] sessions, [B, L, 1 + N] candidates): padded batches
    sim_ok = hnn.similarity_is_stock(lm.torch_model.similarity_module) or tl.training_loss_packed is not base_l.training_loss_packed
    return loss_ok and enc_ok and sim_ok


_PREFETCH_STREAMS: tp.Dict[torch.device, "torch.cuda.Stream"] = {}


class _TrainLoop:
    """One rank's training stream: the session store lives in HBM (`DeviceSequenceStore`), an epoch is a permutation of
    the sessions sharded DistributedSampler-style, and `step()` is one optimiser step on the next batch — device collate
    (`rt_collate`), negatives (`rt_sample_negatives` through the plugged sampler), forward + loss + backward, gradient
    exchange and the fused Adam kernel (lightning.py:311-321 + sasrec.py:86-104 + negative_sampler.py:58-73 per step)."""

    def __init__(self, model: "TransformerModelBase") -> None:
        lm, opt = model.lightning_model, model.optimizer
        assert lm is not None and opt is not None
        self.model, self.lm, self.opt, self.dp = model, lm, opt, model.data_preparator
        self.device = next(lm.parameters()).device
        self.rank, self.world = _dist_info()
        self.store = self.dp.train_store()
        self.dstore = DeviceSequenceStore(self.store, self.device)
        self.seed = 0 if model.seed is None else int(model.seed)
        self.batch_size = model.batch_size
        self.epoch = -1
        self.mine_t: tp.Optional[torch.Tensor] = None
        self.pos = 0
        self.sequences_done = 0   # sessions consumed by step() so far (the last batch of an epoch may be short)
        # packed training batches (no padding rows): the default wherever the stack offers it (causal SASRec blocks,
        # head size 32 / 64); RT_PACKED_TRAIN=0 keeps the padded [B, L] window (the cross-check of tests/test_packed_gpu.py)
        tm = lm.torch_model
        # data parallel, all-reduce exchange: the block weights' gradients leave while the lookup's backward still runs (FlatAdam.set_early_bucket)
        tm.on_input_gradient = None
        if self.world > 1 and hasattr(tm, "on_input_gradient"):
            opt.set_early_bucket([*tm.item_model.parameters(), *tm.pos_encoding_layer.parameters()])
            world = self.world
            tm.on_input_gradient = lambda: opt.begin_early_exchange(world)
        self.bert = type(self.dp).__name__ == "BERT4RecDataPreparator"
        self.stu = isinstance(tm.transformer_layers, hnn.STULayers)      # the only stack that reads the batch's timestamps
        self.packed = (os.environ.get("RT_PACKED_TRAIN", "1") != "0" and type(self.dp).__name__ in _PACKED_PREPARATORS
                       and (not self.dp.add_unix_ts or (self.stu and not self.bert)) and not (self.dp.extra_cols or [])
                       and getattr(tm.transformer_layers, "packed_ok", None) is not None
                       and getattr(tm, "_fused_pos", lambda: False)() and _packed_hooks_are_stock(lm)
                       and tm.transformer_layers.packed_ok(model.n_factors, self.dp.session_max_len, tm.use_causal_attn,
                                                           tm.use_key_padding_mask))
        # a stack whose pad rows carry state (LiGR without key-padding masks: default eSASRec) packs behind a SHARED PAD PREFIX: the
        # window's pad rows ride along once per batch as one more packed session (`nn.LiGRLayers.packed_mode`)
        mode = getattr(tm.transformer_layers, "packed_mode", None)
        self.prefix = bool(self.packed and not self.bert and mode is not None and
                           mode(model.n_factors, self.dp.session_max_len, tm.use_causal_attn, tm.use_key_padding_mask) == "prefix")
        self._prefix_dist: tp.Optional[torch.Tensor] = None
        self._native: tp.Optional[hl.NativeSasrecStep] = None
        self._native_planned = False

    # The next batch is cut while the current step's backward pass runs: collate + negative sampling are a handful of tiny launches that
    # depend on the store and the sampler's counter only — issued on their own stream right behind `loss.backward()` they run beside the
    # backward kernels instead of in front of the next forward pass (~14 us of a 1.48 ms C2 step).  Never across an epoch boundary.
    prefetch_batches = True

    def begin_epoch(self, epoch: int) -> None:
        self._pending = None
        self._native_planned = False      # (the model may have been touched between epochs: callbacks, a checkpoint load)
        perm = epoch_permutation(len(self.store), epoch, self.seed, self.dp.shuffle_train)
        mine = shard_indices(perm, self.rank, self.world)
        self.mine_t = torch.from_numpy(np.ascontiguousarray(mine, dtype=np.int64)).to(self.device)   # one small H2D per epoch
        self.epoch, self.pos = epoch, 0
        if self.packed:
            # packed row offsets of EVERY batch of the epoch, cut on the host from the store's offsets (numpy, vectorised) and
            # uploaded once: a step then needs no device -> host round trip to learn its row count
            B, L = self.batch_size, self.dp.session_max_len
            off = np.asarray(self.store.offsets, dtype=np.int64)
            lens = np.clip(off[mine + 1] - off[mine] - (0 if self.bert else 1), 0, L)   # SASRec: x = tail[:-1]; BERT4Rec: the tail itself
            nb = -(-len(mine) // B)
            # longest session first INSIDE every batch (which sessions form a batch is untouched; their order in it carries no
            # meaning for the loss): the attention kernels run one workgroup per (session, head) in row order, a workgroup's work grows
            # with the square of its session's length, and 512 workgroups on 256 CUs finish earliest when the heavy ones start first
            # (longest-processing-time order) — in random order the last CU to finish was handed two long sessions back to back
            key = (np.arange(len(mine)) // B) * (L + 1) + (L - lens)
            srt = np.argsort(key, kind="stable")
            mine, lens = mine[srt], lens[srt]
            # BERT4Rec draws its masks per slot of the batch: every session keeps the draws of the slot the unsorted batch gave it
            self._slots_host = np.ascontiguousarray(srt % B, dtype=np.int64)
            self._slots = torch.from_numpy(self._slots_host).to(self.device) if self.bert else None
            self.mine_t = torch.from_numpy(np.ascontiguousarray(mine, dtype=np.int64)).to(self.device)
            grid = np.zeros((nb, B), dtype=np.int64)
            grid.reshape(-1)[:len(mine)] = lens
            cu = np.zeros((nb, B + (3 if self.prefix else 2)), dtype=np.int64)
            np.cumsum(grid, axis=1, out=cu[:, 1:B + 1])
            if self.prefix:      # [.. real sessions .., + the window's pad rows as session B, + the tail]
                cu[:, B + 1] = cu[:, B] + L
            # one more entry per batch: the end of the unused tail (rows up to the 128-row GEMM tile).  The attention kernels take the
            # tail as one more "session" — its rows get finite values and zero gradients instead of three memsets per block and step
            cu[:, -1] = np.maximum((cu[:, -2] + 127) // 128 * 128, 128)
            self._cu_host = cu
            self._cu_dev = torch.from_numpy(cu).to(self.device)
            self._reserve_step_memory(int(cu[:, -1].max()))

    def _reserve_step_memory(self, max_rows: int) -> None:
        """Packed batches change their row count every step, and torch's caching allocator answers a size it has not seen with a
        synchronous hipMalloc — dozens of them over the first steps of a run.  The host knows the epoch's largest batch: take ONE
        block that covers a step of that size and hand it back, the allocator then carves every request of every step out of it
        (large blocks are split and re-merged).  ~45 row-sized fp32 buffers are live at the peak of a 2-block step."""
        if self.device.type != "cuda" or getattr(self, "_reserved_rows", 0) >= max_rows:
            return
        d = int(self.model.n_factors)
        n_blocks = max(int(self.model.n_blocks), 1)
        n_neg = int(self.model.n_negatives or 0) if self.dp.negative_sampler is not None else 0
        rows = (max_rows + 127) // 128 * 128
        # row-sized fp32 buffers live at the peak of a step, in units of d per block: the SASRec block keeps ~12; a Pre-LN block's 4d FFN,
        # an STU block's 4 x (u, v, q, k) and a LiGR block's gates + three 4d SwiGLU rows keep more (forward state of every block + one
        # block's backward temporaries) — too small a reservation and the allocator meets new sizes in the timed steps
        units = {"LiGRLayers": 46, "STULayers": 28, "PreLNTransformerLayers": 24}.get(type(self.lm.torch_model.transformer_layers).__name__, 12)
        per_row = 4 * d * (units * n_blocks + 12) + (8 + 4) * (n_neg + 1) + 64
        table = 4 * d * int(self.dp.item_id_map.size) * 3
        try:
            block = torch.empty((int(rows * per_row * 1.25) + table,), dtype=torch.uint8, device=self.device)
            del block
        except RuntimeError:     # not enough free memory for the reservation: the allocator grows step by step instead
            pass
        self._reserved_rows = max_rows

    def _batches_uncut(self) -> int:
        return 0 if self.mine_t is None else -(-(int(self.mine_t.numel()) - self.pos) // self.batch_size)

    def batches_left(self) -> int:
        return self._batches_uncut() + (1 if getattr(self, "_pending", None) is not None else 0)

    def _next_indices(self) -> torch.Tensor:
        """Session indices of the next batch (rolls over to the next epoch when the current one is used up)."""
        if self._batches_uncut() == 0:
            self.begin_epoch(self.epoch + 1)
        assert self.mine_t is not None
        idx = self.mine_t[self.pos:self.pos + self.batch_size]
        self.sequences_done += int(idx.numel())
        self.pos += self.batch_size
        return idx

    def _packed_batch(self, idx: torch.Tensor) -> tp.Dict[str, tp.Any]:
        """The SASRec training batch of the sessions `idx` without padding rows (`rt_collate_packed`): x / y / yw / dist over the
        real positions, padded to the 128-row GEMM tile with (id 0, target 0) rows; negatives from the plugged sampler for every
        row.  Row offsets and row count come from the epoch's host-side table."""
        bi = self.pos // self.batch_size - 1          # `_next_indices` has advanced `pos` past this batch
        nb = int(idx.numel())
        n = int(self._cu_host[bi, nb])
        rows = max((n + 127) // 128 * 128, 128)
        cu = self._cu_dev[bi, :nb + 1]
        full = nb == self.batch_size
        if self.bert:   # the draws of `BERT4RecDataPreparator.collate_train_device`, in its order: the packed batch masks the same positions
            L = self.dp.session_max_len
            probs_h = torch.rand((nb, L), dtype=torch.float32)      # drawn on the host (as `collate_train_device` does): the number of
            #                                                           masked positions sizes the loss's buffers, and the host can count
            #                                                           them here instead of asking the device (`torch.nonzero`: a sync)
            slots_h = self._slots_host[self.pos - self.batch_size:self.pos - self.batch_size + nb]
            n_of_row = np.zeros((nb,), dtype=np.int64)
            n_of_row[slots_h] = np.diff(self._cu_host[bi, :nb + 1])
            hit = (probs_h.numpy() < np.float32(self.dp.mask_prob)) & (np.arange(L)[None, :] >= (L - n_of_row)[:, None])
            n_targets = int(hit.sum())
            probs = probs_h.to(self.device, non_blocking=True)
            rand_ids = torch.randint(self.dp.n_item_extra_tokens, self.dp.item_id_map.size, (nb, L), dtype=torch.int64, device=self.device)
            x, y, yw, dist = ops.collate_packed_bert(self.dstore.offsets, self.dstore.items, self.dstore.weights, idx, cu, rows, L, True,
                                                     self.dp.extra_token_ids[MASKING_VALUE], probs, rand_ids, self.dp.mask_prob,
                                                     draw_rows=self._slots[self.pos - self.batch_size:self.pos - self.batch_size + nb])
        elif self.prefix:
            # the sessions, then the shared pad prefix: `window` rows of (id 0, no target, positions window - 1 .. 0) — collate_packed
            # writes (id 0, target 0) behind the sessions anyway, the positions are set here
            L = self.dp.session_max_len
            rows = max((n + L + 127) // 128 * 128, 128)
            x, y, yw, dist = ops.collate_packed(self.dstore.offsets, self.dstore.items, self.dstore.weights, idx, cu, rows, train=True)
            if self._prefix_dist is None:
                self._prefix_dist = torch.arange(L - 1, -1, -1, dtype=dist.dtype, device=self.device)
            dist[n:n + L] = self._prefix_dist
            tail = self._cu_dev[bi, -2:] if full else torch.tensor([n + L, rows], dtype=torch.int64, device=self.device)
            cu_all = torch.cat([cu, tail])                       # [nb + 3]: sessions, prefix, tail
            batch = {"x": x, "y": y, "yw": yw, "dist": dist, "cu": cu_all[:nb + 2], "window": L, "n_rows": n + L, "n_prefixed": nb}
            if rows > n + L:
                batch["cu_attn"] = cu_all
            if self.dp.negative_sampler is not None:
                batch["negatives"] = self.dp.negative_sampler.get_negatives(
                    {"x": x.view(-1, 1)}, lowest_id=self.dp.n_item_extra_tokens, highest_id=self.dp.item_id_map.size)
            return batch
        else:
            x, y, yw, dist = ops.collate_packed(self.dstore.offsets, self.dstore.items, self.dstore.weights, idx, cu, rows, train=True)
        batch: tp.Dict[str, tp.Any] = {"x": x, "y": y, "yw": yw, "dist": dist, "cu": cu, "window": self.dp.session_max_len, "n_rows": n}
        if self.bert:
            batch["n_targets"] = n_targets
        if self.dp.add_unix_ts:      # the n + 1 timestamps of every session (sasrec.py:96-104: the rows' items and the last row's target)
            batch["ts"] = ops.collate_packed_ts(self.dstore.offsets, self.dstore.unix_ts, idx, cu, n)
        if full and rows > n and rows - n <= self.dp.session_max_len and not self.stu:
            batch["cu_attn"] = self._cu_dev[bi]        # [B + 2]: the sessions + the tail as a session of its own (see begin_epoch)
        if self.dp.negative_sampler is not None:
            batch["negatives"] = self.dp.negative_sampler.get_negatives(
                {"x": x.view(-1, 1)}, lowest_id=self.dp.n_item_extra_tokens, highest_id=self.dp.item_id_map.size)
        return batch

    def _cut_batch(self) -> tp.Dict[str, tp.Any]:
        idx = self._next_indices()
        return self._packed_batch(idx) if self.packed else self.dp.add_negatives(self.dp.collate_train_device(self.dstore, idx))

    def _prefetch(self) -> None:
        """Cut the next batch of THIS epoch on the prefetch stream (see `prefetch_batches`)."""
        if not self.prefetch_batches or self.device.type != "cuda" or self._batches_uncut() == 0:
            return
        main = torch.cuda.current_stream(self.device)
        side = _PREFETCH_STREAMS.get(self.device)
        if side is None:
            # ONE per device and process, like the weight-gradient streams (ops._SIDE, csrc/rt_block.hip).  Measured, mechanism not
            # pinned down: with a stream per loop, the fourth loop of a process ran its weight-gradient products and the prefetch
            # serialised (HSTU C4 17.4 k seqs/s against 21.9 k as the first loop, same kernels, GPU time 7.3 vs 5.8 ms per step;
            # GPU_MAX_HW_QUEUES = 2 / 8 changes neither figure); with the one stream every loop runs like the first
            side = _PREFETCH_STREAMS[self.device] = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(side):
            batch = self._cut_batch()
            done = side.record_event()
        for v in batch.values():      # allocated on the prefetch stream, consumed on the main one
            if isinstance(v, torch.Tensor) and v.is_cuda:
                v.record_stream(main)
        self._pending = (batch, done)

    def _native_step(self) -> tp.Optional[hl.NativeSasrecStep]:
        """The compiled step of the stock packed SASRec configuration, or None (planned once per epoch: `begin_epoch` clears it)."""
        if self._native_planned:
            return self._native
        self._native_planned = True
        self._native = None
        if self.packed and self.world == 1 and not (self.bert or self.prefix or self.stu) and self.device.type == "cuda":
            self._native = hl.NativeSasrecStep.plan(self.lm, self.opt)
        return self._native

    def step(self) -> torch.Tensor:
        """One training step on the next batch of the current epoch (rolls over to the next epoch when it is used up)."""
        pending = getattr(self, "_pending", None)
        if pending is not None:
            batch, done = pending
            self._pending = None
            torch.cuda.current_stream(self.device).wait_event(done)
        else:
            batch = self._cut_batch()
        ops.RNG.next_step()
        native = self._native_step()
        if native is not None and native.ready(batch):
            # the stock packed SASRec step: forward, loss, backward and Adam issued by compiled code (`lightning.NativeSasrecStep`)
            loss = native.forward_backward(batch)
            self._prefetch()
            native.adam()
            return loss
        self.opt.zero_grad()
        loss = self.lm.training_loss_packed(batch) if self.packed else self.lm.training_loss(batch)
        one = getattr(self, "_root_grad", None)
        if one is None or one.device != loss.device or one.shape != loss.shape:
            one = self._root_grad = torch.ones_like(loss)      # the root gradient, made once (autograd fills a fresh one per call: a launch)
        loss.backward(one)
        self._prefetch()
        self.opt.step(self.world)
        return loss


class TransformerModelBase:
    """Config shell + native training loop + device-resident recommend."""

    train_loss_name = "train_loss"
    val_loss_name = "val_loss"
    u2i_dist_default = "dot"
    use_scale_factor_default = False
    recommends_for_warm = False
    recommends_for_cold = False

    def __init__(  # pylint: disable=too-many-arguments, too-many-locals
        self,
        data_preparator_type: tp.Type[TransformerDataPreparatorBase],
        transformer_layers_type: tp.Type[hnn.TransformerLayersBase] = hnn.PreLNTransformerLayers,
        n_blocks: int = 2, n_heads: int = 4, n_factors: int = 256, use_pos_emb: bool = True, use_causal_attn: bool = False,
        use_key_padding_mask: bool = False, dropout_rate: float = 0.2, session_max_len: int = 100,
        dataloader_num_workers: int = 0, batch_size: int = 128, loss: str = "softmax", n_negatives: int = 1,
        gbce_t: float = 0.2, lr: float = 0.001, epochs: int = 3, verbose: int = 0, deterministic: bool = False,
        recommend_batch_size: int = 256, recommend_torch_device: tp.Optional[str] = None,
        train_min_user_interactions: int = 2,
        similarity_module_type: tp.Type[hnn.DistanceSimilarityModule] = hnn.DistanceSimilarityModule,
        item_net_block_types: tp.Sequence[tp.Type[torch.nn.Module]] = (hnn.IdEmbeddingsItemNet, hnn.CatFeaturesItemNet),
        item_net_constructor_type: tp.Type[hnn.SumOfEmbeddingsConstructor] = hnn.SumOfEmbeddingsConstructor,
        item_net_constructor_kwargs: tp.Optional[dict] = None,
        get_val_mask_func: tp.Optional[tp.Callable] = None, get_val_mask_func_kwargs: tp.Optional[dict] = None,
        data_preparator_kwargs: tp.Optional[dict] = None, transformer_layers_kwargs: tp.Optional[dict] = None,
        pos_encoding_kwargs: tp.Optional[dict] = None, lightning_module_kwargs: tp.Optional[dict] = None,
        similarity_module_kwargs: tp.Optional[dict] = None, seed: tp.Optional[int] = None,
        negative_sampler_type: tp.Type[TransformerNegativeSamplerBase] = CatalogUniformSampler,
        negative_sampler_kwargs: tp.Optional[dict] = None,
        pos_encoding_type: tp.Type[torch.nn.Module] = hnn.LearnableInversePositionalEncoding,
        lightning_module_type: tp.Type[hl.TransformerLossModule] = hl.TransformerLossModule,
        backbone_type: tp.Type[hnn.TransformerTorchBackbone] = hnn.TransformerTorchBackbone,
        backbone_kwargs: tp.Optional[dict] = None,
        get_trainer_func: tp.Optional[tp.Callable] = None, get_trainer_func_kwargs: tp.Optional[dict] = None,
        **kwargs: tp.Any,
    ) -> None:
        # get_trainer_func (transformers/base.py:367-380): the reference hands fit() to the pytorch_lightning.Trainer this factory returns.
        # The engine keeps its own loop (`_TrainLoop`: device collate, fused Adam, RCCL exchange) and takes from that object what a loop
        # can honour — see `_trainer_plan`; the factory is called at fit time (it may need a package this environment does not have)
        self._params = dict(
            n_blocks=n_blocks, n_heads=n_heads, n_factors=n_factors, use_pos_emb=use_pos_emb, use_causal_attn=use_causal_attn,
            use_key_padding_mask=use_key_padding_mask, dropout_rate=dropout_rate, session_max_len=session_max_len,
            dataloader_num_workers=dataloader_num_workers, batch_size=batch_size, loss=loss, n_negatives=n_negatives,
            gbce_t=gbce_t, lr=lr, epochs=epochs, verbose=verbose, deterministic=deterministic,
            recommend_batch_size=recommend_batch_size, recommend_torch_device=recommend_torch_device,
            train_min_user_interactions=train_min_user_interactions, get_val_mask_func=get_val_mask_func,
            get_val_mask_func_kwargs=get_val_mask_func_kwargs, data_preparator_kwargs=data_preparator_kwargs,
            transformer_layers_kwargs=transformer_layers_kwargs, pos_encoding_kwargs=pos_encoding_kwargs,
            lightning_module_kwargs=lightning_module_kwargs, similarity_module_kwargs=similarity_module_kwargs, seed=seed,
            data_preparator_type=data_preparator_type, transformer_layers_type=transformer_layers_type,
            similarity_module_type=similarity_module_type, item_net_block_types=tuple(item_net_block_types),
            item_net_constructor_type=item_net_constructor_type, item_net_constructor_kwargs=item_net_constructor_kwargs,
            negative_sampler_type=negative_sampler_type, negative_sampler_kwargs=negative_sampler_kwargs,
            pos_encoding_type=pos_encoding_type, lightning_module_type=lightning_module_type, backbone_type=backbone_type,
            backbone_kwargs=backbone_kwargs, get_trainer_func=get_trainer_func, get_trainer_func_kwargs=get_trainer_func_kwargs,
        )
        self._params.update(kwargs)
        for k, v in self._params.items():
            setattr(self, k, v)
        if self._requires_negatives() is None:
            raise ValueError(f"loss {loss} is not supported")
        if n_factors % n_heads != 0:
            raise ValueError("n_factors must be divisible by n_heads without remainder")   # nn.MultiheadAttention's own check
        self._dim_plan()     # fail at construction, with the reason, for what no padding makes the kernels tile (a head wider than 128)
        self.is_fitted = False
        self.dataset_schema: tp.Dict[str, tp.Any] = {}
        self.lightning_model: tp.Optional[hl.TransformerLossModule] = None
        self.optimizer: tp.Optional[hl.FlatAdam] = None
        self.epochs_done = 0
        self.history: tp.List[tp.Dict[str, float]] = []
        self._init_data_preparator()

    # ---- wiring -------------------------------------------------------------------------------------------
    def _kw(self, d: tp.Optional[dict]) -> dict:
        return dict(d) if d else {}

    def _init_negative_sampler(self) -> TransformerNegativeSamplerBase:
        """transformers/base.py:382-386.  The stock sampler is seeded from the model seed (and the rank, so that
        data-parallel replicas draw different negatives, as the reference's per-process generators do)."""
        kw = self._kw(self.negative_sampler_kwargs)
        if self.negative_sampler_type is CatalogUniformSampler and "seed" not in kw:
            kw["seed"] = (0 if self.seed is None else int(self.seed)) * 1000003 + _dist_info()[0]
        return self.negative_sampler_type(n_negatives=self.n_negatives, **kw)

    def _requires_negatives(self) -> tp.Optional[bool]:
        """`self.lightning_module_type.requires_negatives(self.loss)` (transformers/base.py:354): a plugged module decides."""
        fn = getattr(self.lightning_module_type, "requires_negatives", None)
        return hl.requires_negatives(self.loss) if fn is None else fn(self.loss)

    def _init_data_preparator(self, **extra: tp.Any) -> None:
        requires_negatives = bool(self._requires_negatives())
        self.data_preparator = self.data_preparator_type(
            session_max_len=self.session_max_len, batch_size=self.batch_size, dataloader_num_workers=self.dataloader_num_workers,
            train_min_user_interactions=self.train_min_user_interactions,
            n_negatives=self.n_negatives if requires_negatives else None,
            negative_sampler=self._init_negative_sampler() if requires_negatives else None,
            get_val_mask_func=self.get_val_mask_func, get_val_mask_func_kwargs=self.get_val_mask_func_kwargs,
            **extra, **self._kw(self.data_preparator_kwargs),
        )

    def _dim_plan(self) -> tp.Optional[hnn.DimPlan]:
        """How this model's sizes map onto sizes the kernels tile (`nn.DimPlan`): None when they do as they are (n_factors % 4 == 0, head
        size a multiple of 8, at most 128; HSTU: linear_hidden_dim == attention_dim).  Otherwise the stock modules are built padded
        with zero columns; a PLUGGED layer stack / item net / positional encoding / backbone is built by the caller's own class at the
        caller's own sizes and cannot be padded from outside: NotImplementedError names the constraint."""
        tkw = self._layer_kwargs()
        stu = self.transformer_layers_type is hnn.STULayers
        plan = hnn.DimPlan.make(self.n_factors, self.n_heads, "stu", tkw.get("linear_hidden_dim"), tkw.get("attention_dim")) if stu \
            else hnn.DimPlan.make(self.n_factors, self.n_heads, "mha")
        if plan is None:
            return None
        stock = (self.transformer_layers_type in (hnn.SASRecTransformerLayers, hnn.PreLNTransformerLayers, hnn.LiGRLayers, hnn.STULayers)
                 and self.pos_encoding_type is hnn.LearnableInversePositionalEncoding and self.backbone_type is hnn.TransformerTorchBackbone
                 and self.item_net_constructor_type is hnn.SumOfEmbeddingsConstructor
                 and all(t in (hnn.IdEmbeddingsItemNet, hnn.CatFeaturesItemNet) for t in self.item_net_block_types))
        if not stock:
            raise NotImplementedError(f"the HIP kernels need n_factors % 4 == 0 and a head size (n_factors / n_heads) that is a multiple of 8 and "
                                      f"at most 128; got n_factors={self.n_factors}, n_heads={self.n_heads}.  The stock modules run such sizes "
                                      f"padded with zero columns (nn.DimPlan); plugged module classes cannot be padded from outside")
        return plan

    def _layer_kwargs(self) -> tp.Dict[str, tp.Any]:
        """Keyword arguments of the layer stack beyond (n_blocks, n_factors, n_heads, dropout_rate): `transformer_layers_kwargs`, plus
        what a model kind derives from its own arguments (HSTU).  Derived values are added HERE, when the stack is built — never
        written into `transformer_layers_kwargs`: that dict is part of the config the reference reads back (hstu.py:660-674 passes the
        derived values itself, beside the kwargs)."""
        return self._kw(self.transformer_layers_kwargs)

    def _init_transformer_layers(self, plan: tp.Optional[hnn.DimPlan] = None) -> hnn.TransformerLayersBase:
        tkw = self._layer_kwargs()
        if plan is not None and plan.kind == "stu":
            tkw.update(linear_hidden_dim=plan.hd_pad, attention_dim=plan.hd_pad)
        return self.transformer_layers_type(n_blocks=self.n_blocks, n_factors=self.n_factors if plan is None else plan.d_pad,
                                            n_heads=self.n_heads, dropout_rate=self.dropout_rate, **tkw)

    def _init_similarity_module(self) -> hnn.DistanceSimilarityModule:
        kw = self._kw(self.similarity_module_kwargs)
        kw.setdefault("distance", self.u2i_dist_default)
        return self.similarity_module_type(**kw)

    def _init_item_model(self, item_net_schema: tp.Optional[tp.List[dict]], n_factors: tp.Optional[int] = None) -> hnn.SumOfEmbeddingsConstructor:
        """Item net from the processed train dataset (base.py:317-326), or — when restoring a checkpoint — from the shapes
        recorded with it (the role of `from_dataset_schema`, item_net.py:193-228: buffers are placeholders that the state
        dict overwrites).  n_factors: the width to build (a `DimPlan`'s padded width; default: the model's)."""
        kw = self._kw(self.item_net_constructor_kwargs)
        n_factors = self.n_factors if n_factors is None else n_factors
        if item_net_schema is None:
            return self.item_net_constructor_type.from_dataset(self.data_preparator.train_dataset, n_factors,
                                                               self.dropout_rate, self.item_net_block_types, **kw)
        n_tokens = self.data_preparator.item_id_map.size
        by_kind = {spec["kind"]: spec for spec in item_net_schema}
        blocks: tp.List[torch.nn.Module] = []
        # the configured block types in their configured ORDER decide the module list (`from_dataset_schema`,
        # item_net.py:413-460): an ids-only model fitted on a dataset WITH categorical features has no feature block, and
        # (Cat, Id) order puts the feature block at index 0 — the state dict's `item_net_blocks.N.*` keys follow that
        for block_type in self.item_net_block_types:
            if isinstance(block_type, type) and issubclass(block_type, hnn.IdEmbeddingsItemNet):
                blocks.append(block_type(n_factors, n_tokens, self.dropout_rate))
            elif isinstance(block_type, type) and issubclass(block_type, hnn.CatFeaturesItemNet):
                spec = by_kind.get("cat")
                if spec is None:
                    continue    # no categorical item features in the dataset: the block is skipped, as at fit time
                zeros = torch.zeros(n_tokens, dtype=torch.int64)
                blocks.append(block_type(torch.zeros(spec["nnz"], dtype=torch.int64), zeros, zeros.clone(),
                                         spec["n_cat_feature_values"], n_factors, self.dropout_rate))
            elif callable(getattr(block_type, "from_dataset_schema", None)):     # a plugged block class: the reference's own hook
                block = block_type.from_dataset_schema(_schema_view(self.dataset_schema), n_factors, self.dropout_rate)    # (item_net.py:44-52)
                if block is not None:
                    blocks.append(block)
            else:
                raise NotImplementedError(f"item net block {block_type!r} cannot be rebuilt from a dataset schema: it offers no "
                                          f"from_dataset_schema(dataset_schema, n_factors, dropout_rate)")
        return self.item_net_constructor_type(n_tokens, blocks, **kw)

    def _build_model_from_dataset(self, dataset: tp.Any, item_net_schema: tp.Optional[tp.List[dict]] = None) -> None:
        self._catalog_images = None     # a new model: images kept for recommend() belong to the old weights
        self._hand_prep_device()
        self.data_preparator.process_dataset_train(dataset)
        if self.seed is not None:  # before ANY parameter is created: 1-D parameters keep their constructor init
            torch.manual_seed(self.seed)
        plan = self._dim_plan()
        backbone = self._make_backbone(item_net_schema, None)
        if plan is not None:
            # sizes the kernels cannot tile (nn.DimPlan).  `backbone` above is the model at its REAL sizes, on the host: constructor
            # initialisation + xavier (lightning.py:296-299) happen there, exactly as for any model; the model that computes is built
            # with zero columns behind the real ones and takes the real one's state through the padding hooks
            hl.xavier_normal_init(backbone)
            real_state = backbone.state_dict()
            backbone = self._make_backbone(item_net_schema, plan)
            hnn.apply_dim_plan(backbone, plan)
            backbone.load_state_dict(real_state)
            del real_state
        dp = self.data_preparator
        self.lightning_model = self.lightning_module_type(
            torch_model=backbone, model_config=self.get_config(simple_types=True),
            dataset_schema=dp.train_dataset.get_schema() if dataset is not None else dict(self.dataset_schema),
            item_external_ids=dp.item_id_map.external_ids, item_extra_tokens=dp.item_extra_tokens, data_preparator=dp,
            lr=self.lr, gbce_t=self.gbce_t, loss=self.loss, verbose=self.verbose, train_loss_name=self.train_loss_name,
            val_loss_name=self.val_loss_name, adam_betas=(0.9, 0.98),
            n_negatives=self.n_negatives if self._requires_negatives() else None, n_item_extra_tokens=dp.n_item_extra_tokens,
            **self._kw(self.lightning_module_kwargs))
        device = torch.device(self._device())
        self.lightning_model.to(device)
        if plan is None:
            hl.xavier_normal_init(self.lightning_model.torch_model)  # on_train_start (lightning.py:296-299)
        self.optimizer = hl.FlatAdam(self.lightning_model.torch_model, lr=self.lr, betas=(0.9, 0.98))
        if os.environ.get("RT_DP_BACKEND", "torch") == "rccl" and _dist_info()[1] > 1 and device.type == "cuda":
            self.optimizer.use_rccl_exchange(*_dist_info())   # gradient exchange through rt_dp_* (include/rectools_hip.h)
        self.optimizer.broadcast_parameters()   # data parallel: replicas start from rank 0's weights (DDP semantics)
        self.epochs_done = 0
        self.history = []
        self._train_data_ready = dataset is not None
        self._train_dataset_ref = dataset
        # dropout streams: keyed by the model seed and the rank (replicas must not share masks), restarted with the model
        ops.RNG.seed = ((0 if self.seed is None else int(self.seed)) * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03 * _dist_info()[0]) \
            & 0xFFFFFFFFFFFFFFFF
        ops.RNG.step = 0
        if dataset is not None:   # kept for checkpoints (hyper_parameters.dataset_schema, base.py:470-473)
            self.dataset_schema = self.data_preparator.train_dataset.get_schema()

    def _hand_prep_device(self) -> None:
        """The preparator sorts the interaction columns where the model's parameters will live (`recommend_torch_device`, else
        LOCAL_RANK's GPU) — not on "the current device", which under torchrun is GPU 0 for every rank."""
        try:
            self.data_preparator.prep_device = self._device()
        except Exception:      # pylint: disable=broad-except   # no HIP device: the build below says so; processing itself runs anywhere
            self.data_preparator.prep_device = None

    def _make_backbone(self, item_net_schema: tp.Optional[tp.List[dict]], plan: tp.Optional[hnn.DimPlan]) -> hnn.TransformerTorchBackbone:
        """The torch model out of the classes the caller handed over — the plug-in seams of transformers/base.py:407-449 — at the
        model's own sizes (plan None) or at a `DimPlan`'s padded ones."""
        width = self.n_factors if plan is None else plan.d_pad
        item_model = self._init_item_model(item_net_schema, width)
        pkw = self._kw(self.pos_encoding_kwargs)
        pkw.setdefault("use_scale_factor", self.use_scale_factor_default)
        pos = self.pos_encoding_type(self.use_pos_emb, self.session_max_len, width, **pkw)
        return self.backbone_type(
            n_heads=self.n_heads, dropout_rate=self.dropout_rate, item_model=item_model, pos_encoding_layer=pos,
            transformer_layers=self._init_transformer_layers(plan), similarity_module=self._init_similarity_module(),
            use_causal_attn=self.use_causal_attn, use_key_padding_mask=self.use_key_padding_mask, **self._kw(self.backbone_kwargs))

    def _device(self) -> str:
        if self.recommend_torch_device is not None and str(self.recommend_torch_device) != "cpu":
            return str(self.recommend_torch_device)
        import os

        if torch.cuda.is_available():
            if "LOCAL_RANK" not in os.environ:
                return "cuda"
            # one process per GPU; more ranks than devices (the 2-ranks-on-1-GPU gloo test box) wrap around
            return f"cuda:{int(os.environ['LOCAL_RANK']) % torch.cuda.device_count()}"
        from . import _lib

        raise _lib.HipLibraryError("no HIP device visible: the MI355X engine cannot run (no CPU fallback)")

    @property
    def torch_model(self) -> hnn.TransformerTorchBackbone:
        if self.lightning_model is None:
            raise NotFittedError(type(self).__name__)
        return self.lightning_model.torch_model

    @property
    def require_recommend_context(self) -> bool:
        return False

    # ---- training -----------------------------------------------------------------------------------------
    def _to_device(self, batch: tp.Dict[str, np.ndarray], device: torch.device, validation: bool) -> tp.Dict[str, torch.Tensor]:
        out = {k: torch.from_numpy(v).to(device, non_blocking=True) for k, v in batch.items()}
        return self.data_preparator.add_negatives(out, validation=validation)

    def training_loop(self) -> "_TrainLoop":
        """The per-step machinery of fit(): sessions resident in HBM, device collate, negatives, fwd + bwd, fused Adam
        (+ gradient all-reduce).  `_run_epochs` drives it epoch by epoch; `bench.py` times exactly this object's `step()`."""
        return _TrainLoop(self)

    # what the engine's loop takes from a user-built Trainer, and what it calls on its callbacks (all duck-typed: no Lightning import)
    _TRAINER_HOOKS = ("on_fit_start", "on_train_start", "on_train_epoch_start", "on_train_epoch_end", "on_validation_start",
                      "on_validation_batch_end", "on_validation_epoch_end", "on_validation_end", "on_train_end", "on_fit_end")

    def _trainer_plan(self) -> tp.Optional[tp.Dict[str, tp.Any]]:
        """`get_trainer_func(**get_trainer_func_kwargs)` (transformers/base.py:367-380) read as a PLAN for the engine's own loop.

        Honoured: `max_epochs` / `min_epochs` (fit() trains max_epochs epochs, at least min_epochs once a callback sets
        `trainer.should_stop`), the logger's directory (`trainer.logger.log_dir` / `save_dir` -> the per-epoch CSV, the layout of
        Lightning's CSVLogger), `enable_progress_bar` (a line per epoch, as verbose > 0), `deterministic` (applied by the Trainer's own
        constructor; the engine's reductions run in a fixed order but for three: the sampled losses rank a candidate's pairs with atomics (the
        item table's gradient varies in its last bits from run to run), bias gradients' partial column sums meet in atomicAdds, and the HSTU
        relative-bias gradients — scripts/debug/grad_repro.py), and the hooks of
        USER-DEFINED callbacks — any object in `trainer.callbacks` whose class does not come from pytorch_lightning / lightning — called
        as `hook(trainer, lightning_model[, outputs, batch, batch_idx])` where the loop reaches the matching point.
        Not honoured, said once per fit in a warning: accelerator / devices / strategy / precision / gradient clipping / accumulation /
        limit_* / profilers, and Lightning's built-in callbacks (EarlyStopping, ModelCheckpoint, ...), which drive Lightning's own loop
        objects; `fit_trainer.save_checkpoint` is `model.save_to_checkpoint` here."""
        if self.get_trainer_func is None:
            return None
        trainer = self.get_trainer_func(**self._kw(self.get_trainer_func_kwargs))
        own, foreign = [], []
        for cb in list(getattr(trainer, "callbacks", None) or []):
            mod = type(cb).__module__ or ""
            (foreign if mod.split(".")[0] in ("pytorch_lightning", "lightning", "lightning_fabric") else own).append(cb)
        logger = getattr(trainer, "logger", None)
        log_dir = None
        if logger is not None:
            # `_log_epoch` makes its own `version_N` directory: it wants the directory Lightning's CSVLogger numbers its versions IN
            # (`save_dir/name`); a logger's `log_dir` already ends in `version_N` (metrics landed in version_N/version_M)
            save_dir, name, ldir = (getattr(logger, a, None) for a in ("save_dir", "name", "log_dir"))
            if isinstance(save_dir, str):
                log_dir = os.path.join(save_dir, name) if isinstance(name, str) and name else save_dir
            elif isinstance(ldir, str):
                base = os.path.basename(os.path.normpath(ldir))
                log_dir = os.path.dirname(os.path.normpath(ldir)) if base.startswith("version_") and base[8:].isdigit() else ldir
        plan = {"trainer": trainer, "max_epochs": getattr(trainer, "max_epochs", None), "min_epochs": getattr(trainer, "min_epochs", None),
                "callbacks": own, "log_dir": log_dir,
                "progress": bool(getattr(trainer, "enable_progress_bar", False) or getattr(trainer, "progress_bar_callback", None))}
        ignored = ", ".join(sorted({type(cb).__name__ for cb in foreign}))
        warnings.warn("get_trainer_func: the MI355X engine trains with its own loop (models._TrainLoop).  Taken from the Trainer: max_epochs="
                      f"{plan['max_epochs']}, min_epochs={plan['min_epochs']}, logger directory={log_dir!r}, {len(own)} user-defined callback(s)"
                      f" (hooks: {', '.join(self._TRAINER_HOOKS)}).  Everything else it configures (accelerator, devices, strategy, precision, "
                      "gradient clipping / accumulation, limit_*, profiler) is ignored"
                      + (f"; Lightning's own callbacks are not run: {ignored}" if ignored else ""))
        self.fit_trainer = trainer
        return plan

    @staticmethod
    def _call_hooks(plan: tp.Optional[tp.Dict[str, tp.Any]], hook: str, lm: tp.Any, *args: tp.Any) -> None:
        if plan is None:
            return
        for cb in plan["callbacks"]:
            fn = getattr(cb, hook, None)
            if callable(fn):
                fn(plan["trainer"], lm, *args)

    def _run_epochs(self, first: int, last: int, plan: tp.Optional[tp.Dict[str, tp.Any]] = None, min_last: tp.Optional[int] = None) -> None:
        """Epochs first .. last - 1.  plan: `_trainer_plan()` of a user-built Trainer; min_last: with it, the epoch count after which
        `trainer.should_stop` (set by a callback) ends training early."""
        lm, opt, dp = self.lightning_model, self.optimizer, self.data_preparator
        assert lm is not None and opt is not None
        device = next(lm.parameters()).device
        # the Trainer's directory is this run's, not a hyper-parameter: it stays out of `_params` (get_config / checkpoints)
        self._trainer_log_dir = plan["log_dir"] if plan is not None and plan["log_dir"] and not self._params.get("csv_log_dir") else None
        verbose = self.verbose or (plan is not None and plan["progress"])
        trainer = None if plan is None else plan["trainer"]
        self._call_hooks(plan, "on_fit_start", lm)
        self._call_hooks(plan, "on_train_start", lm)
        loop = self.training_loop()
        rank = loop.rank
        val_store = dp.val_store()
        self._catalog_images = None     # training moves the item embeddings
        ops.RNG.step = opt.step_count   # fit_partial / restored models continue the dropout streams where training stopped
        sampler = getattr(dp, "negative_sampler", None)
        if isinstance(sampler, CatalogUniformSampler) and sampler.calls < opt.step_count:
            sampler.calls = opt.step_count   # ... and the negative draws: a restored sampler must not replay batch 1, 2, ...
        for epoch in range(first, last):
            lm.train()
            loop.begin_epoch(epoch)
            if isinstance(getattr(lm, "__dict__", None), dict):
                lm.__dict__["logged_metrics"] = {}      # a metric logged in an earlier epoch is not this epoch's
            self._call_hooks(plan, "on_train_epoch_start", lm)
            total = torch.zeros((), device=device)
            n_batches = 0
            while loop.batches_left() > 0:
                total += loop.step().detach()
                n_batches += 1
            rec = {"epoch": epoch, self.train_loss_name: float(total) / max(n_batches, 1)}
            if val_store is not None:
                lm.eval()
                vt, vn = torch.zeros((), device=device), 0
                with torch.no_grad():
                    if plan is not None and plan["callbacks"]:      # lightning.py:329-333: the catalog matrix a validation callback reads
                        lm.item_embs = lm.torch_model.item_model.get_all_embeddings()
                    self._call_hooks(plan, "on_validation_start", lm)
                    for bi, b0 in enumerate(range(0, len(val_store), self.batch_size)):
                        vb = self._to_device(dp.collate_val(val_store, np.arange(b0, min(b0 + self.batch_size, len(val_store)))), device, True)
                        nb = int(vb["x"].shape[0])     # Lightning's epoch mean weights every batch by its size
                        # lightning.py:336-359: the outputs a callback's `on_validation_batch_end` receives carry the logits
                        # (the [B, V] product of the softmax loss only when a callback listens)
                        listens = plan is not None and any(callable(getattr(cb, "on_validation_batch_end", None)) for cb in plan["callbacks"])
                        step = getattr(lm, "validation_step", None)
                        if callable(step) and getattr(type(lm), "validation_step", None) is hl.TransformerLossModule.validation_step:
                            outputs = step(vb, bi, want_logits=listens)
                        elif callable(step):                 # a plugged module's own step: the reference's signature (lightning.py:336)
                            outputs = step(vb, bi)
                        else:
                            outputs = {"loss": lm.validation_loss(vb)}
                        vloss = outputs["loss"]
                        vt += vloss * nb
                        vn += nb
                        self._call_hooks(plan, "on_validation_batch_end", lm, outputs, vb, bi)
                    self._call_hooks(plan, "on_validation_epoch_end", lm)
                    self._call_hooks(plan, "on_validation_end", lm)
                    if hasattr(lm, "item_embs"):
                        del lm.item_embs
                rec[self.val_loss_name] = float(vt) / max(vn, 1)
            if loop.world > 1:
                # every rank's callbacks (early stopping, checkpoint-on-best) must see the SAME epoch metrics, or ranks decide differently
                # and meet in different collectives (Lightning: `self.log(..., sync_dist)` / `reduce_boolean_decision`)
                keys = [k for k in (self.train_loss_name, self.val_loss_name) if k in rec]
                vals = self._all_reduce_host([rec[k] for k in keys], "mean", device)
                rec.update(dict(zip(keys, vals)))
            self.history.append(rec)
            def publish() -> None:     # the metrics a callback's early-stopping logic reads (Lightning: trainer.callback_metrics)
                rec.update({k: v for k, v in getattr(lm, "logged_metrics", {}).items() if k not in rec})   # what callbacks logged this epoch
                if plan is None:
                    return
                try:
                    metrics = getattr(trainer, "callback_metrics", None)
                    if isinstance(metrics, dict):
                        metrics.update({k: torch.as_tensor(v) for k, v in rec.items() if k != "epoch"})
                except Exception:      # a real Lightning Trainer's connector: read-only      # pylint: disable=broad-except
                    pass

            publish()                  # validation callbacks have logged: `on_train_epoch_end` hooks read them
            self._call_hooks(plan, "on_train_epoch_end", lm)
            publish()                  # ... and what `on_train_epoch_end` logs belongs to THIS epoch's record too
            if rank == 0:
                self._log_epoch(rec, opt.step_count)
            if verbose and rank == 0:
                print(rec)
            self.epochs_done = epoch + 1
            stop = plan is not None and bool(getattr(trainer, "should_stop", False))
            if plan is not None and loop.world > 1:      # one rank asking to stop stops them all, in the same epoch
                stop = self._all_reduce_host([1.0 if stop else 0.0], "max", device)[0] > 0.0
                if stop:
                    try:
                        trainer.should_stop = True
                    except Exception:      # pylint: disable=broad-except
                        pass
            if stop and self.epochs_done >= (min_last or first):
                break
        # sharded data-parallel exchange: every rank holds only its slice of the Adam moments while training; gather them now, while
        # every rank is here and the process group is alive, so that saving / pickling afterwards is a local operation on any rank
        opt.consolidate_moments()
        self._call_hooks(plan, "on_train_end", lm)
        self._call_hooks(plan, "on_fit_end", lm)

    @staticmethod
    def _all_reduce_host(values: tp.Sequence[float], op: str, device: torch.device) -> tp.List[float]:
        """A few host scalars reduced over the process group ("mean" | "max"); the tensor lives where the backend can reduce it."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1 or not values:
            return [float(v) for v in values]
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
        if op == "mean":
            t /= dist.get_world_size()
        return [float(v) for v in t.cpu()]

    def _log_epoch(self, rec: tp.Dict[str, float], global_step: int) -> None:
        """Per-epoch metrics file in the layout of Lightning's CSVLogger (`<dir>/version_N/metrics.csv`, columns
        `epoch,step,train_loss,val_loss`; one row per logged metric group, validation first — what `self.log(..., on_epoch=
        True)` of lightning.py:320,358 produces).  Directory: `csv_log_dir=` model argument; the reference's default trainer
        logs only when verbose > 0 (base.py:375), to `lightning_logs/`."""
        import os

        log_dir = self._params.get("csv_log_dir") or getattr(self, "_trainer_log_dir", None) or ("lightning_logs" if self.verbose > 0 else None)
        if log_dir is None:
            return
        if getattr(self, "_csv_path", None) is None or rec["epoch"] == 0:
            os.makedirs(log_dir, exist_ok=True)
            taken = [int(n.split("_")[1]) for n in os.listdir(log_dir) if n.startswith("version_") and n.split("_")[1].isdigit()]
            vdir = os.path.join(log_dir, f"version_{max(taken) + 1 if taken else 0}")
            os.makedirs(vdir)
            self._csv_path = os.path.join(vdir, "metrics.csv")
            with open(self._csv_path, "w") as f:
                f.write(f"epoch,step,{self.train_loss_name},{self.val_loss_name}\n")
        with open(self._csv_path, "a") as f:
            step = global_step - 1
            if self.val_loss_name in rec:
                f.write(f"{rec['epoch']},{step},,{rec[self.val_loss_name]}\n")
            f.write(f"{rec['epoch']},{step},{rec[self.train_loss_name]},\n")

    @property
    def log_path(self) -> tp.Optional[str]:
        return getattr(self, "_csv_path", None)

    def fit(self, dataset: tp.Any) -> "TransformerModelBase":
        """Fit from scratch (models/base.py:326-341 -> transformers/base.py:481-489)."""
        self._build_model_from_dataset(dataset)
        plan = self._trainer_plan()
        if plan is None:
            self._run_epochs(0, self.epochs)
        else:     # the Trainer's epoch counts stand for the default trainer's max_epochs = min_epochs = self.epochs (base.py:369-371)
            n = int(plan["max_epochs"]) if plan["max_epochs"] is not None and int(plan["max_epochs"]) >= 0 else self.epochs
            self._run_epochs(0, n, plan, min_last=int(plan["min_epochs"] or 0))
        self.is_fitted = True
        return self

    def fit_partial(self, dataset: tp.Any, min_epochs: int = 1, max_epochs: int = 1) -> "TransformerModelBase":
        """Continue training for `max_epochs` more epochs (transformers/base.py:505-533)."""
        if not self.is_fitted:
            self._build_model_from_dataset(dataset)
        else:
            # the reference re-processes the dataset it is handed on EVERY call (transformers/base.py:515-520: "assumed that dataset
            # is same as in `fit`") — new interactions of known items are picked up; a restored model (checkpoint / pickle) has to
            # rebuild its training sessions anyway (base.py:520-523).  The dataset must index the SAME embedding rows the weights
            # were trained on: checked on a copy of the preparator's state, committed only when it holds.
            self._reprocess_train_dataset(dataset)
        plan = self._trainer_plan()       # (epoch counts: the call's own arguments, transformers/base.py:527-529)
        self._run_epochs(self.epochs_done, self.epochs_done + max_epochs, plan, min_last=self.epochs_done + min_epochs)
        self.is_fitted = True
        return self

    def _reprocess_train_dataset(self, dataset: tp.Any) -> None:
        """`data_preparator.process_dataset_train(dataset)` for a model that already holds weights: validate before mutating — if
        the dataset maps items to other embedding rows the preparator is left exactly as it was and ValueError is raised (a caller
        that catches it must not end up with trained weights paired with a different item map)."""
        dp = self.data_preparator
        same = getattr(self, "_train_data_ready", False) and getattr(self, "_train_dataset_ref", None) is dataset
        if same:
            return      # the very Dataset object the sessions were cut from (Datasets are immutable): nothing to redo
        self._hand_prep_device()
        keep = ("item_id_map", "train_dataset", "extra_token_ids", "val_interactions", "_train_store")
        snapshot = {k: getattr(dp, k) for k in keep if hasattr(dp, k)}
        known = np.asarray(dp.item_id_map.external_ids)
        try:
            dp.process_dataset_train(dataset)
            now = np.asarray(dp.item_id_map.external_ids)
            if len(known) != len(now) or not (known == now).all():
                raise ValueError("fit_partial: the dataset maps items to other embedding rows than the model was trained with "
                                 "(different items, or a different order of first appearance)")
        except Exception:
            for k in keep:
                if k in snapshot:
                    setattr(dp, k, snapshot[k])
                elif hasattr(dp, k):
                    delattr(dp, k)
            raise
        self._train_data_ready = True
        self._train_dataset_ref = dataset

    # ---- inference ----------------------------------------------------------------------------------------
    # ---- the catalog's coarse-pass images are kept between recommend() calls -----------------------------------------
    def _catalog_key(self, distance: tp.Any, item_embs: torch.Tensor) -> tp.Tuple:
        """Identifies the CONTENT of the item embeddings without reading them: torch-side writes bump a parameter's `_version`
        (load_state_dict, manual edits), this engine's own writes go through the optimiser (`step_count`); a new model / a new
        flat buffer changes the pointers."""
        lm = self.lightning_model
        params = tuple((p.data_ptr(), p._version, tuple(p.shape)) for p in lm.torch_model.item_model.parameters())   # pylint: disable=protected-access
        # ... and a cheap look AT them for the writes neither counter sees (`param.data.copy_()`, a raw pointer write): the sum
        # of ~1,024 evenly spaced rows, one small reduction + one scalar read per recommend() call
        stash = getattr(self, "_item_probe", None)       # taken by `_item_embeddings()`, before the encoder was queued (no mid-call sync)
        probe = stash[1] if stash is not None and stash[0] == (item_embs.data_ptr(), tuple(item_embs.shape)) else self._probe(item_embs)
        return (str(distance), params, 0 if self.optimizer is None else self.optimizer.step_count, tuple(item_embs.shape), str(item_embs.device),
                probe)

    @staticmethod
    def _probe(item_embs: torch.Tensor) -> float:
        V = int(item_embs.shape[0])
        return float(item_embs[:: max(1, V // 1024)].sum(dtype=torch.float64)) if V else 0.0

    def invalidate_catalog_images(self) -> None:
        """Drop the catalog's coarse-pass images kept between recommend() calls (hm + one-plane + fragment forms: up to ~8 d bytes per
        item, ~20 GB at 5 M x 512).  Call after writing item embeddings behind torch's back; they are also never kept when larger than
        RT_CATALOG_IMAGES_MAX_GB (default 64) — every call then rebuilds them (2.6 ms at 5 M x 512)."""
        self._catalog_images = None

    def _ranker(self, distance: tp.Any, device: tp.Any, user_embs: torch.Tensor, item_embs: torch.Tensor) -> HipRanker:
        """A ranker for this call's user factors that inherits the catalog images (and the largest item norm) an earlier call built,
        as long as the item embeddings are still the same (a fresh HipRanker per call re-read the whole catalog and
        synchronised once more)."""
        ranker = HipRanker(distance, device, user_embs, item_embs)
        key = self._catalog_key(distance, item_embs)
        kept = getattr(self, "_catalog_images", None)
        if kept is not None and kept[0] == key:
            ranker.adopt_images(kept[1])
        ranker._catalog_key = key    # pylint: disable=protected-access
        return ranker

    def _keep_images(self, ranker: HipRanker) -> None:
        images = ranker.export_images()
        total = sum(int(t.numel()) * t.element_size() for t in self._tensors_of(images))
        if total > float(os.environ.get("RT_CATALOG_IMAGES_MAX_GB", "64")) * 2 ** 30:
            self._catalog_images = None
            return
        self._catalog_images = (ranker._catalog_key, images)    # pylint: disable=protected-access

    @staticmethod
    def _tensors_of(obj: tp.Any) -> tp.List[torch.Tensor]:
        if isinstance(obj, torch.Tensor):
            return [obj]
        if isinstance(obj, dict):
            obj = list(obj.values())
        if isinstance(obj, (list, tuple)):
            return [t for o in obj for t in TransformerModelBase._tensors_of(o)]
        return []

    def _item_embeddings(self) -> torch.Tensor:
        """Catalog matrix in eval mode, produced once per recommend call (lightning.py:386-389)."""
        lm = self.lightning_model
        assert lm is not None
        lm.eval()
        with torch.no_grad():
            item_embs = lm.torch_model.item_model.get_all_embeddings().detach()
        self._item_probe = ((item_embs.data_ptr(), tuple(item_embs.shape)), self._probe(item_embs))
        return item_embs

    def _user_embeddings(self, store: SequenceStore, device: torch.device, item_embs: torch.Tensor) -> torch.Tensor:
        """Last-slot encodings of every session, eval mode, kept on the device (lightning.py:378-400)."""
        lm = self.lightning_model
        assert lm is not None
        lm.eval()
        outs = []
        dstore = DeviceSequenceStore(store, device)
        all_idx = torch.arange(len(store), dtype=torch.int64, device=device)
        with torch.no_grad():
            bs = self._encode_batch_size()
            for b0 in range(0, len(store), bs):
                batch = self.data_preparator.collate_recommend_device(dstore, all_idx[b0:b0 + bs])
                outs.append(lm.torch_model.encode_last(batch, item_embs))
        return torch.cat(outs) if outs else torch.zeros((0, self.n_factors), device=device)

    @staticmethod
    def _build_session_index(u_t: torch.Tensor, i_t: torch.Tensor, t_t: torch.Tensor, w_t: torch.Tensor, lookup_t: torch.Tensor,
                             n_users: int, V: int) -> tp.Tuple[torch.Tensor, ...]:
        """Session store + viewed-items CSR of EVERY user of a Dataset, on the device the columns live on (plain tensor ops).

        Rows = the Dataset's internal user ids.  Items are model ids (`lookup_t`: dataset item id -> model item id, -1 = unknown
        to the model, dropped); a user's interactions keep the reference's session order — stable sort by time, grouped by
        user (data_preparator.py:73-99); the CSR holds each user's DISTINCT items ascending (dataset.py:314-348)."""
        mitem = lookup_t[i_t]
        keep = mitem >= 0
        u, m, t, w = u_t[keep], mitem[keep], t_t[keep], w_t[keep]
        o1 = torch.sort(t, stable=True).indices
        o2 = torch.sort(u[o1], stable=True).indices
        order = o1[o2]
        u_s, item_s, w_s, t_s = u[order], m[order], w[order], t[order]
        offsets = torch.zeros((n_users + 1,), dtype=torch.int64, device=u_t.device)
        torch.cumsum(torch.bincount(u_s, minlength=n_users), 0, out=offsets[1:])
        key = torch.unique(u_s * V + item_s)                         # sorted: (user, item) pairs, distinct
        frow = torch.div(key, V, rounding_mode="floor")
        indptr = torch.zeros((n_users + 1,), dtype=torch.int64, device=u_t.device)
        torch.cumsum(torch.bincount(frow, minlength=n_users), 0, out=indptr[1:])
        indices = (key - frow * V).to(torch.int32)
        return offsets, item_s, w_s, indptr, indices, t_s

    @staticmethod
    def _select_csr_rows(indptr: torch.Tensor, indices: torch.Tensor, rows: torch.Tensor,
                         total: tp.Optional[int] = None) -> tp.Tuple[torch.Tensor, torch.Tensor]:
        """CSR of the given rows, in the given order (indices of a row stay ascending).  `total`: the entry count when the caller
        knows it (from host offsets) — saves the device -> host round trip that sizes the result."""
        lens = indptr[rows + 1] - indptr[rows]
        new_indptr = torch.zeros((rows.numel() + 1,), dtype=torch.int64, device=indptr.device)
        torch.cumsum(lens, 0, out=new_indptr[1:])
        if total is None:
            total = int(new_indptr[-1])
        src = torch.repeat_interleave(indptr[rows] - new_indptr[:-1], lens, output_size=total) \
            + torch.arange(total, dtype=torch.int64, device=indptr.device)
        return new_indptr, indices[src]

    def _device_session_index(self, dataset: tp.Any, device: torch.device) -> tp.Tuple[torch.Tensor, ...]:
        """`_build_session_index` of `dataset` for this model's item map, resident in HBM.  At ML-20M scale (19.8 M rows) the
        per-call upload (555 MB over PCIe), the float64 -> float32 pass and the filter / sort / unique over the whole table were
        most of what a 16,384-user recommend() spent outside the encoder; neither a Dataset nor a fitted model's item map is
        mutated after construction, so the index is built by the first call and kept on the Dataset's `interactions` object
        (per device and item map; the objects the key names are referenced by the cache entry, so an id cannot be reused
        while the entry lives).  The raw columns are only temporaries of the build."""
        inter, dp = dataset.interactions, self.data_preparator
        df = inter.df
        key = (str(device), id(df), len(df), id(dp.item_id_map), id(dataset.item_id_map), id(dataset.user_id_map))
        cache = getattr(inter, "_rt_session_index", None)
        if cache is not None and cache[0] == key:
            return cache[1]
        to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device, non_blocking=True)  # noqa: E731
        # dataset item id -> model item id (-1: unknown to the model)
        lookup = pd.Series(dp.item_id_map.to_internal).reindex(dataset.item_id_map.external_ids).fillna(-1).values.astype(np.int64)
        index = self._build_session_index(
            to_dev(df[Columns.User].values.astype(np.int64, copy=False)), to_dev(df[Columns.Item].values.astype(np.int64, copy=False)),
            to_dev(df[Columns.Datetime].values.astype("datetime64[ns]").view(np.int64)),
            to_dev(df[Columns.Weight].values.astype(np.float32, copy=False)), to_dev(lookup),
            dataset.user_id_map.size, dp.item_id_map.size)
        # host copies of the two offset arrays (n_users + 1 int64 each, one D2H per Dataset): a request computes its row counts —
        # encoder rows, filter entries — from them, so no step of a recommend() call waits on the device to size a buffer
        offsets_d, t_ns = index[0], index[5]
        ts_s = last_ns_h = None
        if dp.add_unix_ts:
            # the sessions' timestamps in seconds, by the arithmetic of `_to_unix_ts` (float64 division, truncation), and every user's
            # last interaction time (ns, host): a request's context must not lie before it (it is the LAST row of the session then)
            ts_s = (t_ns.double() / 10**9).to(torch.int64)
            last = torch.full((dataset.user_id_map.size,), torch.iinfo(torch.int64).min, dtype=torch.int64, device=device)
            has = offsets_d[1:] > offsets_d[:-1]
            last[has] = t_ns[offsets_d[1:][has] - 1]
            last_ns_h = last.cpu().numpy()
        index = tuple(index[:5]) + (index[0].cpu().numpy(), index[3].cpu().numpy(), ts_s, last_ns_h)
        try:
            inter._rt_session_index = (key, index, (df, dp.item_id_map, dataset.item_id_map, dataset.user_id_map))   # pylint: disable=protected-access
        except AttributeError:   # a duck-typed Interactions object with __slots__: no cache
            pass
        return index

    def _encode_batch_size(self) -> int:
        """Sessions per encoder launch in recommend().  `recommend_batch_size` (reference default 256) is a memory knob of the
        reference's DataLoader; every row of the encoder is independent of the batch it travels in, so the engine groups up to
        8,192 sessions per launch (≈ 0.9 M packed rows, ≈ 12 GB of scratch at d = 256; round 6, all 138,493 users of C2: encoder 111.0 /
        110.5 / 104.5 / 105.8 ms at 2,048 / 4,096 / 8,192 / 16,384 sessions per launch — with the final block on one row per session its
        chain of small launches is paid per encoder launch; RT_ENCODE_SESSIONS overrides)."""
        env = os.environ.get("RT_ENCODE_SESSIONS")
        if env:
            return max(int(env), 1)
        if int(self.recommend_batch_size) != 256:      # set by the caller (the reference's default is 256): its memory knob stands
            return int(self.recommend_batch_size)
        # the default: up to 8,192 sessions, fewer when the launch's scratch (~24 row-sized fp32 buffers of window x n_factors) would
        # take more than a fifth of the free HBM (wide / long-window models, shared GPUs: ADVICE r4); a launch that still runs out of
        # memory is halved and repeated (`_recommend_device_glue`)
        sessions = 8192
        if torch.cuda.is_available() and self.lightning_model is not None:
            dev = next(self.lightning_model.parameters()).device
            if dev.type == "cuda":
                free, _ = torch.cuda.mem_get_info(dev)
                per_session = 24 * 4 * int(self.session_max_len) * int(self.n_factors)
                sessions = int(min(8192, max(256, free // 5 // max(per_session, 1))))
        return max(int(self.recommend_batch_size), sessions)

    def _check(self, k: int) -> None:
        if not self.is_fitted:
            raise NotFittedError(type(self).__name__)
        if k <= 0:
            raise ValueError("`k` must be positive integer")

    def _whitelist(self, items_to_recommend: tp.Optional[tp.Any]) -> np.ndarray:
        dp = self.data_preparator
        if items_to_recommend is None:
            return dp.get_known_items_sorted_internal_ids()
        internal = dp.item_id_map.convert_to_internal(items_to_recommend, strict=False)
        internal = internal[internal >= dp.n_item_extra_tokens]
        return np.unique(internal)

    def recommend(self, users: tp.Any, dataset: tp.Any, k: int, filter_viewed: bool, items_to_recommend: tp.Optional[tp.Any] = None,
                  add_rank_col: bool = True, on_unsupported_targets: str = "raise",
                  context: tp.Optional[pd.DataFrame] = None) -> pd.DataFrame:
        """Top-k items for `users` (models/base.py:385-519)."""
        self._check(k)
        if on_unsupported_targets not in ("raise", "warn", "ignore"):
            raise ValueError("`on_unsupported_targets` must be one of 'raise', 'warn', 'ignore'")
        if self.require_recommend_context and context is None:
            raise ValueError("This model requires `context` to be provided for recommendations generation")
        if not self.require_recommend_context and context is not None:
            context = None
        users = np.asarray(users)
        known = pd.Series(users).isin(dataset.user_id_map.external_ids).values
        if not known.all():
            if on_unsupported_targets == "raise":
                raise ValueError("Model doesn't support recommendations for cold users, but some of given users are cold")
            if on_unsupported_targets == "warn":
                warnings.warn("Model doesn't support recommendations for cold users, but some of given users are cold")
            users = users[known]
        dp_ = self.data_preparator
        if not hnn.similarity_is_stock(self.torch_model.similarity_module):
            return self._recommend_via_similarity(users, dataset, k, filter_viewed, items_to_recommend, add_rank_col,
                                                  on_unsupported_targets, context)
        plain = context is None and not dp_.add_unix_ts
        # a model that reads timestamps (HSTU) takes the device path when its stack packs: the request's context time becomes the last
        # timestamp of every packed session (`rt_collate_packed_ts`)
        timed = context is not None and dp_.add_unix_ts and type(dp_).__name__ == "SASRecDataPreparator" and \
            isinstance(self.torch_model.transformer_layers, hnn.STULayers) and os.environ.get("RT_PACKED", "1") != "0"
        if (plain or timed) and not (dp_.extra_cols or []):
            fast = self._recommend_device_glue(users, dataset, k, filter_viewed, items_to_recommend, add_rank_col,
                                               on_unsupported_targets, context if timed else None)
            if fast is not None:
                return fast
        with warnings.catch_warnings():
            if on_unsupported_targets == "ignore":
                warnings.simplefilter("ignore")
            rec_ds = self.data_preparator.transform_dataset_u2i(dataset, users, context)
        user_ids = rec_ds.user_id_map.convert_to_internal(users, strict=False)
        whitelist = self._whitelist(items_to_recommend)
        device = next(self.lightning_model.parameters()).device
        store = SequenceStore.from_interactions(rec_ds.interactions.df, sort_users=True)  # session i <-> internal user i
        if len(user_ids) == 0 or len(whitelist) == 0:
            return self._frame(np.array([], users.dtype), np.array([], object), np.array([], np.float32), add_rank_col, Columns.User)
        item_embs = self._item_embeddings()
        user_embs = self._user_embeddings(store, device, item_embs)
        ranker = self._ranker(self.lightning_model.torch_model.similarity_module.distance, device, user_embs, item_embs)
        filt = None
        if filter_viewed:
            filt = DeviceCSR.from_scipy(rec_ds.get_user_item_matrix(include_weights=False)[user_ids], device)
        ids, scores, counts, _ = ranker.rank_device(user_ids, k=k, filter_pairs_csr=filt, sorted_object_whitelist=whitelist)
        self._keep_images(ranker)
        return self._assemble(rec_ds.user_id_map.convert_to_external(user_ids), ids, scores, counts, add_rank_col, Columns.User)

    def _recommend_via_similarity(self, users: np.ndarray, dataset: tp.Any, k: int, filter_viewed: bool,
                                  items_to_recommend: tp.Optional[tp.Any], add_rank_col: bool, on_unsupported_targets: str,
                                  context: tp.Optional[pd.DataFrame]) -> pd.DataFrame:
        """recommend() of a model whose `similarity_module_type` is not the stock one: the module's own towers and its own
        `_recommend_u2i` are called with the reference's arguments (lightning.py:378-428: user embeddings through
        `session_tower_forward`, the catalog through `item_tower_forward`, then `similarity_module._recommend_u2i(user_embs, item_embs,
        user_ids, k, sorted_item_ids_to_recommend, ui_csr_for_filter)`), its triplet becomes the frame (models/base.py:502-519)."""
        with warnings.catch_warnings():
            if on_unsupported_targets == "ignore":
                warnings.simplefilter("ignore")
            rec_ds = self.data_preparator.transform_dataset_u2i(dataset, users, context)
        user_ids = rec_ds.user_id_map.convert_to_internal(users, strict=False)
        whitelist = self._whitelist(items_to_recommend)
        if len(user_ids) == 0 or len(whitelist) == 0:
            return self._frame(np.array([], users.dtype), np.array([], object), np.array([], np.float32), add_rank_col, Columns.User)
        lm = self.lightning_model
        device = next(lm.parameters()).device
        sim = lm.torch_model.similarity_module
        store = SequenceStore.from_interactions(rec_ds.interactions.df, sort_users=True)  # session i <-> internal user i
        item_embs = self._item_embeddings()
        user_embs = self._user_embeddings(store, device, item_embs)
        with torch.no_grad():
            user_embs = sim.session_tower_forward(user_embs)
            item_embs = sim.item_tower_forward(item_embs)
            ui_csr = rec_ds.get_user_item_matrix(include_weights=False)[user_ids] if filter_viewed else None
            uids, reco, scores = sim._recommend_u2i(          # pylint: disable=protected-access
                user_embs=user_embs, item_embs=item_embs, user_ids=user_ids, k=k, sorted_item_ids_to_recommend=whitelist,
                ui_csr_for_filter=ui_csr)
        ext_u = rec_ds.user_id_map.convert_to_external(np.asarray(uids))
        ext_i = self.data_preparator.item_id_map.convert_to_external(np.asarray(reco).astype(np.int64))
        return self._frame(ext_u, ext_i, np.asarray(scores, dtype=np.float32), add_rank_col, Columns.User)

    def recommend_distributed(self, users: tp.Any, dataset: tp.Any, k: int, filter_viewed: bool, **kwargs: tp.Any) -> pd.DataFrame:
        """recommend() over all ranks of an initialised process group: users are independent, every rank
        holds the whole catalog, so rank r ranks the r-th contiguous slice of `users` on its own GPU and the frames are
        gathered (one `all_gather_object` of the result frames — no collective on the data path).  Every rank returns the
        full frame, rows in the order a single-process `recommend(users, ...)` produces.  The reference's recommend() is
        single-device (lightning.py:371-376)."""
        import torch.distributed as dist

        rank, world = _dist_info()
        users = np.asarray(users)
        if world == 1:
            return self.recommend(users, dataset, k, filter_viewed, **kwargs)
        bounds = np.linspace(0, len(users), world + 1).astype(np.int64)
        part = self.recommend(users[bounds[rank]:bounds[rank + 1]], dataset, k, filter_viewed, **kwargs)
        parts: tp.List[tp.Optional[pd.DataFrame]] = [None] * world
        dist.all_gather_object(parts, part)
        return pd.concat([p for p in parts if p is not None and len(p)], ignore_index=True) if any(
            p is not None and len(p) for p in parts) else part

    def _recommend_device_glue(self, users: np.ndarray, dataset: tp.Any, k: int, filter_viewed: bool,
                               items_to_recommend: tp.Optional[tp.Any], add_rank_col: bool,
                               on_unsupported_targets: str, context: tp.Optional[pd.DataFrame] = None) -> tp.Optional[pd.DataFrame]:
        """recommend() without the pandas / scipy round trips of the reference's glue (`models/base.py:
        502-519,735-791`, `data_preparator.py:354-424`, `dataset.py:314-348`): the Dataset's time-ordered session store and
        viewed-items CSR (known items only) are built ONCE on the device (`_device_session_index`); a call selects the requested
        users' rows from them, encodes and ranks on the device.  Same rows, order and values as the reference-shaped path
        below (tests/test_models_gpu.py compares the two).  Returns None when the request needs that path (duplicated users)."""
        dp, lm = self.data_preparator, self.lightning_model
        assert lm is not None
        device = next(lm.parameters()).device
        import time as _time

        plog = getattr(self, "phase_log", None)     # bench.py: {} -> seconds per phase of THIS call (device-synchronised ticks)
        t_last = [_time.perf_counter()]

        def tick(name: str) -> None:
            if plog is not None:
                torch.cuda.synchronize(device)
                now = _time.perf_counter()
                plog[name] = plog.get(name, 0.0) + now - t_last[0]
                t_last[0] = now

        req = dataset.user_id_map.convert_to_internal(users, strict=False)
        if len(np.unique(req)) != len(req):
            return None
        whitelist = self._whitelist(items_to_recommend)
        empty = self._frame(np.array([], users.dtype), np.array([], object), np.array([], np.float32), add_rank_col, Columns.User)
        if len(req) == 0 or len(whitelist) == 0:
            return empty
        n_req, V = len(req), dp.item_id_map.size
        offsets, item_s, w_s, f_indptr, f_indices, off_h, fptr_h, ts_s, last_ns_h = self._device_session_index(dataset, device)
        req = np.ascontiguousarray(req.astype(np.int64))
        lens_h = off_h[req + 1] - off_h[req]
        valid_h = lens_h > 0                                        # users with at least one item the model knows
        rows_h = req[valid_h]                                       # rows of the session index, request order
        n_valid = int(len(rows_h))
        n_cold = n_req - n_valid
        if n_cold > 0 and on_unsupported_targets != "ignore":
            warnings.warn(f"{n_cold} target users were considered cold because of missing known items")
        if n_valid == 0:
            return empty
        ctx_d = None
        if context is not None:      # data_preparator.transform_dataset_u2i's checks, then the request times of the valid users
            if not pd.Series(np.asarray(users)).isin(context[Columns.User].unique()).all():
                raise ValueError("No context for some target users")
            if context.duplicated(subset=Columns.User).any():
                raise ValueError("Duplicated user entries found in context. Each user must have exactly one context row.")
            when = context.set_index(Columns.User)[Columns.Datetime].reindex(np.asarray(users)[valid_h])
            ctx_ns = pd.to_datetime(when).values.astype("datetime64[ns]").astype("int64")
            if ts_s is None or last_ns_h is None or bool((ctx_ns < last_ns_h[rows_h]).any()):
                return None      # a context that is not the session's last row: the reference-shaped path sorts it where it falls
            ctx_d = torch.from_numpy(dp._to_unix_ts(when)).to(device, non_blocking=True)    # pylint: disable=protected-access
        valid_rows = torch.from_numpy(rows_h).to(device, non_blocking=True)
        dstore = DeviceSequenceStore.from_device(offsets, item_s, w_s, None)
        item_embs = self._item_embeddings()
        tick("glue")
        unsort = None

        def encode(bs: int) -> tp.Optional[tp.List[torch.Tensor]]:
            outs: tp.List[torch.Tensor] = []
            # packed encoder (no padding rows: 45 % of the [B, L] window at ML-20M scale) where the stack offers it; RT_PACKED=0
            # keeps the padded window.  Same encodings up to fp32 rounding (tests/test_packed_gpu.py).
            packed = os.environ.get("RT_PACKED", "1") != "0" and type(dp).__name__ in _PACKED_PREPARATORS \
                and (not dp.add_unix_ts or ctx_d is not None) and lm.torch_model.can_encode_packed(item_embs.shape[1], dp.session_max_len)
            if ctx_d is not None and not packed:
                return None
            mask_id = dp.extra_token_ids[MASKING_VALUE] if type(dp).__name__ == "BERT4RecDataPreparator" else None
            if packed:   # packed row offsets of every encoder launch, cut on the host, one upload
                L = dp.session_max_len
                n_launch = -(-n_valid // bs)
                lens_v = np.minimum(lens_h[valid_h], L) if mask_id is None else np.minimum(lens_h[valid_h], L - 1) + 1
                enc_rows = valid_rows     # request order: an encoder launch holds >= 1024 sessions x heads = 16 workgroup rounds — sorting them
                #                           by length (as the 128-session training batches are) bought nothing and cost a host argsort
                grid = np.zeros((n_launch, bs), dtype=np.int64)
                grid.reshape(-1)[:n_valid] = lens_v
                cu_h = np.zeros((n_launch, bs + 1), dtype=np.int64)
                np.cumsum(grid, axis=1, out=cu_h[:, 1:])
                cu_d = torch.from_numpy(cu_h).to(device, non_blocking=True)
            for bi, b0 in enumerate(range(0, n_valid, bs)):
                nb = min(bs, n_valid - b0)
                if packed:
                    kw = {} if ctx_d is None else {"ts_store": ts_s, "ts_ctx": ctx_d[b0:b0 + nb]}
                    outs.append(lm.torch_model.encode_last_packed(offsets, item_s, enc_rows[b0:b0 + nb], dp.session_max_len, item_embs,
                                                                  cu=cu_d[bi, :nb + 1], n_rows=int(cu_h[bi, nb]), mask_id=mask_id,
                                                                  cache=call_cache, **kw))
                    continue
                batch = dp.collate_recommend_device(dstore, valid_rows[b0:b0 + nb])
                outs.append(lm.torch_model.encode_last(batch, item_embs))   # last-position encodings, [b, d]
            return outs

        bs = self._encode_batch_size()
        call_cache: tp.Dict[str, tp.Any] = {}      # what is valid for THIS call's fixed weights (nn.TransformerTorchBackbone.encode_last_packed)
        with torch.no_grad():
            while True:      # a launch that does not fit beside what else lives on this GPU is halved and repeated
                try:
                    outs = encode(bs)
                    break
                except torch.cuda.OutOfMemoryError:
                    if bs <= 64:
                        raise
                    bs //= 2
                    torch.cuda.empty_cache()
        if outs is None:
            return None
        user_embs = torch.cat(outs)
        if unsort is not None:
            user_embs = user_embs.index_select(0, unsort)
        tick("encoder")
        ranker = self._ranker(lm.torch_model.similarity_module.distance, device, user_embs, item_embs)
        filt = None
        if filter_viewed:  # CSR of the distinct (user, item) pairs, rows in request order, indices ascending
            flens = fptr_h[rows_h + 1] - fptr_h[rows_h]
            indptr, indices = self._select_csr_rows(f_indptr, f_indices, valid_rows, total=int(flens.sum()))
            if indices.numel() == 0:
                indices = torch.zeros((1,), dtype=torch.int32, device=device)
            filt = DeviceCSR(indptr, indices, (n_valid, V))
        ids, scores, cnt, _ = ranker.rank_device(np.arange(n_valid), k=k, filter_pairs_csr=filt, sorted_object_whitelist=whitelist)
        self._keep_images(ranker)
        tick("ranker")
        ext_users = np.asarray(users)[valid_h]
        frame = self._assemble(ext_users, ids, scores, cnt, add_rank_col, Columns.User)
        tick("frame")
        return frame

    def recommend_to_items(self, target_items: tp.Any, dataset: tp.Any, k: int, filter_itself: bool = True,
                           items_to_recommend: tp.Optional[tp.Any] = None, add_rank_col: bool = True,
                           on_unsupported_targets: str = "raise") -> pd.DataFrame:
        """Item-to-item by cosine over the item table (lightning.py:428-449, models/base.py:521-646)."""
        self._check(k)
        dp = self.data_preparator
        target_items = np.asarray(target_items)
        known = pd.Series(target_items).isin(dp.get_known_item_ids()).values
        if not known.all():
            if on_unsupported_targets == "raise":
                raise ValueError("Model doesn't support recommendations for cold items, but some of given items are cold")
            if on_unsupported_targets == "warn":
                warnings.warn("Model doesn't support recommendations for cold items, but some of given items are cold")
            target_items = target_items[known]
        target_ids = dp.item_id_map.convert_to_internal(target_items)
        whitelist = self._whitelist(items_to_recommend)
        device = next(self.lightning_model.parameters()).device
        item_embs = self._item_embeddings()
        ranker = self._ranker(Distance.COSINE, device, item_embs, item_embs)
        kk = k + 1 if filter_itself else k
        ids, scores, counts, _ = ranker.rank_device(target_ids, k=kk, sorted_object_whitelist=whitelist)
        self._keep_images(ranker)
        ids, scores, counts = ids.cpu().numpy(), scores.cpu().numpy(), counts.cpu().numpy()
        t_out, i_out, s_out = [], [], []
        for r, t in enumerate(target_ids):
            row_ids, row_sc = ids[r, :counts[r]], scores[r, :counts[r]]
            if filter_itself:
                keep = row_ids != t
                row_ids, row_sc = row_ids[keep][:k], row_sc[keep][:k]
            t_out.append(np.repeat(target_items[r], len(row_ids))); i_out.append(row_ids); s_out.append(row_sc)
        tt = np.concatenate(t_out) if t_out else np.array([], target_items.dtype)
        ii = dp.item_id_map.convert_to_external(np.concatenate(i_out).astype(np.int64)) if i_out else np.array([], object)
        return self._frame(tt, ii, np.concatenate(s_out) if s_out else np.array([], np.float32), add_rank_col, Columns.TargetItem)

    def _assemble(self, ext_users: np.ndarray, ids: torch.Tensor, scores: torch.Tensor, counts: torch.Tensor, add_rank_col: bool,
                  target_col: str) -> pd.DataFrame:
        return self._assemble_frame(ext_users, ids.cpu().numpy(), scores.cpu().numpy(), counts.cpu().numpy(),
                                    self.data_preparator.item_id_map, add_rank_col, target_col)

    @staticmethod
    def _assemble_frame(ext_users: np.ndarray, ids: np.ndarray, scores: np.ndarray, counts: np.ndarray, item_id_map: tp.Any,
                        add_rank_col: bool, target_col: str) -> pd.DataFrame:
        """Ranker output ([n, k] ids / scores, valid entries leading every row, `counts` of them) -> the reference's long frame
        (models/base.py:735-791).  When every row is full — the usual case — the mask, the masked gathers and the running count
        are skipped (half of the 5 ms this takes for 16,384 x 10)."""
        # The columns are handed to pandas in ONE call and adopted as they are (copy=False; every array is freshly made here): building
        # the frame column by column copied each of them once more and consolidated the blocks when the rank column arrived (1.0 of the
        # 2.0 ms this took for 16,384 x 10 on the build host; 0.06 ms now)
        n, kk = ids.shape
        if n and bool((counts >= kk).all()) and bool((scores > -np.inf).all()):
            cols = {target_col: np.repeat(ext_users, kk), Columns.Item: item_id_map.convert_to_external(ids.reshape(-1)),
                    Columns.Score: scores.reshape(-1).astype(np.float32)}      # (a copy: the frame must not alias the caller's array)
            if add_rank_col:
                cols[Columns.Rank] = np.tile(np.arange(1, kk + 1, dtype=np.int64), n)
            return pd.DataFrame(cols, copy=False)
        valid = (np.arange(kk)[None, :] < counts[:, None]) & (scores > -np.inf)
        tt = np.repeat(ext_users, kk).reshape(len(ext_users), kk)[valid]
        ii = item_id_map.convert_to_external(ids[valid])
        cols = {target_col: tt, Columns.Item: ii, Columns.Score: scores[valid].astype(np.float32)}
        if add_rank_col:  # valid entries lead every row: rank = running count inside the row (models/base.py:788-789)
            cols[Columns.Rank] = (np.cumsum(valid, axis=1)[valid]).astype(np.int64)
        return pd.DataFrame(cols, copy=False)

    @staticmethod
    def _frame(targets: np.ndarray, items: np.ndarray, scores: np.ndarray, add_rank_col: bool, target_col: str) -> pd.DataFrame:
        df = pd.DataFrame({target_col: targets, Columns.Item: items, Columns.Score: scores.astype(np.float32)})
        if add_rank_col:
            df[Columns.Rank] = df.groupby(target_col, sort=False).cumcount() + 1  # models/base.py:788-789
        return df

    # ---- config / persistence -----------------------------------------------------------------------------
    def get_config(self, simple_types: bool = True) -> tp.Dict[str, tp.Any]:
        cfg = {"cls": _full_path(type(self)) if simple_types else type(self)}
        for k, v in self._params.items():
            if isinstance(v, type) or callable(v) and not isinstance(v, (int, float, str, bool)) and v is not None:
                cfg[k] = _full_path(v) if simple_types else v
            elif isinstance(v, (tuple, list)) and v and all(isinstance(t, type) for t in v):
                cfg[k] = [_full_path(t) for t in v] if simple_types else tuple(v)   # item_net_block_types (base.py:95-109)
            else:
                cfg[k] = v
        return cfg

    def get_params(self, simple_types: bool = True, sep: str = ".") -> tp.Dict[str, tp.Any]:
        flat: tp.Dict[str, tp.Any] = {}
        for k, v in self.get_config(simple_types).items():
            if isinstance(v, dict):
                for kk, vv in v.items():
                    flat[f"{k}{sep}{kk}"] = vv
            else:
                flat[k] = v
        return flat

    @classmethod
    def from_config(cls, config: tp.Dict[str, tp.Any]) -> "TransformerModelBase":
        cfg = dict(config)
        klass = _import_object(cfg.pop("cls", cls))
        for key in ("data_preparator_type", "transformer_layers_type", "similarity_module_type", "get_val_mask_func",
                    "item_net_constructor_type", "negative_sampler_type", "pos_encoding_type", "lightning_module_type",
                    "backbone_type", "get_trainer_func"):
            if cfg.get(key) is not None:
                cfg[key] = _import_object(cfg[key])
        if cfg.get("item_net_block_types") is not None:
            cfg["item_net_block_types"] = tuple(_import_object(t) for t in cfg["item_net_block_types"])
        return klass(**cfg)

    # ---- persistence: Lightning-layout checkpoints (checkpoint.py; base.py:591-724) --------------------------
    def __getstate__(self) -> tp.Dict[str, tp.Any]:
        """base.py:656-668: a fitted model pickles as its checkpoint, an unfitted one as its config."""
        if self.lightning_model is not None:
            buf = io.BytesIO()
            torch.save(ckpt.to_checkpoint(self, reference_paths=False), buf)
            return {"fitted_checkpoint": buf.getvalue(), "is_fitted": self.is_fitted}
        return {"model_config": self.get_config(simple_types=True)}

    def __setstate__(self, state: tp.Dict[str, tp.Any]) -> None:
        if "fitted_checkpoint" in state:
            checkpoint = torch.load(io.BytesIO(state["fitted_checkpoint"]), map_location="cpu", weights_only=False)
            loaded = self._model_from_checkpoint(checkpoint)
            loaded.is_fitted = state.get("is_fitted", True)
        else:
            loaded = self.from_config(state["model_config"])
        self.__dict__.update(loaded.__dict__)

    @classmethod
    def _model_from_checkpoint(cls, checkpoint: tp.Dict[str, tp.Any]) -> "TransformerModelBase":
        """base.py:591-654 without a Trainer: config -> model, item ids -> data preparator, dataset schema -> item net,
        then weights and Adam moments."""
        from .dataset import IdMap

        hyper = checkpoint["hyper_parameters"]
        config = ckpt.translate_config(dict(hyper["model_config"]))
        config.pop("cls", None)       # the class the method is called on decides (the reference stores a short name)
        for key, value in ((checkpoint.get("rectools_amd") or {}).get("model_params") or {}).items():
            config.setdefault(key, value)      # arguments only this engine has (`checkpoint.ENGINE_ONLY_PARAMS`) travel beside the config
        if config.get("get_trainer_func") is not None:
            # a checkpoint trained under a user-built Trainer: the factory is restored when its dotted path imports here (fit_partial()
            # of the restored model then reads it again, `_trainer_plan`), dropped aloud when it does not
            try:
                _import_object(config["get_trainer_func"])
            except Exception:      # pylint: disable=broad-except
                warnings.warn(f"checkpoint names get_trainer_func={config['get_trainer_func']!r}, which cannot be imported here: dropped, "
                              f"fit_partial() of the restored model runs the engine's loop with the model's own epoch settings")
                config["get_trainer_func"] = None
        loaded = cls.from_config(config)
        dp = loaded.data_preparator
        ext = hyper["item_external_ids"]
        ext = ext.tolist() if isinstance(ext, np.ndarray) else list(ext)
        dp.item_id_map = IdMap(np.asarray(ext, dtype=object) if any(isinstance(v, str) for v in ext) else np.asarray(ext))
        dp.extra_token_ids = dict(zip(dp.item_extra_tokens, dp.item_id_map.convert_to_internal(list(dp.item_extra_tokens))))
        loaded.dataset_schema = hyper.get("dataset_schema") or {}
        loaded._build_from_item_map(ckpt.item_net_schema(loaded.dataset_schema))
        device = loaded._device()
        weights = ckpt.strip_state_dict(checkpoint["state_dict"])
        loaded.torch_model.load_state_dict({k: v.to(device) for k, v in weights.items()})
        if checkpoint.get("optimizer_states"):
            mine = [n for n, p in loaded.torch_model.named_parameters() if p.requires_grad]
            theirs = [k for k in weights if k in set(mine)]
            ckpt.load_adam_state_dict(loaded.optimizer, checkpoint["optimizer_states"][0], theirs, mine)
        loaded.epochs_done = int(checkpoint.get("epoch", 0))
        loaded.history = list((checkpoint.get("rectools_amd") or {}).get("history", []))
        loaded.is_fitted = True
        return loaded

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, map_location: tp.Optional[tp.Any] = None,
                             model_params_update: tp.Optional[tp.Dict[str, tp.Any]] = None) -> "TransformerModelBase":
        """base.py:678-711.  Reads checkpoints written by `save_to_checkpoint` here AND Lightning checkpoints of the
        reference's models (class paths are translated).  `model_params_update`: flat `a.b` keys overriding
        `hyper_parameters.model_config` (e.g. to drop a `get_trainer_func` that cannot be imported)."""
        checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        if model_params_update:
            config = dict(checkpoint["hyper_parameters"]["model_config"])
            for key, value in model_params_update.items():
                node = config
                parts = key.split(".")
                for part in parts[:-1]:
                    node[part] = dict(node.get(part) or {})
                    node = node[part]
                node[parts[-1]] = value
            checkpoint["hyper_parameters"]["model_config"] = config
        return cls._model_from_checkpoint(checkpoint)

    def load_weights_from_checkpoint(self, checkpoint_path: str) -> None:
        """base.py:713-724: weights only, into an already fitted model."""
        if self.lightning_model is None:
            raise RuntimeError("Model weights cannot be loaded from checkpoint into unfitted model")
        checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        device = self._device()
        self.torch_model.load_state_dict({k: v.to(device) for k, v in ckpt.strip_state_dict(checkpoint["state_dict"]).items()})
        self._catalog_images = None

    def save_to_checkpoint(self, checkpoint_path: str, reference_paths: bool = True) -> None:
        """What `fit_trainer.save_checkpoint(path)` does for the reference (base.py:660-662), in the same dict layout."""
        torch.save(ckpt.to_checkpoint(self, reference_paths=reference_paths), checkpoint_path)

    def _build_from_item_map(self, item_net_schema: tp.List[dict]) -> None:
        """Modules of the right sizes without a dataset (the data preparator already holds the item id map)."""
        n_tokens = self.data_preparator.item_id_map.size
        dp_process = self.data_preparator.process_dataset_train
        self.data_preparator.process_dataset_train = lambda ds: None  # type: ignore
        try:
            self._build_model_from_dataset(None, item_net_schema)
        finally:
            self.data_preparator.process_dataset_train = dp_process  # type: ignore
        assert self.lightning_model.torch_model.item_model.n_items == n_tokens

    def save(self, path: str) -> int:
        data = pickle.dumps(self)
        with open(path, "wb") as f:
            return f.write(data)

    @classmethod
    def load(cls, path: str) -> "TransformerModelBase":
        with open(path, "rb") as f:
            return pickle.load(f)

    def dumps(self) -> bytes:
        return pickle.dumps(self)

    @classmethod
    def loads(cls, data: bytes) -> "TransformerModelBase":
        return pickle.loads(data)


class SASRecModel(TransformerModelBase):
    """SASRec (sasrec.py:315-541): causal attention, shifted-sequence objective, SASRec blocks by default."""

    def __init__(self, n_blocks: int = 2, n_heads: int = 4, n_factors: int = 256, use_pos_emb: bool = True,
                 use_causal_attn: bool = True, use_key_padding_mask: bool = False, dropout_rate: float = 0.2,
                 session_max_len: int = 100, dataloader_num_workers: int = 0, batch_size: int = 128, loss: str = "softmax",
                 n_negatives: int = 1, gbce_t: float = 0.2, lr: float = 0.001, epochs: int = 3, verbose: int = 0,
                 deterministic: bool = False, recommend_batch_size: int = 256, recommend_torch_device: tp.Optional[str] = None,
                 train_min_user_interactions: int = 2,
                 transformer_layers_type: tp.Type[hnn.TransformerLayersBase] = hnn.SASRecTransformerLayers,
                 data_preparator_type: tp.Type[TransformerDataPreparatorBase] = SASRecDataPreparator, **kwargs: tp.Any) -> None:
        super().__init__(
            data_preparator_type=data_preparator_type, transformer_layers_type=transformer_layers_type, n_blocks=n_blocks,
            n_heads=n_heads, n_factors=n_factors, use_pos_emb=use_pos_emb, use_causal_attn=use_causal_attn,
            use_key_padding_mask=use_key_padding_mask, dropout_rate=dropout_rate, session_max_len=session_max_len,
            dataloader_num_workers=dataloader_num_workers, batch_size=batch_size, loss=loss, n_negatives=n_negatives, gbce_t=gbce_t,
            lr=lr, epochs=epochs, verbose=verbose, deterministic=deterministic, recommend_batch_size=recommend_batch_size,
            recommend_torch_device=recommend_torch_device, train_min_user_interactions=train_min_user_interactions, **kwargs)


class BERT4RecModel(TransformerModelBase):
    """BERT4Rec (bert4rec.py:204-452): key-padding mask only, masked-item objective, Pre-LN blocks."""

    def __init__(self, n_blocks: int = 2, n_heads: int = 4, n_factors: int = 256, use_pos_emb: bool = True,
                 use_causal_attn: bool = False, use_key_padding_mask: bool = True, dropout_rate: float = 0.2,
                 session_max_len: int = 100, dataloader_num_workers: int = 0, batch_size: int = 128, loss: str = "softmax",
                 n_negatives: int = 1, gbce_t: float = 0.2, lr: float = 0.001, epochs: int = 3, verbose: int = 0,
                 mask_prob: float = 0.15, deterministic: bool = False, recommend_batch_size: int = 256,
                 recommend_torch_device: tp.Optional[str] = None, train_min_user_interactions: int = 2,
                 transformer_layers_type: tp.Type[hnn.TransformerLayersBase] = hnn.PreLNTransformerLayers,
                 data_preparator_type: tp.Type[TransformerDataPreparatorBase] = BERT4RecDataPreparator, **kwargs: tp.Any) -> None:
        self.mask_prob = mask_prob
        super().__init__(
            data_preparator_type=data_preparator_type, transformer_layers_type=transformer_layers_type, n_blocks=n_blocks,
            n_heads=n_heads, n_factors=n_factors, use_pos_emb=use_pos_emb, use_causal_attn=use_causal_attn,
            use_key_padding_mask=use_key_padding_mask, dropout_rate=dropout_rate, session_max_len=session_max_len,
            dataloader_num_workers=dataloader_num_workers, batch_size=batch_size, loss=loss, n_negatives=n_negatives, gbce_t=gbce_t,
            lr=lr, epochs=epochs, verbose=verbose, deterministic=deterministic, recommend_batch_size=recommend_batch_size,
            recommend_torch_device=recommend_torch_device, train_min_user_interactions=train_min_user_interactions,
            mask_prob=mask_prob, **kwargs)

    def _init_data_preparator(self, **extra: tp.Any) -> None:
        """bert4rec.py:430-441: `mask_prob` is the model's own argument, handed to the preparator beside `data_preparator_kwargs` — it
        must not be written INTO those kwargs (the config the reference reads back would then pass it twice)."""
        super()._init_data_preparator(mask_prob=self.mask_prob, **extra)


class HSTUModel(TransformerModelBase):
    """HSTU (hstu.py:412-729): STU blocks, relative time/position bias, cosine similarity, sqrt(d) embedding scale."""

    u2i_dist_default = "cosine"
    use_scale_factor_default = True

    def __init__(self, n_blocks: int = 2, n_heads: int = 4, n_factors: int = 256, use_pos_emb: bool = True,
                 use_causal_attn: bool = True, use_key_padding_mask: bool = False, dropout_rate: float = 0.2,
                 session_max_len: int = 100, dataloader_num_workers: int = 0, batch_size: int = 128, loss: str = "softmax",
                 n_negatives: int = 1, gbce_t: float = 0.2, lr: float = 0.001, epochs: int = 3, verbose: int = 0,
                 relative_time_attention: bool = True, relative_pos_attention: bool = True, deterministic: bool = False,
                 recommend_batch_size: int = 256, recommend_torch_device: tp.Optional[str] = None,
                 train_min_user_interactions: int = 2,
                 transformer_layers_type: tp.Type[hnn.TransformerLayersBase] = hnn.STULayers,
                 data_preparator_type: tp.Type[TransformerDataPreparatorBase] = SASRecDataPreparator, **kwargs: tp.Any) -> None:
        if n_factors % n_heads != 0:
            raise ValueError("n_factors must be divisible by n_heads without remainder")  # hstu.py:606-607
        if use_key_padding_mask:
            warnings.warn("'use_key_padding_mask' is not supported for HSTU and enforced to False.")  # hstu.py:608-612
            use_key_padding_mask = False
        self.relative_time_attention, self.relative_pos_attention = relative_time_attention, relative_pos_attention
        super().__init__(
            data_preparator_type=data_preparator_type, transformer_layers_type=transformer_layers_type, n_blocks=n_blocks,
            n_heads=n_heads, n_factors=n_factors, use_pos_emb=use_pos_emb, use_causal_attn=use_causal_attn,
            use_key_padding_mask=use_key_padding_mask, dropout_rate=dropout_rate, session_max_len=session_max_len,
            dataloader_num_workers=dataloader_num_workers, batch_size=batch_size, loss=loss, n_negatives=n_negatives, gbce_t=gbce_t,
            lr=lr, epochs=epochs, verbose=verbose, deterministic=deterministic, recommend_batch_size=recommend_batch_size,
            recommend_torch_device=recommend_torch_device, train_min_user_interactions=train_min_user_interactions,
            relative_time_attention=relative_time_attention, relative_pos_attention=relative_pos_attention, **kwargs)

    def _layer_kwargs(self) -> tp.Dict[str, tp.Any]:
        """hstu.py:660-674: head size, window and the two bias switches derive from the model's own arguments.  Where the reference
        passes them beside `transformer_layers_kwargs` (a duplicate there is a TypeError), this engine lets the kwargs override the head
        sizes (`linear_hidden_dim != attention_dim`, hstu.py:186-221, through `nn.DimPlan`) — a superset; a config without such an
        override is the reference's config."""
        hd = self.n_factors // self.n_heads
        tkw = dict(linear_hidden_dim=hd, attention_dim=hd)
        tkw.update(self._kw(self.transformer_layers_kwargs))
        tkw.update(session_max_len=self.session_max_len, relative_time_attention=self.relative_time_attention,
                   relative_pos_attention=self.relative_pos_attention)
        return tkw

    def _init_data_preparator(self, **extra: tp.Any) -> None:
        """hstu.py:676-694: a relative time bias needs the batches' timestamps."""
        if self.relative_time_attention and "add_unix_ts" not in self._kw(self.data_preparator_kwargs):
            extra["add_unix_ts"] = True
        super()._init_data_preparator(**extra)

    @property
    def require_recommend_context(self) -> bool:
        return bool(self.relative_time_attention)  # hstu.py:719-729
