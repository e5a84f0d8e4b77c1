"""Torch-model mirror of the reference's transformer backbone, computing with the HIP ops (`rectools_amd.ops`).

Module / parameter names and shapes are IDENTICAL to the reference, so `state_dict`s
interchange with `rectools.models.nn.transformers.*`:

  item_model.item_net_blocks.0.ids_emb.weight            IdEmbeddingsItemNet            item_net.py:236-281
  item_model.item_net_blocks.1.embedding_bag.weight      CatFeaturesItemNet (+ buffers offsets, emb_bag_inputs,
                                                         input_lengths)                  item_net.py:60-233
  pos_encoding_layer.pos_emb.weight                      LearnableInversePositionalEncoding  net_blocks.py:346-400
  transformer_layers.transformer_blocks.i.*              SASRecTransformerLayers        sasrec.py:169-304
                                                         PreLNTransformerLayers         net_blocks.py:188-335
                                                         LiGRLayers                     ligr.py:25-191
  transformer_layers.stu_blocks.i.*                      STULayers                      hstu.py:156-399
  similarity_module                                      DistanceSimilarityModule       similarity.py:67-140

What differs from the reference's plug-in signatures: layer stacks receive the flattened `[B*L, d]` activations
plus the item ids (masks are derived in-kernel from ids and indices; no `[B*H, L, L]` mask tensor is ever
built, torch_backbone.py:172-218,249-257), and an ids-only item table is read in place (no `get_all_embeddings()`
copy, item_net.py:361-368); with a category-feature block the catalog matrix is one fused pass per step (K1b).
"""
from __future__ import annotations

import dataclasses
import math
import os
import typing as tp
import warnings

import numpy as np
import torch
from torch import nn

from . import ops
from .rank import Distance

Batch = tp.Dict[str, torch.Tensor]


# ---- parameter holders with the reference's names -----------------------------------------------------
class LinearParams(nn.Module):
    def __init__(self, n_in: int, n_out: int, bias: bool = True) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n_out, n_in))
        self.bias = nn.Parameter(torch.empty(n_out)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))  # nn.Linear default; 1-D params keep it
        if bias:
            bound = 1 / math.sqrt(n_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x: torch.Tensor, residual: tp.Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
        return ops.linear(x, self.weight, self.bias, residual, relu)


class LayerNormParams(nn.Module):
    def __init__(self, d: int, eps: float = 1e-5) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))
        self.eps = eps
        self.cols: tp.Optional[tp.Tuple[int, int]] = None    # (grp, grp_real) when the rows carry zero columns (`DimPlan`)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.layer_norm(x, self.weight, self.bias, self.eps, self.cols)


class MultiheadAttnParams(nn.Module):
    """Parameters of torch.nn.MultiheadAttention(d, H, batch_first=True) (packed in_proj), SURVEY.md A.4."""

    def __init__(self, d: int, n_heads: int) -> None:
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = LinearParams(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)
        self.d, self.n_heads = d, n_heads
        self.scale = 0.0      # logit scale; 0 = 1 / sqrt(d / n_heads).  Heads padded with zero columns keep the scale of their real size

    def forward(self, q_in: torch.Tensor, kv_in: tp.Optional[torch.Tensor], ids: torch.Tensor, B: int, L: int,
                causal: bool, keypad: bool, p: float, residual: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
        d = self.d
        if kv_in is None:  # self-attention on one input: one packed GEMM, one packed gradient
            qkv = ops.linear(q_in, self.in_proj_weight, self.in_proj_bias)
            o = ops.mha_packed(qkv, ids, B, self.n_heads, L, causal, keypad, p, self.scale)
            return self.out_proj(o, residual=residual)
        else:  # SASRec: Q from LN(x), K/V from x (sasrec.py:221-224)
            q = ops.linear(q_in, self.in_proj_weight[:d], self.in_proj_bias[:d])
            kv = ops.linear(kv_in, self.in_proj_weight[d:], self.in_proj_bias[d:])
            k, v = kv[:, :d], kv[:, d:]
        o = ops.mha(q, k, v, ids, B, self.n_heads, L, causal, keypad, p, self.scale)
        return self.out_proj(o, residual=residual)


def _take_last(x: torch.Tensor, B: int, L: int) -> torch.Tensor:
    """[B*L, d] -> [B, d]: the row of the last position of every session."""
    return x.view(B, L, -1)[:, L - 1, :].contiguous()


def _attend_last(mha: MultiheadAttnParams, kv_in: torch.Tensor, q_last: torch.Tensor, ids: torch.Tensor, B: int, L: int,
                 causal: bool, keypad: bool) -> torch.Tensor:
    """Inference: attention output (before out_proj) of the LAST query of every session.  Keys / values are projected for every
    position of `kv_in` [B*L, d], the query for `q_last` [B, d] only (`rt_mha_last_fwd`)."""
    d = mha.d
    kv = ops.linear(kv_in, mha.in_proj_weight[d:], mha.in_proj_bias[d:])          # [B*L, 2d]
    q = ops.linear(q_last, mha.in_proj_weight[:d], mha.in_proj_bias[:d])          # [B, d]
    out = torch.empty((B, d), dtype=torch.float32, device=q.device)
    ops._c("rt_mha_last_fwd_scaled", q, d, kv, 2 * d, kv[:, d:], 2 * d, ids.reshape(-1), B, mha.n_heads, L, d // mha.n_heads,   # pylint: disable=protected-access
           float(mha.scale), int(causal), int(keypad), out, d)
    return out


# ---- item net / positions -----------------------------------------------------------------------------
class EmbeddingParams(nn.Module):
    def __init__(self, n: int, d: int) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, d))
        nn.init.normal_(self.weight)


class IdEmbeddingsItemNet(nn.Module):
    """Item embeddings based on item ids only (item_net.py:236-281)."""

    def __init__(self, n_factors: int, n_items: int, dropout_rate: float = 0.0, **kwargs: tp.Any) -> None:
        super().__init__()
        self.n_items = n_items
        self.ids_emb = EmbeddingParams(n_items, n_factors)

    @classmethod
    def from_dataset(cls, dataset: tp.Any, n_factors: int, dropout_rate: float, **kwargs: tp.Any) -> "IdEmbeddingsItemNet":
        return cls(n_factors, dataset.item_id_map.size, dropout_rate)

    @property
    def out_dim(self) -> int:
        return self.ids_emb.weight.shape[1]


class CatFeaturesItemNet(nn.Module):
    """Item embeddings from categorical item features (item_net.py:60-233): an embedding per (feature, value) pair,
    summed over the pairs an item carries (`nn.EmbeddingBag(mode="sum")`), then dropout.

    Parameter / buffer names and shapes are the reference's (`embedding_bag.weight`, `offsets`, `emb_bag_inputs`,
    `input_lengths`), so state dicts are interchangeable.  There is no per-call gather of the structure
    (`_get_item_inputs_offsets`, item_net.py:123-132): the kernels read the three buffers in place."""

    def __init__(self, emb_bag_inputs: torch.Tensor, input_lengths: torch.Tensor, offsets: torch.Tensor,
                 n_cat_feature_values: int, n_factors: int, dropout_rate: float, **kwargs: tp.Any) -> None:
        super().__init__()
        self.n_cat_feature_values = n_cat_feature_values
        self.embedding_bag = EmbeddingParams(n_cat_feature_values, n_factors)
        self.dropout_rate = dropout_rate
        self.register_buffer("offsets", offsets.to(torch.int64))
        self.register_buffer("emb_bag_inputs", emb_bag_inputs.to(torch.int64))
        self.register_buffer("input_lengths", input_lengths.to(torch.int64))
        self._structure: tp.Optional[ops.BagStructure] = None

    @classmethod
    def from_dataset(cls, dataset: tp.Any, n_factors: int, dropout_rate: float, **kwargs: tp.Any) -> tp.Optional["CatFeaturesItemNet"]:
        """item_net.py:149-191: None (block skipped, with the reference's warnings) unless the dataset carries sparse
        item features with at least one categorical column."""
        feats = getattr(dataset, "item_features", None)
        if feats is None:
            warnings.warn("Ignoring `CatFeaturesItemNet` block because dataset doesn't contain item features.")
            return None
        if not hasattr(feats, "get_cat_features"):
            warnings.warn("Ignoring `CatFeaturesItemNet` block because dataset item features are dense and "
                          "one-hot-encoded categorical features were not created when constructing dataset.")
            return None
        cat = feats.get_cat_features()
        if len(cat.names) == 0:
            warnings.warn("Ignoring `CatFeaturesItemNet` block because dataset item features do not contain categorical features.")
        if cat.values.size == 0:
            return None
        csr = cat.values.tocsr()
        indptr = torch.as_tensor(np.asarray(csr.indptr), dtype=torch.int64)
        return cls(emb_bag_inputs=torch.as_tensor(np.asarray(csr.indices), dtype=torch.int64), offsets=indptr[:-1],
                   input_lengths=torch.diff(indptr), n_cat_feature_values=len(cat.names), n_factors=n_factors,
                   dropout_rate=dropout_rate)

    def structure(self) -> ops.BagStructure:
        if self._structure is None or self._structure.inputs.device != self.emb_bag_inputs.device:
            self._structure = ops.BagStructure(self.emb_bag_inputs, self.offsets, self.input_lengths, self.n_cat_feature_values)
        return self._structure

    def _load_from_state_dict(self, *args: tp.Any, **kwargs: tp.Any) -> None:
        super()._load_from_state_dict(*args, **kwargs)
        self._structure = None     # the buffers may describe another catalog now

    @property
    def out_dim(self) -> int:
        return self.embedding_bag.weight.shape[1]


class SumOfEmbeddingsConstructor(nn.Module):
    """Item-net constructor (item_net.py:289-487): the catalog matrix is the sum of its blocks' embeddings.

    With id embeddings only, the parameter itself is the matrix (no `get_all_embeddings()` copy).  With a
    CatFeaturesItemNet block the matrix is produced by one fused pass (`ops.item_table`, K1b): every read of `table` runs it (with a
    fresh dropout mask in training), so callers take it ONCE per step / per recommend() call and pass it on, as the
    reference's backbone does with `get_all_embeddings()` (torch_backbone.py:290-291)."""

    def __init__(self, n_items: int, item_net_blocks: tp.Sequence[nn.Module]) -> None:
        super().__init__()
        if len(item_net_blocks) == 0:
            raise ValueError("At least one type of net to calculate item embeddings should be provided.")
        ids = [b for b in item_net_blocks if isinstance(b, IdEmbeddingsItemNet)]
        cats = [b for b in item_net_blocks if isinstance(b, CatFeaturesItemNet)]
        for b in item_net_blocks:      # a plugged block class (item_net.py:26-57, ItemNetBase): it must be able to produce its [n_items, d] rows
            if not isinstance(b, (IdEmbeddingsItemNet, CatFeaturesItemNet)) and not callable(getattr(b, "get_all_embeddings", None)):
                raise TypeError(f"item net block {type(b).__name__} offers no get_all_embeddings(): the catalog matrix is the sum of the "
                                f"blocks' [n_items, n_factors] rows (item_net.py:361-368)")
        self.n_items = n_items
        self.n_item_blocks = len(item_net_blocks)
        self.item_net_blocks = nn.ModuleList(item_net_blocks)
        # positions, not references: a second attribute holding a block would register it twice in the state dict.  The first id block and
        # the first category block are produced by ONE fused pass (K1b); any further block of either kind — the reference sums whatever
        # list it is given (item_net.py:463-482) — and any plugged block adds its rows to that
        self._ids_at = item_net_blocks.index(ids[0]) if ids else None
        self._cat_at = item_net_blocks.index(cats[0]) if cats else None
        self._more_at = [i for i in range(len(item_net_blocks)) if i not in (self._ids_at, self._cat_at)]

    @classmethod
    def from_dataset(cls, dataset: tp.Any, n_factors: int, dropout_rate: float,
                     item_net_block_types: tp.Sequence[tp.Type[nn.Module]], **kwargs: tp.Any) -> "SumOfEmbeddingsConstructor":
        blocks = []
        for block_type in item_net_block_types:
            block = block_type.from_dataset(dataset, n_factors, dropout_rate, **kwargs)
            if block is not None:
                blocks.append(block)
        return cls(dataset.item_id_map.size, blocks)

    def _rows_of(self, block: nn.Module) -> torch.Tensor:
        if isinstance(block, IdEmbeddingsItemNet):
            return block.ids_emb.weight
        if isinstance(block, CatFeaturesItemNet):
            return ops.item_table(None, block.embedding_bag.weight, block.structure(), block.dropout_rate if self.training else 0.0)
        return block.get_all_embeddings()

    @property
    def table(self) -> torch.Tensor:
        ids_w = self.item_net_blocks[self._ids_at].ids_emb.weight if self._ids_at is not None else None
        if self._cat_at is None:
            out = ids_w
        else:
            cat = self.item_net_blocks[self._cat_at]
            p = cat.dropout_rate if self.training else 0.0
            out = ops.item_table(ids_w, cat.embedding_bag.weight, cat.structure(), p)
        for i in self._more_at:
            rows = self._rows_of(self.item_net_blocks[i])
            out = rows if out is None else ops.add(out, rows)
        return out

    def get_all_embeddings(self) -> torch.Tensor:
        """The table itself (the reference re-materialises it with a gather every call, item_net.py:361-368)."""
        return self.table


class LearnableInversePositionalEncoding(nn.Module):
    def __init__(self, use_pos_emb: bool, session_max_len: int, n_factors: int, use_scale_factor: bool = False,
                 **kwargs: tp.Any) -> None:
        super().__init__()
        self.pos_emb = EmbeddingParams(session_max_len, n_factors) if use_pos_emb else None
        self.use_scale_factor = use_scale_factor

    def forward(self, sessions: torch.Tensor) -> torch.Tensor:
        """net_blocks.py:374-400 on [B, L, d] item embeddings, out of torch ops.  The stock backbone never calls it — the scale
        and the inverse positional rows ride inside `rt_embed_fwd` (`TransformerTorchBackbone._fused_pos`); it serves subclasses
        that extend the encoding and call `super().forward`."""
        _, L, d = sessions.shape
        if self.use_scale_factor:
            sessions = sessions * (float(getattr(self, "d_real", None) or d) ** 0.5)
        if self.pos_emb is not None:
            sessions = sessions + self.pos_emb.weight[:L].flip(0)[None]
        return sessions


# ---- feed-forward networks ----------------------------------------------------------------------------
class PointWiseFeedForward(nn.Module):
    def __init__(self, n_factors: int, n_factors_ff: int, dropout_rate: float, activation: str, bias: bool = True) -> None:
        super().__init__()
        self.ff_linear_1 = LinearParams(n_factors, n_factors_ff, bias)
        self.ff_linear_2 = LinearParams(n_factors_ff, n_factors, bias)
        self.activation = activation
        self.p = dropout_rate

    def forward(self, x: torch.Tensor, residual: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.activation == "relu":
            h = self.ff_linear_1(x, relu=True)
            h = ops.dropout(h, self.p if self.training else 0.0)
        else:
            z = self.ff_linear_1(x)
            h = ops.act_dropout(z, ops.ACT_GELU, self.p if self.training else 0.0)
        return self.ff_linear_2(h, residual=residual)


class SwigluFeedForward(nn.Module):
    def __init__(self, n_factors: int, n_factors_ff: int, dropout_rate: float, bias: bool = True) -> None:
        super().__init__()
        self.ff_linear_1 = LinearParams(n_factors, n_factors_ff, bias)
        self.ff_linear_2 = LinearParams(n_factors_ff, n_factors, bias)
        self.ff_linear_3 = LinearParams(n_factors, n_factors_ff, bias)
        self.p = dropout_rate

    def forward(self, x: torch.Tensor, residual: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
        h = ops.swiglu(self.ff_linear_1(x), self.ff_linear_3(x), self.p if self.training else 0.0)
        return self.ff_linear_2(h, residual=residual)


def init_feed_forward(n_factors: int, ff_factors_multiplier: int, dropout_rate: float, ff_activation: str,
                      bias: bool = True) -> nn.Module:
    if ff_activation == "swiglu":
        return SwigluFeedForward(n_factors, n_factors * ff_factors_multiplier, dropout_rate, bias=bias)
    if ff_activation in ("gelu", "relu"):
        return PointWiseFeedForward(n_factors, n_factors * ff_factors_multiplier, dropout_rate, ff_activation, bias=bias)
    raise ValueError(f"Unsupported ff_activation: {ff_activation}")  # net_blocks.py:151


# ---- layer stacks ---------------------------------------------------------------------------------------
class TransformerLayersBase(nn.Module):
    def forward(self, seqs: torch.Tensor, ids: torch.Tensor, B: int, L: int, causal: bool, keypad: bool,
                batch: Batch) -> torch.Tensor:
        raise NotImplementedError()

    def _fresh_planes(self, refresh: bool = True):
        """bf16 planes of the stack's weights for the pre-split-weight GEMM (`ops.WeightPlanes`, csrc/rt_gemm_wp.hip), re-split NOW (one
        launch over the stack's few MB of parameters): whatever changed the weights since the last pass — the optimiser, a checkpoint,
        a test poking a parameter — the planes the blocks are about to read are current.  None when the parameters are not views of one
        flat buffer."""
        if not ops.weight_planes_enabled():
            return None
        ps = self.__dict__.get("_planes_params")
        if ps is None:      # the module walk is ~0.1 ms of host time: once, not per pass (the Parameter objects of a stack do not change)
            ps = list(self.parameters())
            object.__setattr__(self, "_planes_params", ps)
        key = tuple([p.data_ptr() for p in ps])
        cached = self.__dict__.get("_planes_cache")
        if cached is None or cached[0] != key:
            cached = (key, ops.WeightPlanes(ps))
            object.__setattr__(self, "_planes_cache", cached)
        if refresh:      # (False: the caller re-splits the range itself — `lightning.NativeSasrecStep`, inside its one compiled call)
            cached[1].refresh()
        return cached[1] if cached[1].ok else None



class SASRecTransformerLayer(nn.Module):
    def __init__(self, n_factors: int, n_heads: int, dropout_rate: float) -> None:
        super().__init__()
        self.multi_head_attn = MultiheadAttnParams(n_factors, n_heads)
        self.q_layer_norm = LayerNormParams(n_factors)
        self.ff_layer_norm = LayerNormParams(n_factors)
        self.feed_forward = PointWiseFeedForward(n_factors, n_factors, dropout_rate, "relu")
        self.p = dropout_rate
        self.generic = False      # `DimPlan`: zero-padded columns -> the block runs out of the individual ops (column-aware LayerNorm)

    def forward(self, seqs, ids, B, L, causal, keypad):
        """`seqs` comes in unmasked: the timeline mask (sasrec.py:300) is the first step of the fused block."""
        p = self.p if self.training else 0.0
        ff = self.feed_forward
        if not self.generic and ff.ff_linear_1.bias is not None and ff.ff_linear_2.bias is not None and ff.activation == "relu":
            mha = self.multi_head_attn
            return ops.sasrec_layer(
                seqs, ids, B, L, mha.n_heads, causal, keypad, p,
                (self.q_layer_norm.weight, self.q_layer_norm.bias, self.q_layer_norm.eps),
                (mha.in_proj_weight, mha.in_proj_bias), (mha.out_proj.weight, mha.out_proj.bias),
                (self.ff_layer_norm.weight, self.ff_layer_norm.bias, self.ff_layer_norm.eps),
                (ff.ff_linear_1.weight, ff.ff_linear_1.bias), (ff.ff_linear_2.weight, ff.ff_linear_2.bias))
        return self.forward_modular(ops.mul_mask(seqs, None, ids), ids, B, L, causal, keypad)

    def forward_last(self, seqs, ids, B, L, causal, keypad):
        """Inference: the block's output at the last position of every session only, [B, d] (see ops.sasrec_layer_last)."""
        ff, mha = self.feed_forward, self.multi_head_attn
        if self.generic or self.training or torch.is_grad_enabled() or ff.ff_linear_1.bias is None or ff.ff_linear_2.bias is None \
                or ff.activation != "relu":
            return self(seqs, ids, B, L, causal, keypad).view(B, L, -1)[:, -1, :].contiguous()
        return ops.sasrec_layer_last(
            seqs, ids, B, L, mha.n_heads, causal, keypad,
            (self.q_layer_norm.weight, self.q_layer_norm.bias, self.q_layer_norm.eps),
            (mha.in_proj_weight, mha.in_proj_bias), (mha.out_proj.weight, mha.out_proj.bias),
            (self.ff_layer_norm.weight, self.ff_layer_norm.bias, self.ff_layer_norm.eps),
            (ff.ff_linear_1.weight, ff.ff_linear_1.bias), (ff.ff_linear_2.weight, ff.ff_linear_2.bias))

    def packed_ok(self) -> bool:
        ff = self.feed_forward
        return not self.generic and ff.ff_linear_1.bias is not None and ff.ff_linear_2.bias is not None and ff.activation == "relu"

    def forward_packed(self, seqs, cu, B, window, pad_keys, last_rows=None, rows_real=None, planes=None, kv_in=None, q_in=None, Q_in=None):
        """Inference over packed sessions (no padding rows; see ops.sasrec_layer_packed): [Np, d], or [B, d] with `last_rows`.
        kv_in (+ q_in, Q_in): this block's keys | values [Np, 2d] (and LN1(x), the projected queries) made by the caller
        (`SASRecTransformerLayers.forward_last_packed`, first block)."""
        ff, mha = self.feed_forward, self.multi_head_attn
        return ops.sasrec_layer_packed(
            seqs, cu, B, mha.n_heads, window, pad_keys, last_rows,
            (self.q_layer_norm.weight, self.q_layer_norm.bias, self.q_layer_norm.eps),
            (mha.in_proj_weight, mha.in_proj_bias), (mha.out_proj.weight, mha.out_proj.bias),
            (self.ff_layer_norm.weight, self.ff_layer_norm.bias, self.ff_layer_norm.eps),
            (ff.ff_linear_1.weight, ff.ff_linear_1.bias), (ff.ff_linear_2.weight, ff.ff_linear_2.bias), rows_real=rows_real, planes=planes,
            kv_in=kv_in, q_in=q_in, Q_in=Q_in)

    def forward_packed_train(self, seqs, cu, B, window, pad_keys, rows_real=None, planes=None):
        """The block on packed rows with autograd (training): ONE autograd node (`ops.sasrec_layer_packed_train`) when the feed-forward
        is the fused kind (ReLU, biases), else the individual ops."""
        ff, mha = self.feed_forward, self.multi_head_attn
        if not self.packed_ok():
            return self.forward_packed_modular(seqs, cu, B, window, pad_keys)
        return ops.sasrec_layer_packed_train(
            seqs, cu, B, mha.n_heads, window, pad_keys, self.p if self.training else 0.0,
            (self.q_layer_norm.weight, self.q_layer_norm.bias, self.q_layer_norm.eps),
            (mha.in_proj_weight, mha.in_proj_bias), (mha.out_proj.weight, mha.out_proj.bias),
            (self.ff_layer_norm.weight, self.ff_layer_norm.bias, self.ff_layer_norm.eps),
            (ff.ff_linear_1.weight, ff.ff_linear_1.bias), (ff.ff_linear_2.weight, ff.ff_linear_2.bias), rows_real=rows_real, planes=planes)

    def forward_packed_modular(self, seqs, cu, B, window, pad_keys):
        """The same block out of the individual autograd ops (as `forward_modular`; the cross-check of the fused packed node).  No
        masks: there are no pad rows; the pad keys of the reference's window are the virtual key of `ops.mha_varlen`."""
        p = self.p if self.training else 0.0
        mha, d = self.multi_head_attn, seqs.shape[1]
        q = self.q_layer_norm(seqs)
        in_w, in_b = mha.in_proj_weight, mha.in_proj_bias
        Q = ops.linear(q, in_w[:d], in_b[:d])
        KV = ops.linear(seqs, in_w[d:], in_b[d:])                      # K | V from the raw block input (sasrec.py:221-224)
        bk, bv = (in_b[d:2 * d], in_b[2 * d:]) if pad_keys else (None, None)
        att = ops.mha_varlen(Q, KV, bk, bv, cu, B, mha.n_heads, window, p)
        seqs = mha.out_proj(att, residual=q)
        ff_in = self.ff_layer_norm(seqs)
        if p > 0:
            return ops.add(ops.dropout(self.feed_forward(ff_in), p), ff_in)
        return self.feed_forward(ff_in, residual=ff_in)

    def forward_modular(self, seqs, ids, B, L, causal, keypad):
        """Same block out of the individual autograd ops (`seqs` already masked); kept as the cross-check of the fused node."""
        p = self.p if self.training else 0.0
        q = self.q_layer_norm(seqs)
        seqs = self.multi_head_attn(q, seqs, ids, B, L, causal, keypad, p, residual=q)   # q + mha(q, x, x)
        ff_in = self.ff_layer_norm(seqs)
        if p > 0:
            return ops.add(ops.dropout(self.feed_forward(ff_in), p), ff_in)
        return self.feed_forward(ff_in, residual=ff_in)


class SASRecTransformerLayers(TransformerLayersBase):
    def __init__(self, n_blocks: int, n_factors: int, n_heads: int, dropout_rate: float, **kwargs: tp.Any) -> None:
        super().__init__()
        self.n_blocks = n_blocks
        self.transformer_blocks = nn.ModuleList(
            [SASRecTransformerLayer(n_factors, n_heads, dropout_rate) for _ in range(n_blocks)])
        self.last_layernorm = LayerNormParams(n_factors, eps=1e-8)

    def forward(self, seqs, ids, B, L, causal, keypad, batch):
        for blk in self.transformer_blocks:
            seqs = blk(seqs, ids, B, L, causal, keypad)   # seqs *= timeline_mask (sasrec.py:300) happens inside
        ln = self.last_layernorm
        return ops.layer_norm_masked(seqs, ids, ln.weight, ln.bias, ln.eps, ln.cols)   # seqs * mask, then the last LayerNorm

    def forward_last(self, seqs, ids, B, L, causal, keypad, batch):
        """Inference: [B, d] encodings of the last position (what recommend() keeps of `encode_sessions`, lightning.py:393-397):
        every block but the final one runs in full, the final one on one query row per session."""
        blocks = list(self.transformer_blocks)
        for blk in blocks[:-1]:
            seqs = blk(seqs, ids, B, L, causal, keypad)
        last = blocks[-1].forward_last(seqs, ids, B, L, causal, keypad)
        last = ops.mul_mask(last, None, ids.view(B, L)[:, L - 1].contiguous())
        return self.last_layernorm(last)

    def packed_ok(self, n_factors: int, window: int, causal: bool, keypad: bool = False) -> bool:
        """Can recommend() encode packed sessions with this stack?  Causal attention (the closed form of the pad keys needs left
        padding + a causal mask, or no pad keys at all), ReLU feed-forward with biases, head size 32 / 64, K / V image in LDS."""
        blocks = list(self.transformer_blocks)
        return bool(blocks) and causal and all(b.packed_ok() for b in blocks) and \
            ops.mha_varlen_supported(blocks[0].multi_head_attn.n_heads, n_factors, window)

    def forward_packed_train(self, seqs, cu, B, window, keypad, rows_real=None, causal=True):
        """Training forward over packed rows, [Np, d] (every row is a real position or belongs to the unused tail).  rows_real: the
        number of session rows when the caller knows it on the host (selects the native block executor)."""
        planes = self._fresh_planes() if rows_real is not None else None
        for blk in self.transformer_blocks:
            seqs = blk.forward_packed_train(seqs, cu, B, window, not keypad, rows_real, planes)
        return self.last_layernorm(seqs)

    accepts_first_kv = True      # (`TransformerTorchBackbone.encode_last_packed` asks before it builds the projected tables)
    accepts_first_pre = "sasrec"

    def forward_last_packed(self, seqs, cu, B, window, keypad, rows_real=None, causal=True, first_kv=None, first_pre=None):
        """[B, d] encodings of the last position from PACKED rows: every block input is the real rows only — the
        reference masks pad rows to zero before each block (sasrec.py:300) and their only trace, the pad keys a causal block
        without key-padding masks shows to every query, is the virtual key of `rt_mha_varlen_*`.  first_kv(in_proj_weight,
        in_proj_bias) -> [Np, 2d]: the FIRST block's keys | values made from projected tables (its input is embedding row +
        positional row, the key / value projection is linear: sasrec.py:221-224 reads the raw block input).  first_pre(block) ->
        (LN1(x), Q, K | V): all three of the first block's inputs from projected tables (`rt_embed_block1_fwd`); `seqs` is not read then."""
        blocks = list(self.transformer_blocks)
        planes = self._fresh_planes() if rows_real is not None else None
        for i, blk in enumerate(blocks[:-1]):
            kv_in = q_in = Q_in = None
            if i == 0 and first_pre is not None:
                q_in, Q_in, kv_in = first_pre(blk)
            elif i == 0 and first_kv is not None:
                kv_in = first_kv(blk.multi_head_attn.in_proj_weight, blk.multi_head_attn.in_proj_bias)
            seqs = blk.forward_packed(seqs, cu, B, window, not keypad, rows_real=rows_real, planes=planes, kv_in=kv_in, q_in=q_in, Q_in=Q_in)
        last = blocks[-1].forward_packed(seqs, cu, B, window, not keypad, last_rows=cu[1:] - 1, rows_real=rows_real, planes=planes)
        return self.last_layernorm(last)


class PreLNTransformerLayer(nn.Module):
    def __init__(self, n_factors: int, n_heads: int, dropout_rate: float, ff_factors_multiplier: int = 4) -> None:
        super().__init__()
        self.multi_head_attn = MultiheadAttnParams(n_factors, n_heads)
        self.layer_norm_1 = LayerNormParams(n_factors)
        self.layer_norm_2 = LayerNormParams(n_factors)
        self.feed_forward = PointWiseFeedForward(n_factors, n_factors * ff_factors_multiplier, dropout_rate, "gelu")
        self.p = dropout_rate
        self.generic = False      # `DimPlan`: zero-padded columns (no packed rows, no native executor)

    def forward(self, seqs, ids, B, L, causal, keypad):
        p = self.p if self.training else 0.0
        ln1, ln2 = self.layer_norm_1, self.layer_norm_2
        h, skip = ops.layer_norm_skip(seqs, ln1.weight, ln1.bias, ln1.eps, ln1.cols)     # skip = seqs, its gradient rides in LN1's backward
        if p > 0:
            seqs = ops.dropout_add(self.multi_head_attn(h, None, ids, B, L, causal, keypad, p), skip, p)
            g, skip = ops.layer_norm_skip(seqs, ln2.weight, ln2.bias, ln2.eps, ln2.cols)
            seqs = ops.dropout_add(self.feed_forward(g), skip, p)
            return ops.dropout(seqs, p)  # dropout_3 (net_blocks.py:260)
        seqs = self.multi_head_attn(h, None, ids, B, L, causal, keypad, 0.0, residual=skip)
        g, skip = ops.layer_norm_skip(seqs, ln2.weight, ln2.bias, ln2.eps, ln2.cols)
        return self.feed_forward(g, residual=skip)


    def forward_last(self, seqs, ids, B, L, causal, keypad):
        """Inference: the block's output at the last position of every session, [B, d].  Only LN_1 and the key / value projection
        see every position; the query, out_proj, LN_2 and the 4x feed-forward (two thirds of the block's flops) run on B rows."""
        h = self.layer_norm_1(seqs)
        a = _attend_last(self.multi_head_attn, h, _take_last(h, B, L), ids, B, L, causal, keypad)
        x1 = self.multi_head_attn.out_proj(a, residual=_take_last(seqs, B, L))
        return self.feed_forward(self.layer_norm_2(x1), residual=x1)

    def forward_packed(self, seqs, cu, B, window, causal, covers_all_rows=False, qkv_in=None):
        """The block over PACKED sessions ([Np, d], real positions only): with key-padding masks the reference's pad positions are
        seen by no real query (net_blocks.py:236-262 under torch_backbone.py:254's mask), so dropping their rows changes nothing for
        the real ones.  Same ops as `forward`; the attention is `ops.mha_varlen_qkv` on the packed in_proj output.  qkv_in [Np, 3d]
        (inference, first block): the in_proj output made by the caller from projected tables (`rt_embed_block1_preln_fwd`)."""
        p = self.p if self.training else 0.0
        ln1, ln2, mha = self.layer_norm_1, self.layer_norm_2, self.multi_head_attn
        if qkv_in is not None:
            skip, qkv = seqs, qkv_in
        else:
            h, skip = ops.layer_norm_skip(seqs, ln1.weight, ln1.bias, ln1.eps)
            qkv = ops.linear(h, mha.in_proj_weight, mha.in_proj_bias)
        a = ops.mha_varlen_qkv(qkv, cu, B, mha.n_heads, window, causal, p, covers_all_rows)
        if p > 0:
            seqs = ops.dropout_add(mha.out_proj(a), skip, p)
            g, skip = ops.layer_norm_skip(seqs, ln2.weight, ln2.bias, ln2.eps)
            seqs = ops.dropout_add(self.feed_forward(g), skip, p)
            return ops.dropout(seqs, p)  # dropout_3 (net_blocks.py:260)
        seqs = mha.out_proj(a, residual=skip)
        g, skip = ops.layer_norm_skip(seqs, ln2.weight, ln2.bias, ln2.eps)
        return self.feed_forward(g, residual=skip)

    def native_ok(self) -> bool:
        ff = self.feed_forward
        return isinstance(ff, PointWiseFeedForward) and ff.activation == "gelu" and ff.ff_linear_1.bias is not None and \
            ff.ff_linear_2.bias is not None and self.multi_head_attn.out_proj.bias is not None

    def forward_packed_native(self, seqs, cu, B, window, causal, rows_real, planes=None):
        ff, mha, ln1, ln2 = self.feed_forward, self.multi_head_attn, self.layer_norm_1, self.layer_norm_2
        return ops.preln_layer_packed_train(
            seqs, cu, B, mha.n_heads, window, causal, self.p if self.training else 0.0, (ln1.weight, ln1.bias, ln1.eps),
            (mha.in_proj_weight, mha.in_proj_bias), (mha.out_proj.weight, mha.out_proj.bias), (ln2.weight, ln2.bias, ln2.eps),
            (ff.ff_linear_1.weight, ff.ff_linear_1.bias), (ff.ff_linear_2.weight, ff.ff_linear_2.bias), rows_real, planes)

    def forward_last_packed(self, seqs, cu, B, window, causal):
        """Inference: the block's output at the last row of every packed session, [B, d] (cf. `forward_last`): keys / values of every
        row, one query per session (`rt_mha_varlen_last_fwd`: the last query sees its whole session, causal or not)."""
        mha, d = self.multi_head_attn, seqs.shape[1]
        last_rows = cu[1:B + 1] - 1
        h = self.layer_norm_1(seqs)
        q = ops.linear(h.index_select(0, last_rows), mha.in_proj_weight[:d], mha.in_proj_bias[:d])
        if ops.mha_varlen_last_x_supported(d, mha.n_heads):      # the last query needs no key / value rows (round 6): one pass over h
            a = ops.mha_varlen_last_x(q, h, mha.in_proj_weight, mha.in_proj_bias, cu, B, mha.n_heads, window, False)
        else:
            kv = ops.linear(h, mha.in_proj_weight[d:], mha.in_proj_bias[d:])            # [Np, 2d]
            a = torch.empty((B, d), dtype=torch.float32, device=seqs.device)
            ops._c("rt_mha_varlen_last_fwd", q, d, kv, 2 * d, kv[:, d:], 2 * d, cu, None, None, B, mha.n_heads, d // mha.n_heads,   # pylint: disable=protected-access
                   window, window, a, d)
        x1 = mha.out_proj(a, residual=seqs.index_select(0, last_rows))
        return self.feed_forward(self.layer_norm_2(x1), residual=x1)


class PreLNTransformerLayers(TransformerLayersBase):
    def __init__(self, n_blocks: int, n_factors: int, n_heads: int, dropout_rate: float, ff_factors_multiplier: int = 4,
                 **kwargs: tp.Any) -> None:
        super().__init__()
        self.n_blocks = n_blocks
        self.transformer_blocks = nn.ModuleList(
            [PreLNTransformerLayer(n_factors, n_heads, dropout_rate, ff_factors_multiplier) for _ in range(n_blocks)])

    def forward(self, seqs, ids, B, L, causal, keypad, batch):
        for blk in self.transformer_blocks:
            seqs = blk(seqs, ids, B, L, causal, keypad)
        return seqs

    def forward_last(self, seqs, ids, B, L, causal, keypad, batch):
        """Inference: [B, d] encodings of the last position (lightning.py:393-397); the final block on one query row per session."""
        blocks = list(self.transformer_blocks)
        for blk in blocks[:-1]:
            seqs = blk(seqs, ids, B, L, causal, keypad)
        return blocks[-1].forward_last(seqs, ids, B, L, causal, keypad)

    def packed_ok(self, n_factors: int, window: int, causal: bool, keypad: bool = False) -> bool:
        """Packed rows serve this stack when the pad positions are masked as keys (BERT4Rec's default, bert4rec.py:331): nothing
        re-masks the rows between Pre-LN blocks (net_blocks.py:290-310), so without the key-padding mask the pad rows carry state
        that real queries read, and the padded window has to stay."""
        blocks = list(self.transformer_blocks)
        if not blocks or not keypad or any(b.generic for b in blocks):
            return False
        heads = blocks[0].multi_head_attn.n_heads
        return ops.mha_varlen_supported(heads, n_factors, window) if causal else ops.mha_bidir_supported(heads, n_factors, window)

    def forward_packed_train(self, seqs, cu, B, window, keypad, rows_real=None, causal=False):
        covers = rows_real is not None and int(rows_real) == int(seqs.shape[0])
        native = ops.native_block_enabled() and rows_real is not None and seqs.shape[0] % 128 == 0 and \
            all(blk.native_ok() for blk in self.transformer_blocks)
        planes = self._fresh_planes() if native else None
        for blk in self.transformer_blocks:
            if native:     # the block's launch sequence issued by the native executor (csrc/rt_block.hip: rt_preln_block_packed_*)
                seqs = blk.forward_packed_native(seqs, cu, B, window, causal, int(rows_real), planes)
            else:
                seqs = blk.forward_packed(seqs, cu, B, window, causal, covers)
        return seqs

    accepts_first_pre = "preln"      # (`TransformerTorchBackbone.encode_last_packed`: x and the first block's in_proj output from projected tables)

    def forward_last_packed(self, seqs, cu, B, window, keypad, rows_real=None, causal=False, first_pre=None):
        blocks = list(self.transformer_blocks)
        for i, blk in enumerate(blocks[:-1]):
            seqs = blk.forward_packed(seqs, cu, B, window, causal, qkv_in=first_pre(blk) if (i == 0 and first_pre is not None) else None)
        return blocks[-1].forward_last_packed(seqs, cu, B, window, causal)


class LiGRLayer(nn.Module):
    def __init__(self, n_factors: int, n_heads: int, dropout_rate: float, ff_factors_multiplier: int = 4,
                 bias_in_ff: bool = False, ff_activation: str = "swiglu") -> None:
        super().__init__()
        self.multi_head_attn = MultiheadAttnParams(n_factors, n_heads)
        self.layer_norm_1 = LayerNormParams(n_factors)
        self.layer_norm_2 = LayerNormParams(n_factors)
        self.feed_forward = init_feed_forward(n_factors, ff_factors_multiplier, dropout_rate, ff_activation, bias_in_ff)
        self.gating_linear_1 = LinearParams(n_factors, n_factors)
        self.gating_linear_2 = LinearParams(n_factors, n_factors)
        self.p = dropout_rate
        self.generic = False      # `DimPlan`: zero-padded columns (no packed rows)

    def forward(self, seqs, ids, B, L, causal, keypad):
        p = self.p if self.training else 0.0
        ln1, ln2 = self.layer_norm_1, self.layer_norm_2
        g1, g2 = self.gating_linear_1, self.gating_linear_2
        h, seqs = ops.layer_norm_skip(seqs, ln1.weight, ln1.bias, ln1.eps, ln1.cols)   # the skip branch's gradient rides in LN1's backward
        a = self.multi_head_attn(h, None, ids, B, L, causal, keypad, p)
        seqs = ops.gated_residual(seqs, g1.weight, g1.bias, a, p)    # seqs + sigmoid(Wg1 seqs + bg1) * drop(mha): one node
        g, seqs = ops.layer_norm_skip(seqs, ln2.weight, ln2.bias, ln2.eps, ln2.cols)
        f = self.feed_forward(g)
        return ops.gated_residual(seqs, g2.weight, g2.bias, f, p)    # seqs + sigmoid(Wg2 seqs + bg2) * drop(ffn)


    def forward_packed(self, seqs, B, window, causal, pad_idx, pad_ids, n_real, cu=None, n_prefixed=None, qkv_in=None):
        """The block over PACKED rows ([Np, d]: real positions + the unused tail of the row block) — exact under key-padding masks: no
        real query sees a pad key, and nothing else couples rows (ligr.py:66-106 is LayerNorm, Linear, gates, SwiGLU: row-wise).
        cu given (head size 32 / 64 / 128: the streamed packed kernels, K4v3): the attention runs on the packed in_proj output as it
        is.  Else the [B, L] window kernels: the packed in_proj output is scattered into the window (`pad_idx`; pad slots zero, masked
        as keys through `pad_ids`) and the attention output gathered back.  Every GEMM, LayerNorm, gate and dropout of the block runs
        on the packed rows only either way."""
        p = self.p if self.training else 0.0
        ln1, ln2, mha = self.layer_norm_1, self.layer_norm_2, self.multi_head_attn
        g1, g2 = self.gating_linear_1, self.gating_linear_2
        if qkv_in is not None:      # (inference, first block: the in_proj output from projected tables, `rt_embed_block1_preln_fwd`)
            qkv = qkv_in
        else:
            h, seqs = ops.layer_norm_skip(seqs, ln1.weight, ln1.bias, ln1.eps)
            qkv = ops.linear(h, mha.in_proj_weight, mha.in_proj_bias)
        if cu is not None:      # (n_prefixed: the sessions sit behind the shared pad prefix, `ops.mha_varlen_qkv`)
            a = ops.mha_varlen_qkv(qkv, cu, int(cu.numel()) - 1, mha.n_heads, window, causal, p,
                                   n_real is not None and int(n_real) == int(seqs.shape[0]), n_prefixed=n_prefixed)
        else:
            qkv_w = ops.scatter_rows(qkv, pad_idx, B * window + 1)
            a_w = ops.mha_packed(qkv_w[:B * window], pad_ids[:B * window], B, mha.n_heads, window, causal, True, p)
            a = ops.gather_rows(_with_dump_row(a_w), pad_idx)
        a = mha.out_proj(a)
        seqs = ops.gated_residual(seqs, g1.weight, g1.bias, a, p)
        g, seqs = ops.layer_norm_skip(seqs, ln2.weight, ln2.bias, ln2.eps)
        f = self.feed_forward(g)
        return ops.gated_residual(seqs, g2.weight, g2.bias, f, p)

    def forward_last_packed(self, seqs, cu, B, window, prefix_row=-1):
        """Inference over PACKED rows: the block's output at the last row of the B real sessions, [B, d] (cf. `forward_last`).  Only
        LayerNorm_1 sees every row: the last query's attention needs no key / value rows (`ops.mha_varlen_last_x`; prefix_row >= 0: the
        shared pad prefix of `LiGRLayers.packed_mode() == "prefix"` starts there), the out-projection, both gates, LayerNorm_2 and the
        feed-forward run on B rows."""
        mha, d = self.multi_head_attn, seqs.shape[1]
        last_rows = cu[1:B + 1] - 1
        h = self.layer_norm_1(seqs)
        q = ops.linear(h.index_select(0, last_rows), mha.in_proj_weight[:d], mha.in_proj_bias[:d])
        a = mha.out_proj(ops.mha_varlen_last_x(q, h, mha.in_proj_weight, mha.in_proj_bias, cu, B, mha.n_heads, window, False, prefix_row))
        x = seqs.index_select(0, last_rows)
        x = ops.gate(x, self.gating_linear_1(x), a, 0.0)
        f = self.feed_forward(self.layer_norm_2(x))
        return ops.gate(x, self.gating_linear_2(x), f, 0.0)

    def forward_last(self, seqs, ids, B, L, causal, keypad):
        """Inference: the block's output at the last position of every session, [B, d] (see PreLNTransformerLayer.forward_last)."""
        h = self.layer_norm_1(seqs)
        a = self.multi_head_attn.out_proj(_attend_last(self.multi_head_attn, h, _take_last(h, B, L), ids, B, L, causal, keypad))
        x = _take_last(seqs, B, L)
        x = ops.gate(x, self.gating_linear_1(x), a, 0.0)
        f = self.feed_forward(self.layer_norm_2(x))
        return ops.gate(x, self.gating_linear_2(x), f, 0.0)


class LiGRLayers(TransformerLayersBase):
    def __init__(self, n_blocks: int, n_factors: int, n_heads: int, dropout_rate: float, ff_factors_multiplier: int = 4,
                 ff_activation: str = "swiglu", bias_in_ff: bool = False, **kwargs: tp.Any) -> None:
        super().__init__()
        self.n_blocks = n_blocks
        self.transformer_blocks = nn.ModuleList(
            [LiGRLayer(n_factors, n_heads, dropout_rate, ff_factors_multiplier, bias_in_ff, ff_activation)
             for _ in range(n_blocks)])

    def forward(self, seqs, ids, B, L, causal, keypad, batch):
        with ops.active_planes(self._fresh_planes()):      # the blocks' Linear products read the pre-split weight planes (K7w)
            for blk in self.transformer_blocks:
                seqs = blk(seqs, ids, B, L, causal, keypad)
        return seqs

    def forward_last(self, seqs, ids, B, L, causal, keypad, batch):
        """Inference: [B, d] encodings of the last position; the final block on one query row per session."""
        blocks = list(self.transformer_blocks)
        with ops.active_planes(self._fresh_planes()):
            for blk in blocks[:-1]:
                seqs = blk(seqs, ids, B, L, causal, keypad)
            return blocks[-1].forward_last(seqs, ids, B, L, causal, keypad)

    def packed_mode(self, n_factors: int, window: int, causal: bool, keypad: bool = False) -> tp.Optional[str]:
        """How this stack runs on packed rows.  "rows": pad positions are masked as keys (`use_key_padding_mask=True`) — no real query
        sees a pad, and nothing else couples rows (ligr.py:66-106 is LayerNorm, Linear, gates, SwiGLU).  "prefix": no key-padding mask
        — the reference's default for SASRec-style models, eSASRec included: nothing re-zeroes the pad rows between blocks
        (ligr.py:161-191), they carry a state that real queries read; that state depends on the POSITION only (a pad row sees pad
        rows; every session's pads start from the same zero item row + positional row), so the batch carries the window's pad rows
        ONCE, as one more packed session, and every real session attends to its first window - n rows (`ops.mha_varlen_qkv`
        n_prefixed; causal attention on the streamed kernels only).  Exact without dropout; with dropout the pad rows' masks are
        shared by the sessions of a batch where the reference draws them per session — every session's own marginal distribution,
        hence the expected gradient, is the reference's.  None: the padded window."""
        if len(self.transformer_blocks) == 0 or any(b.generic for b in self.transformer_blocks):
            return None
        if keypad:
            return "rows"
        if causal and self._packed_attention_on_rows(window, True) and os.environ.get("RT_PACKED_PREFIX", "1") != "0":
            return "prefix"
        return None

    def packed_ok(self, n_factors: int, window: int, causal: bool, keypad: bool = False) -> bool:
        return self.packed_mode(n_factors, window, causal, keypad) is not None

    def _packed_attention_on_rows(self, window: int, causal: bool) -> bool:
        mha = self.transformer_blocks[0].multi_head_attn
        d = int(mha.in_proj_weight.shape[1])
        return ops.mha_varlen_supported(mha.n_heads, d, window) if causal else ops.mha_bidir_supported(mha.n_heads, d, window)

    def forward_packed_train(self, seqs, cu, B, window, keypad, rows_real=None, causal=True, n_prefixed=None):
        """n_prefixed: session number n_prefixed of `cu` is the shared pad prefix (mode "prefix"), sessions before it sit behind it."""
        if not keypad and n_prefixed is None:
            raise ValueError("a LiGR stack without key-padding masks packs only behind a shared pad prefix (packed_mode 'prefix'): n_prefixed")
        n_real = int(rows_real) if rows_real is not None else None
        on_rows = n_prefixed is not None or self._packed_attention_on_rows(window, causal)
        pad_idx, pad_ids = (None, None) if on_rows else ops.padded_index(cu, B, window, int(seqs.shape[0]))
        with ops.active_planes(self._fresh_planes()):
            for blk in self.transformer_blocks:
                seqs = blk.forward_packed(seqs, B, window, causal, pad_idx, pad_ids, n_real, cu if on_rows else None, n_prefixed)
        return seqs

    accepts_first_pre = "preln"

    def forward_last_packed(self, seqs, cu, B, window, keypad, rows_real=None, causal=True, n_prefixed=None, first_pre=None):
        """Inference over packed rows: all blocks on the packed rows, then the last row of every session (B: the real sessions)."""
        if not keypad and n_prefixed is None:
            raise ValueError("a LiGR stack without key-padding masks packs only behind a shared pad prefix (packed_mode 'prefix'): n_prefixed")
        on_rows = n_prefixed is not None or self._packed_attention_on_rows(window, causal)
        pad_idx, pad_ids = (None, None) if on_rows else ops.padded_index(cu, B, window, int(seqs.shape[0]))
        blocks = list(self.transformer_blocks)
        d, heads = int(seqs.shape[1]), blocks[-1].multi_head_attn.n_heads
        # the FINAL block on one query row per session (round 6): causal stacks only (the last query then sees exactly its session — and
        # the pad prefix in front of it), with the row count known on the host when the prefix's first row has to be named
        last_only = (on_rows and causal and not blocks[-1].generic and ops.mha_varlen_last_x_supported(d, heads)
                     and (n_prefixed is None or rows_real is not None))
        with ops.active_planes(self._fresh_planes()):
            for i, blk in enumerate(blocks[:-1] if last_only else blocks):
                seqs = blk.forward_packed(seqs, B, window, causal, pad_idx, pad_ids, None, cu if on_rows else None, n_prefixed,
                                          qkv_in=first_pre(blk) if (i == 0 and first_pre is not None and len(blocks) > 1) else None)
            if last_only:
                return blocks[-1].forward_last_packed(seqs, cu, B, window, -1 if n_prefixed is None else int(rows_real) - window)
        return seqs.index_select(0, cu[1:B + 1] - 1)


def _with_dump_row(x: torch.Tensor) -> torch.Tensor:
    """[n, d] -> [n + 1, d] with a zero row behind it: the slot the unused tail rows of a packed block gather from."""
    return torch.nn.functional.pad(x, (0, 0, 0, 1))


class RelativeAttentionBias(nn.Module):
    """Parameter holder for the HSTU relative bias (hstu.py:47-82); the bias itself is computed in-kernel."""

    def __init__(self, session_max_len: int, relative_time_attention: bool, relative_pos_attention: bool,
                 num_buckets: int = 128) -> None:
        super().__init__()
        self.num_buckets = num_buckets      # any value: the kernels search the unclamped bucket, `ops.hstu_time_thresholds` carries the clamp
        if relative_time_attention:
            self.time_weights = nn.Parameter(torch.empty(num_buckets + 1).normal_(mean=0, std=0.02))
        if relative_pos_attention:
            self.pos_weights = nn.Parameter(torch.empty(2 * session_max_len - 1).normal_(mean=0, std=0.02))
        self.relative_time_attention = relative_time_attention
        self.relative_pos_attention = relative_pos_attention


class STULayer(nn.Module):
    def __init__(self, n_factors: int, n_heads: int, linear_hidden_dim: int, attention_dim: int, session_max_len: int,
                 relative_time_attention: bool, relative_pos_attention: bool, attn_dropout_rate: float, dropout_rate: float,
                 epsilon: float, num_buckets: int = 128) -> None:
        super().__init__()
        # linear_hidden_dim != attention_dim: the kernels run ONE head size for u / v and q / k.  Such a block is a parameter holder only
        # (the real-size twin of a `DimPlan` model: models._build_model_from_dataset pads both head sizes to one the kernels tile)
        self.lin = linear_hidden_dim
        self.generic = False      # `DimPlan`: zero-padded columns -> the block runs out of the individual ops
        self.rel_attn = RelativeAttentionBias(session_max_len, relative_time_attention, relative_pos_attention, num_buckets)
        self.n_heads, self.hd, self.L = n_heads, attention_dim, session_max_len
        self.uvqk_proj = nn.Parameter(torch.empty(n_factors, 2 * n_heads * linear_hidden_dim + 2 * n_heads * attention_dim))
        # the reference leaves this parameter uninitialised until xavier (hstu.py:207-214: torch.empty) and so draws nothing here; a private
        # generator keeps the global stream where the reference's is (same seed -> same 1-D parameters as the reference's model)
        nn.init.normal_(self.uvqk_proj, std=0.02, generator=torch.Generator().manual_seed(0))
        self.output_mlp = LinearParams(n_heads * linear_hidden_dim, n_factors)
        self.norm_input = LayerNormParams(n_factors, eps=epsilon)
        self.norm_attn_output = LayerNormParams(n_heads * linear_hidden_dim, eps=epsilon)
        self.p_mlp, self.p_attn = dropout_rate, attn_dropout_rate

    def forward(self, seqs, ids, B, L, batch, thr):
        """`seqs` comes in unmasked: the row mask (hstu.py:256) is the first step of the fused block."""
        if self.lin != self.hd:
            raise NotImplementedError("the HIP HSTU kernels run one head size for u / v and q / k: linear_hidden_dim != attention_dim runs "
                                      "through `nn.DimPlan` (models.HSTUModel / transformer_layers_type=STULayers build it)")
        if self.output_mlp.bias is not None and not self.generic:
            tw = self.rel_attn.time_weights if self.rel_attn.relative_time_attention else None
            pw = self.rel_attn.pos_weights if self.rel_attn.relative_pos_attention else None
            return ops.stu_layer(seqs, ids, batch.get("unix_ts") if tw is not None else None, thr, B, L, self.n_heads, self.hd,
                                 self.p_attn if self.training else 0.0, self.p_mlp if self.training else 0.0,
                                 (self.norm_input.weight, self.norm_input.bias, self.norm_input.eps), self.uvqk_proj, tw, pw,
                                 (self.norm_attn_output.weight, self.norm_attn_output.bias, self.norm_attn_output.eps),
                                 (self.output_mlp.weight, self.output_mlp.bias))
        return self.forward_modular(ops.mul_mask(seqs, None, ids), ids, B, L, batch, thr)

    def forward_last(self, seqs, ids, B, L, batch, thr):
        """Inference: the block's output at the last position of every session, [B, d].  v and k are projected for every position
        (two column blocks of `uvqk_proj`, read in place through a row stride), u and q for the last row only; the attention of
        that row is `rt_hstu_attn_last_fwd`."""
        if self.generic:
            return _take_last(self(seqs, ids, B, L, batch, thr), B, L)
        hh, d = self.n_heads * self.hd, seqs.shape[1]
        M = seqs.shape[0]
        dev = seqs.device
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
        ids_flat = ids.reshape(-1)
        ids_last = ids.view(B, L)[:, L - 1].contiguous()
        x0 = ops.mul_mask(seqs, None, ids_flat)
        normed = ops.mul_mask(self.norm_input(x0), None, ids_flat)
        P = self.uvqk_proj                                                        # [d, 4 hh]: u | v | q | k column blocks
        vk = new(2, M, hh)
        for i, c0 in enumerate((hh, 3 * hh)):                                     # v, k of every position
            ops._gemm(normed, d, 1, P[:, c0:], 4 * hh, 0, vk[i], hh, None, None, 0, M, hh, d)   # pylint: disable=protected-access
        vk = ops.act_dropout(vk.view(2 * M, hh), ops.ACT_SILU, 0.0).view(2, M, hh)
        n_last = _take_last(normed, B, L)
        uq = new(2, B, hh)
        for i, c0 in enumerate((0, 2 * hh)):                                      # u, q of the last position
            ops._gemm(n_last, d, 1, P[:, c0:], 4 * hh, 0, uq[i], hh, None, None, 0, B, hh, d)   # pylint: disable=protected-access
        uq = ops.act_dropout(uq.view(2 * B, hh), ops.ACT_SILU, 0.0).view(2, B, hh)
        tw = self.rel_attn.time_weights if self.rel_attn.relative_time_attention else None
        pw = self.rel_attn.pos_weights if self.rel_attn.relative_pos_attention else None
        attn = new(B, hh)
        ops._c("rt_hstu_attn_last_fwd", uq[1], hh, vk[1], hh, vk[0], hh, ids_flat, batch.get("unix_ts") if tw is not None else None,   # pylint: disable=protected-access
               tw, thr if tw is not None else None, pw, B, self.n_heads, L, self.hd, attn, hh)
        o_in = ops.mul_mask(uq[0], self.norm_attn_output(attn), ids_last)
        return self.output_mlp(o_in, residual=_take_last(x0, B, L))

    def forward_last_packed(self, seqs, cu, B, window, ts, thr):
        """Inference over PACKED sessions: the block's output at the last row of every session, [B, d] (cf. `forward_last`): v and k are
        projected for every row, u and q for the last rows only, the attention of that row is `rt_hstu_attn_varlen_last_fwd`;
        LayerNorm(attn), the gate and the output MLP run on B rows."""
        hh, d = self.n_heads * self.hd, seqs.shape[1]
        M = seqs.shape[0]
        new = lambda *s_: torch.empty(s_, dtype=torch.float32, device=seqs.device)  # noqa: E731
        last_rows = cu[1:B + 1] - 1
        normed = self.norm_input(seqs)
        P = self.uvqk_proj                                                        # [d, 4 hh]: u | v | q | k column blocks
        vk = new(2, M, hh)
        for i, c0 in enumerate((hh, 3 * hh)):                                     # v, k of every row
            ops._gemm(normed, d, 1, P[:, c0:], 4 * hh, 0, vk[i], hh, None, None, 0, M, hh, d)   # pylint: disable=protected-access
        vk = ops.act_dropout(vk.view(2 * M, hh), ops.ACT_SILU, 0.0).view(2, M, hh)
        n_last = normed.index_select(0, last_rows)
        uq = new(2, B, hh)
        for i, c0 in enumerate((0, 2 * hh)):                                      # u, q of the last rows
            ops._gemm(n_last, d, 1, P[:, c0:], 4 * hh, 0, uq[i], hh, None, None, 0, B, hh, d)   # pylint: disable=protected-access
        uq = ops.act_dropout(uq.view(2 * B, hh), ops.ACT_SILU, 0.0).view(2, B, hh)
        tw = self.rel_attn.time_weights if self.rel_attn.relative_time_attention else None
        pw = self.rel_attn.pos_weights if self.rel_attn.relative_pos_attention else None
        attn = new(B, hh)
        ops._c("rt_hstu_attn_varlen_last_fwd", uq[1], hh, vk[1], hh, vk[0], hh, cu, ts if tw is not None else None, tw,   # pylint: disable=protected-access
               thr if tw is not None else None, pw, B, self.n_heads, window, self.hd, window, attn, hh)
        o_in = ops.mul_mask(uq[0], self.norm_attn_output(attn), None)
        return self.output_mlp(o_in, residual=seqs.index_select(0, last_rows))

    def forward_packed(self, seqs, cu, B, window, ts, thr, rows_real=None):
        """The block over PACKED sessions ([Np, d], real positions only).  The reference zeroes pad rows before the bias-free projection
        (hstu.py:256-262): their u / v / q / k are silu(0) = 0, a pad key adds nothing to any query — the packed rows see exactly what
        they see in the padded window.  With the row count known on the host: ONE autograd node (`ops.stu_layer_packed`); else the ops
        of `forward_modular` without the masks, attention = `ops.hstu_attn_varlen`."""
        if rows_real is not None and self.output_mlp.bias is not None:
            tw = self.rel_attn.time_weights if self.rel_attn.relative_time_attention else None
            pw = self.rel_attn.pos_weights if self.rel_attn.relative_pos_attention else None
            return ops.stu_layer_packed(seqs, cu, rows_real, ts if tw is not None else None, thr, B, window, self.n_heads, self.hd,
                                        self.p_attn if self.training else 0.0, self.p_mlp if self.training else 0.0,
                                        (self.norm_input.weight, self.norm_input.bias, self.norm_input.eps), self.uvqk_proj, tw, pw,
                                        (self.norm_attn_output.weight, self.norm_attn_output.bias, self.norm_attn_output.eps),
                                        (self.output_mlp.weight, self.output_mlp.bias))
        hh = self.n_heads * self.hd
        normed = self.norm_input(seqs)
        uvqk = ops.act_dropout(ops.matmul_nn(normed, self.uvqk_proj), ops.ACT_SILU, 0.0)
        u, v, q, k = uvqk[:, :hh], uvqk[:, hh:2 * hh], uvqk[:, 2 * hh:3 * hh], uvqk[:, 3 * hh:]
        tw = self.rel_attn.time_weights if self.rel_attn.relative_time_attention else None
        pw = self.rel_attn.pos_weights if self.rel_attn.relative_pos_attention else None
        attn = ops.hstu_attn_varlen(q, k, v, tw, pw, cu, ts if tw is not None else None, thr, B, self.n_heads, window)
        attn = ops.dropout(attn, self.p_attn if self.training else 0.0)
        o_in = ops.mul_mask(u, self.norm_attn_output(attn), None)         # u * LN(attn)
        o_in = ops.dropout(o_in, self.p_mlp if self.training else 0.0)
        return self.output_mlp(o_in, residual=seqs)

    def forward_modular(self, seqs, ids, B, L, batch, thr):
        """Same block out of the individual autograd ops (`seqs` already masked); the cross-check of the fused node."""
        hh = self.n_heads * self.hd
        normed = ops.mul_mask(self.norm_input(seqs), None, ids)
        uvqk = ops.act_dropout(ops.matmul_nn(normed, self.uvqk_proj), ops.ACT_SILU, 0.0)
        u, v, q, k = uvqk[:, :hh], uvqk[:, hh:2 * hh], uvqk[:, 2 * hh:3 * hh], uvqk[:, 3 * hh:]
        tw = self.rel_attn.time_weights if self.rel_attn.relative_time_attention else None
        pw = self.rel_attn.pos_weights if self.rel_attn.relative_pos_attention else None
        attn = ops.hstu_attn(q, k, v, tw, pw, ids, batch.get("unix_ts") if tw is not None else None, thr, B, self.n_heads, L)
        attn = ops.dropout(attn, self.p_attn if self.training else 0.0)
        o_in = ops.mul_mask(u, self.norm_attn_output(attn), ids)          # u * LN(attn) * mask
        o_in = ops.dropout(o_in, self.p_mlp if self.training else 0.0)
        return self.output_mlp(o_in, residual=seqs)


class STULayers(TransformerLayersBase):
    def __init__(self, n_blocks: int, n_factors: int, n_heads: int, linear_hidden_dim: int, attention_dim: int,
                 session_max_len: int, relative_time_attention: bool, relative_pos_attention: bool,
                 attn_dropout_rate: float = 0.0, dropout_rate: float = 0.2, epsilon: float = 1e-6, num_buckets: int = 128,
                 **kwargs: tp.Any) -> None:
        """num_buckets: `RelativeAttentionBias`'s (hstu.py:63-71; the reference's STULayer always builds it with the default 128)."""
        super().__init__()
        self.n_blocks = n_blocks
        self.stu_blocks = nn.ModuleList([
            STULayer(n_factors, n_heads, linear_hidden_dim, attention_dim, session_max_len, relative_time_attention,
                     relative_pos_attention, attn_dropout_rate, dropout_rate, epsilon, num_buckets) for _ in range(n_blocks)])
        nb = self.stu_blocks[0].rel_attn.num_buckets if n_blocks > 0 else 128
        self.register_buffer("time_thr", ops.hstu_time_thresholds(nb), persistent=False)

    def forward(self, seqs, ids, B, L, causal, keypad, batch):
        with ops.active_planes(self._fresh_planes()):      # uvqk_proj / output_mlp products read the pre-split weight planes (K7w)
            for blk in self.stu_blocks:
                seqs = blk(seqs, ids, B, L, batch, self.time_thr)   # seqs * mask happens inside the block
        return ops.mul_mask(seqs, None, ids)

    def forward_last(self, seqs, ids, B, L, causal, keypad, batch):
        """Inference: [B, d] encodings of the last position; the final STU block on one query row per session."""
        blocks = list(self.stu_blocks)
        with ops.active_planes(self._fresh_planes()):
            for blk in blocks[:-1]:
                seqs = blk(seqs, ids, B, L, batch, self.time_thr)
            last = blocks[-1].forward_last(seqs, ids, B, L, batch, self.time_thr)
        return ops.mul_mask(last, None, ids.view(B, L)[:, L - 1].contiguous())

    def packed_ok(self, n_factors: int, window: int, causal: bool, keypad: bool = False) -> bool:
        """Packed rows serve the STU stack as it is: pad rows are zeroed before every bias-free projection, so they add nothing to a
        real row (`STULayer.forward_packed`)."""
        blocks = list(self.stu_blocks)
        return bool(blocks) and not any(b.generic for b in blocks) and ops.hstu_varlen_supported(blocks[0].n_heads, blocks[0].hd, window) \
            and window == blocks[0].L

    def forward_packed_train(self, seqs, cu, B, window, keypad, rows_real=None, causal=True, ts=None):
        with ops.active_planes(self._fresh_planes()):
            for blk in self.stu_blocks:
                seqs = blk.forward_packed(seqs, cu, B, window, ts, self.time_thr, rows_real)
        return seqs

    def forward_last_packed(self, seqs, cu, B, window, keypad, rows_real=None, causal=True, ts=None):
        """[B, d]: the packed blocks on every row — the FINAL block on one query row per session (`STULayer.forward_last_packed`:
        only its v / k projection and LayerNorm see every row)."""
        blocks = list(self.stu_blocks)
        if not blocks or blocks[-1].generic or blocks[-1].lin != blocks[-1].hd:
            return self.forward_packed_train(seqs, cu, B, window, keypad, ts=ts).index_select(0, cu[1:B + 1] - 1)
        with ops.active_planes(self._fresh_planes()):
            for blk in blocks[:-1]:
                seqs = blk.forward_packed(seqs, cu, B, window, ts, self.time_thr)
            return blocks[-1].forward_last_packed(seqs, cu, B, window, ts, self.time_thr)


# ---- similarity + backbone ------------------------------------------------------------------------------
class DistanceSimilarityModule(nn.Module):
    dist_available = [Distance.DOT, Distance.COSINE]

    def __init__(self, distance: str = "dot", **kwargs: tp.Any) -> None:
        super().__init__()
        if distance not in self.dist_available:
            raise ValueError("`dist` can only be either `dot` or `cosine`.")  # similarity.py:79-80
        self.distance = Distance(distance)

    def session_tower_forward(self, session_embs: torch.Tensor) -> torch.Tensor:
        return session_embs

    def item_tower_forward(self, item_embs: torch.Tensor) -> torch.Tensor:
        return item_embs

    # The reference's own methods (similarity.py:84-140).  The engine's stock training step and recommend() never call them — the
    # logits live inside the fused loss kernels (`rt_sampled_loss_*`, `rt_gemm` + `rt_softmax_ce_rows`) and the ranker
    # (`rt_topk_score*`) — but they ARE the seam: a subclass that overrides any of them is called exactly where the reference calls it
    # (`similarity_is_stock`, lightning.TransformerLossModule._loss_via_similarity, models.TransformerModelBase.recommend), and a
    # callback that asks the stock module for logits (examples/tutorials/utils.py:87-91) gets them.
    def _get_full_catalog_logits(self, session_embs: torch.Tensor, item_embs: torch.Tensor) -> torch.Tensor:
        lead = session_embs.shape[:-1]
        s2 = session_embs.reshape(-1, session_embs.shape[-1])
        if s2.is_cuda and s2.dtype == torch.float32 and item_embs.is_contiguous():
            return ops.linear(s2.contiguous(), item_embs).view(*lead, item_embs.shape[0])       # session_embs @ item_embs.T (rt_gemm)
        return session_embs @ item_embs.T

    def _get_pos_neg_logits(self, session_embs: torch.Tensor, item_embs: torch.Tensor, candidate_item_ids: torch.Tensor) -> torch.Tensor:
        pos_neg_embs = item_embs[candidate_item_ids]                       # [..., 1 + N, d]
        return (pos_neg_embs * session_embs.unsqueeze(-2)).sum(-1)         # the same dot products as similarity.py:94, no [.., d, 1] matmul

    def _get_embeddings_norm(self, embeddings: torch.Tensor) -> torch.Tensor:
        if embeddings.is_cuda and embeddings.dtype == torch.float32:
            return ops.l2norm(embeddings.reshape(-1, embeddings.shape[-1]).contiguous()).view(embeddings.shape)   # x / max(|x|, 1e-8)
        norm = torch.norm(embeddings, p=2, dim=-1, keepdim=True)
        return embeddings / torch.clamp(norm, min=1e-8)

    def forward(self, session_embs: torch.Tensor, item_embs: torch.Tensor,
                candidate_item_ids: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
        """Logits of the sessions against the whole catalog ([..., V]) or against `candidate_item_ids` ([..., C] -> [..., C])
        (similarity.py:100-113)."""
        if self.distance == Distance.COSINE:
            session_embs = self._get_embeddings_norm(session_embs)
            item_embs = self._get_embeddings_norm(item_embs)
        if candidate_item_ids is None:
            return self._get_full_catalog_logits(session_embs, item_embs)
        return self._get_pos_neg_logits(session_embs, item_embs, candidate_item_ids)

    def _recommend_u2i(self, user_embs: torch.Tensor, item_embs: torch.Tensor, user_ids: np.ndarray, k: int,
                       sorted_item_ids_to_recommend: np.ndarray, ui_csr_for_filter: tp.Any) -> tp.Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """similarity.py:115-140 on the HIP ranker: -> (user ids, item ids, scores), the reference's triplet."""
        from .rank import HipRanker

        ranker = HipRanker(self.distance, item_embs.device, user_embs[torch.as_tensor(user_ids, device=user_embs.device)], item_embs)
        idx, reco, scores = ranker.rank(np.arange(len(user_ids)), k=k, filter_pairs_csr=ui_csr_for_filter,
                                        sorted_object_whitelist=sorted_item_ids_to_recommend)
        return np.asarray(user_ids)[idx], reco, scores


_SIMILARITY_SEAM = ("forward", "session_tower_forward", "item_tower_forward", "_get_full_catalog_logits", "_get_pos_neg_logits",
                    "_get_embeddings_norm", "_recommend_u2i")


def similarity_is_stock(sim: tp.Any) -> bool:
    """True when `sim` computes what the fused kernels compute: the stock `DistanceSimilarityModule`, or a subclass that overrides none
    of the methods the reference calls (similarity.py:26-64) and owns no parameters.  Anything else — a learned temperature, projection
    towers, another ranker — is CALLED, not approximated: training logits, validation outputs and recommend() go through its methods."""
    if not isinstance(sim, DistanceSimilarityModule):
        return False
    cls = type(sim)
    if any(getattr(cls, name, None) is not getattr(DistanceSimilarityModule, name) for name in _SIMILARITY_SEAM):
        return False
    return next(iter(sim.parameters()), None) is None


def pack_last_items(offsets: torch.Tensor, items: torch.Tensor, rows: torch.Tensor, window: int) -> tp.Tuple[torch.Tensor, ...]:
    """The last `window` items of the sessions `rows` of a CSR store as ONE packed block (plain tensor ops, any device):
    -> cu [B+1] (session b = packed rows cu[b] .. cu[b+1]-1, oldest first), ids [N], dist [N] = distance of a row from its
    session's end (the index of its positional row: inverse positions, net_blocks.py:388-399)."""
    dev = offsets.device
    B = int(rows.numel())
    lens = torch.clamp(offsets[rows + 1] - offsets[rows], max=window)
    cu = torch.zeros((B + 1,), dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=cu[1:])
    N = int(cu[-1])
    sess = torch.repeat_interleave(torch.arange(B, dtype=torch.int64, device=dev), lens, output_size=N)
    j = torch.arange(N, dtype=torch.int64, device=dev) - cu[sess]
    ids = items[(offsets[rows + 1] - lens)[sess] + j]
    return cu, ids, lens[sess] - 1 - j


def pack_train_items(offsets: torch.Tensor, items: torch.Tensor, weights: torch.Tensor, rows: torch.Tensor,
                     window: int) -> tp.Tuple[torch.Tensor, ...]:
    """The SASRec training batch of the sessions `rows` (sasrec.py:86-104: the last window + 1 items; x = all but the last, y = all
    but the first, yw = the weights of y) as ONE packed block without padding rows: -> cu [B+1], x [N], y [N], yw [N], dist [N]
    (distance of a row from its session's last INPUT position = the index of its positional row).  Sessions with fewer than two
    items contribute no rows (the reference drops them at fit time, `train_min_user_interactions`)."""
    dev = offsets.device
    B = int(rows.numel())
    lens = torch.clamp(offsets[rows + 1] - offsets[rows] - 1, min=0, max=window)          # input positions per session
    cu = torch.zeros((B + 1,), dtype=torch.int64, device=dev)
    torch.cumsum(lens, 0, out=cu[1:])
    N = int(cu[-1])
    sess = torch.repeat_interleave(torch.arange(B, dtype=torch.int64, device=dev), lens, output_size=N)
    j = torch.arange(N, dtype=torch.int64, device=dev) - cu[sess]
    src = (offsets[rows + 1] - 1 - lens)[sess] + j                                       # x = tail[:-1]
    return cu, items[src], items[src + 1], weights[src + 1], lens[sess] - 1 - j


class TransformerTorchBackbone(nn.Module):
    """encode_sessions / training loss of the reference backbone (torch_backbone.py:118-286) on the HIP ops."""

    def __init__(self, n_heads: int, dropout_rate: float, item_model: SumOfEmbeddingsConstructor,
                 pos_encoding_layer: LearnableInversePositionalEncoding, transformer_layers: TransformerLayersBase,
                 similarity_module: DistanceSimilarityModule, use_causal_attn: bool = True,
                 use_key_padding_mask: bool = False, **kwargs: tp.Any) -> None:
        super().__init__()
        self.item_model = item_model
        self.pos_encoding_layer = pos_encoding_layer
        self.transformer_layers = transformer_layers
        self.similarity_module = similarity_module
        self.use_causal_attn = use_causal_attn
        # called (no arguments) when a backward pass has produced every gradient BEHIND the blocks' input and is about to run the
        # lookup's backward: a data-parallel loop starts the exchange of the block weights' gradients there (`lightning.FlatAdam.
        # begin_early_exchange`).  A plain attribute: not a parameter, not in the state_dict.
        self.on_input_gradient: tp.Optional[tp.Callable[[], None]] = None
        self.use_key_padding_mask = use_key_padding_mask
        self.n_heads = n_heads
        self.dropout_rate = dropout_rate
        self.d_real: tp.Optional[int] = None     # `DimPlan`: n_factors of the model when the tables carry zero columns behind it

    def _pos_scale(self, table: torch.Tensor) -> float:
        """sqrt(n_factors) of `use_scale_factor` (net_blocks.py:390-391) — of the model's width, not of a zero-padded table's."""
        if not self.pos_encoding_layer.use_scale_factor:
            return 1.0
        return float(self.d_real if self.d_real is not None else table.shape[1]) ** 0.5

    def _fused_pos(self) -> bool:
        """The stock positional encoding (scale + inverse learnable rows) is fused into `rt_embed_fwd`; any other class plugged
        through `pos_encoding_type` (transformers/base.py:407-413) — a subclass that overrides `forward` included — is CALLED
        on the [B, L, d] item embeddings, as the reference does (torch_backbone.py:245-246)."""
        pe = self.pos_encoding_layer
        return isinstance(pe, LearnableInversePositionalEncoding) and type(pe).forward is LearnableInversePositionalEncoding.forward

    def _watch_input_gradient(self, seqs: torch.Tensor) -> torch.Tensor:
        if self.on_input_gradient is not None and seqs.requires_grad:
            notify = self.on_input_gradient

            def hook(_g: torch.Tensor) -> None:      # (returns None: the gradient passes unchanged)
                notify()

            seqs.register_hook(hook)
        return seqs

    def _embed_sessions(self, table: torch.Tensor, ids: torch.Tensor, B: int, L: int, p: float) -> torch.Tensor:
        """[B*L, d] = dropout(pos_encoding(item embeddings)) (torch_backbone.py:245-247)."""
        d = table.shape[1]
        if self._fused_pos():
            pos = self.pos_encoding_layer.pos_emb.weight if self.pos_encoding_layer.pos_emb is not None else None
            return ops.embed(table, pos, ids, L, self._pos_scale(table), p)
        seqs = ops.embed(table, None, ids, L, 1.0, 0.0).view(B, L, d)
        seqs = self.pos_encoding_layer(seqs).reshape(B * L, d)
        return ops.dropout(seqs, p) if p > 0 else seqs

    def encode_sessions(self, batch: Batch, item_embs: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
        """-> [B, L, d] session encodings (torch_backbone.py:220-260)."""
        x = batch["x"]
        B, L = x.shape
        table = self.item_model.table if item_embs is None else item_embs
        d = table.shape[1]
        ids = x.reshape(-1)
        seqs = self._watch_input_gradient(self._embed_sessions(table, ids, B, L, self.dropout_rate if self.training else 0.0))
        seqs = self.transformer_layers(seqs, ids, B, L, self.use_causal_attn, self.use_key_padding_mask, batch)
        return seqs.view(B, L, d)

    def can_encode_packed(self, n_factors: int, window: int) -> bool:
        """Does `encode_last_packed` serve this backbone (inference, a layer stack with a packed forward, see its `packed_ok`)?"""
        ok = getattr(self.transformer_layers, "packed_ok", None)
        # a plugged backbone that overrides the reference-shaped encoders (and brings no packed twin) keeps the path its override runs on
        cls, base = type(self), TransformerTorchBackbone
        stock = (cls.encode_sessions is base.encode_sessions and cls.encode_last is base.encode_last) \
            or cls.encode_last_packed is not base.encode_last_packed
        return ok is not None and stock and self._fused_pos() and not self.training and not torch.is_grad_enabled() \
            and ok(n_factors, window, self.use_causal_attn, self.use_key_padding_mask)

    def encode_last_packed(self, offsets: torch.Tensor, items: torch.Tensor, rows: torch.Tensor, window: int,
                           item_embs: tp.Optional[torch.Tensor] = None, cu: tp.Optional[torch.Tensor] = None,
                           n_rows: tp.Optional[int] = None, mask_id: tp.Optional[int] = None,
                           ts_store: tp.Optional[torch.Tensor] = None, ts_ctx: tp.Optional[torch.Tensor] = None,
                           cache: tp.Optional[tp.Dict[str, tp.Any]] = None) -> torch.Tensor:
        """-> [B, d] = `encode_last` of the sessions `rows` of a CSR session store (`offsets`, `items`: model item ids, oldest
        first), without ever building the padded [B, L] batch: the last `window` items of every session are gathered into ONE
        packed row block (embedding + positional row by distance from the session's end, torch_backbone.py:245-246 /
        net_blocks.py:388-399 with inverse positions) and the stack runs on those rows only.  Every session must hold at least
        one item.  `cu` [B+1] / `n_rows` = cu[-1]: the packed row offsets cut by the caller on the HOST (no device round trip);
        without them they are taken from the device offsets (one synchronisation).  mask_id: the BERT4Rec batch instead — the last
        window - 1 items and the MASK token as the last row of every session (bert4rec.py:182-193).  ts_store / ts_ctx: the store's
        timestamps (seconds) and the request time of every session — the packed timestamps a relative time bias reads.
        cache: a dict the caller keeps for ONE recommend() call (weights fixed): the projected tables of the first block's keys /
        values live there between the call's encoder launches."""
        table = self.item_model.table if item_embs is None else item_embs
        d = table.shape[1]
        B = int(rows.numel())
        if cu is None or n_rows is None:
            lens = torch.clamp(offsets[rows + 1] - offsets[rows], max=window)
            if mask_id is not None:
                lens = torch.clamp(offsets[rows + 1] - offsets[rows], max=window - 1) + 1
            cu = torch.zeros((B + 1,), dtype=torch.int64, device=offsets.device)
            torch.cumsum(lens, 0, out=cu[1:])
            n_rows = int(cu[-1])
        mode = getattr(self.transformer_layers, "packed_mode", None)
        prefix = mask_id is None and mode is not None and \
            mode(int(table.shape[1]), window, self.use_causal_attn, self.use_key_padding_mask) == "prefix"
        Np = (n_rows + (window if prefix else 0) + 127) // 128 * 128
        if mask_id is None:
            ids, dist = ops.collate_packed(offsets, items, None, rows, cu, Np, train=False)
        else:
            ids, dist = ops.collate_packed_bert(offsets, items, None, rows, cu, Np, window, False, mask_id)
        if prefix:      # the window's pad rows once, behind the sessions: ids 0 (the rows behind n_rows already are), positions window - 1 .. 0
            dist[n_rows:n_rows + window] = torch.arange(window - 1, -1, -1, dtype=dist.dtype, device=dist.device)
            cu = torch.cat([cu, cu[-1:] + window])
        pos = self.pos_encoding_layer.pos_emb.weight if self.pos_encoding_layer.pos_emb is not None else None
        scale = self._pos_scale(table)
        x = torch.empty((Np, d), dtype=torch.float32, device=table.device)
        kw = {}
        skip_embed = False
        if ts_store is not None:     # a stack with a relative time bias (HSTU): the sessions' timestamps + the request's, packed
            kw["ts"] = ops.collate_packed_ts(offsets, ts_store, rows, cu, n_rows, ctx=ts_ctx)
        if prefix:
            kw["n_prefixed"] = B
            n_rows = n_rows + window
        layers = self.transformer_layers
        kind = getattr(layers, "accepts_first_pre", None)
        n_blk = len(getattr(layers, "transformer_blocks", ()))
        if (cache is not None and kind == "preln" and pos is not None and n_blk > 1 and d <= 512 and not getattr(layers.transformer_blocks[0], "generic", False)
                and ("qkv_tables" in cache or Np >= int(table.shape[0]))):
            # A Pre-LN / LiGR first block reads LN1(x) for queries, keys AND values: all of its in_proj output is the affine gather
            # (see the SASRec form below); the skip branch wants x itself, which the same kernel writes.
            def first_pre_ln(blk: nn.Module) -> torch.Tensor:
                mha, ln = blk.multi_head_attn, blk.layer_norm_1
                in_w, in_b = mha.in_proj_weight, mha.in_proj_bias
                tabs = cache.get("qkv_tables")
                if tabs is None:
                    g = ln.weight
                    tabs = cache["qkv_tables"] = (ops.linear((table * g).contiguous(), in_w, None), ops.linear((pos * g).contiguous(), in_w, None),
                                                  torch.mv(in_w, g).contiguous(), (torch.mv(in_w, ln.bias) + in_b).contiguous())
                qkv = torch.empty((Np, 3 * d), dtype=torch.float32, device=table.device)
                ops._c("rt_embed_block1_preln_fwd", ids, dist, table, pos, float(scale), float(ln.eps), tabs[0], tabs[1], tabs[2], tabs[3], Np, d, x, qkv)
                return qkv

            kw["first_pre"] = first_pre_ln
            skip_embed = True
        if (cache is not None and kind == "sasrec" and mask_id is None and not prefix and pos is not None
                and n_blk > 1 and ("kv_tables" in cache or Np >= int(table.shape[0]))):
            # The first block's inputs without a product over the rows: x = scale * E[id] + P[dist], the key / value projection is linear in
            # it and the query projection of LN1(x) is linear in it once the row's mean / rstd are known — K | V = scale * (E W_kv^T)[id] +
            # (P W_kv^T + b_kv)[dist], Q = rstd * (scale * (E G W_q^T)[id] + (P G W_q^T)[dist] - mean * W_q g) + (W_q beta + b_q): six small
            # products ONCE per recommend() call, then ONE gather kernel (`rt_embed_block1_fwd`) instead of the embedding kernel, LayerNorm_1
            # and the block's [rows, d] x [d, 3d] projection.  Worth it when the call's rows outnumber the catalog (a whole-catalog request:
            # 15.6 M rows, 26,744 items).
            def first_pre(blk: nn.Module) -> tp.Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
                mha, ln = blk.multi_head_attn, blk.q_layer_norm
                in_w, in_b = mha.in_proj_weight, mha.in_proj_bias
                tabs = cache.get("kv_tables")
                if tabs is None:      # once per recommend() call: six small products over the catalog and the positions
                    g = ln.weight
                    tabs = cache["kv_tables"] = (
                        ops.linear((table * g).contiguous(), in_w[:d], None), ops.linear((pos * g).contiguous(), in_w[:d], None),
                        torch.mv(in_w[:d], g).contiguous(), (torch.mv(in_w[:d], ln.bias) + in_b[:d]).contiguous(),
                        ops.linear(table.contiguous(), in_w[d:], None), ops.linear(pos.contiguous(), in_w[d:], in_b[d:]))
                qn, Q = x, torch.empty((Np, d), dtype=torch.float32, device=table.device)      # (x's buffer takes LN1(x): x itself is never written)
                kv = torch.empty((Np, 2 * d), dtype=torch.float32, device=table.device)
                ops._c("rt_embed_block1_fwd", ids, dist, table, pos, float(scale), ln.weight, ln.bias, float(ln.eps), tabs[0], tabs[1], tabs[2], tabs[3],
                       tabs[4], tabs[5], Np, d, qn, Q, kv)
                return qn, Q, kv

            kw["first_pre"] = first_pre
            skip_embed = True
        if not skip_embed:
            ops._c("rt_embed_packed_fwd", ids, dist, table, pos, float(scale), Np, d, 0.0, 0, 0, x)
        return layers.forward_last_packed(x, cu, B, window, self.use_key_padding_mask, rows_real=n_rows, causal=self.use_causal_attn, **kw)

    def encode_packed_train(self, ids: torch.Tensor, dist: torch.Tensor, cu: torch.Tensor, B: int, window: int,
                            item_embs: tp.Optional[torch.Tensor] = None, rows_real: tp.Optional[int] = None,
                            cu_attn: tp.Optional[torch.Tensor] = None, ts: tp.Optional[torch.Tensor] = None,
                            n_prefixed: tp.Optional[int] = None) -> torch.Tensor:
        """Training twin of `encode_sessions` on packed rows: ids / dist [Np] (tail rows: id 0, dist 0), -> [Np, d].  ONE fused
        pass (`ops.embed_packed`): embedding rows (pad id 0 has no gradient), positional rows by the distance from the session's
        end, the embedding dropout (torch_backbone.py:245-247)."""
        table = self.item_model.table if item_embs is None else item_embs
        scale = self._pos_scale(table)
        pos = self.pos_encoding_layer.pos_emb.weight if self.pos_encoding_layer.pos_emb is not None else None
        seqs = self._watch_input_gradient(ops.embed_packed(table, pos, ids, dist, cu, B, window, scale, self.dropout_rate if self.training else 0.0))
        pk = {} if n_prefixed is None else {"n_prefixed": int(n_prefixed)}      # (session n_prefixed of cu: the shared pad prefix, `LiGRLayers.packed_mode`)
        if cu_attn is not None and rows_real is not None:
            # cu_attn [B + 2]: the unused tail of the row block as one more session of the attention — every row of every buffer of
            # the blocks is then written with finite values (zero gradients flow into the tail), no tail memsets
            return self.transformer_layers.forward_packed_train(seqs, cu_attn, B + 1, window, self.use_key_padding_mask, int(seqs.shape[0]),
                                                                causal=self.use_causal_attn, **pk)
        kw = dict(pk) if ts is None else dict(pk, ts=ts)     # (the STU stack's relative time bias: packed timestamps, `ops.collate_packed_ts`)
        return self.transformer_layers.forward_packed_train(seqs, cu, B, window, self.use_key_padding_mask, rows_real,
                                                            causal=self.use_causal_attn, **kw)

    def encode_last(self, batch: Batch, item_embs: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
        """-> [B, d] = encode_sessions(batch)[:, -1, :], the only rows recommend() uses (lightning.py:393-397).  Layer stacks
        that offer `forward_last` (SASRec) run their final block on one query row per session."""
        fast = getattr(self.transformer_layers, "forward_last", None)
        if fast is None or self.training or torch.is_grad_enabled() or type(self).encode_sessions is not TransformerTorchBackbone.encode_sessions:
            return self.encode_sessions(batch, item_embs)[:, -1, :].contiguous()      # (a subclass's encode_sessions is the one that runs)
        x = batch["x"]
        B, L = x.shape
        table = self.item_model.table if item_embs is None else item_embs
        ids = x.reshape(-1)
        seqs = self._embed_sessions(table, ids, B, L, 0.0)
        return fast(seqs, ids, B, L, self.use_causal_attn, self.use_key_padding_mask, batch)


# ---- models whose width / head size the kernels cannot tile: zero-padded columns ---------------------------------------------------
@dataclasses.dataclass(frozen=True)
class DimPlan:
    """The reference accepts any `n_factors % n_heads == 0` (nn.MultiheadAttention's own check; hstu.py:606-607) — its published HSTU
    quality numbers come from n_factors = 50 with 1 and 2 heads.  The kernels move float4 columns and tile heads in
    8-column groups.  A `DimPlan` runs such a model at sizes the kernels tile, EXACTLY: every activation row and every parameter carries
    zero columns / rows behind its real ones, head by head —

      * a zero column of the residual stream meets zero weight columns in every product and stays zero through every skip connection;
        LayerNorm takes its statistics over the real columns only and writes zeros to the others (`rt_layernorm_*_cols`);
      * zero q / k columns add nothing to a logit (softmax scale = 1 / sqrt(REAL head size): `rt_mha_*_scaled`), zero v columns give zero
        output columns; relu / gelu / silu / SwiGLU map 0 to 0, a sigmoid gate multiplies a zero branch;
      * every gradient of a padded entry is a product with one of those zeros, so Adam (m = v = 0 -> update 0) keeps them at zero.

    `state_dict()` / `load_state_dict()` / the checkpoints' Adam state speak the REAL shapes (hooks below): reference checkpoints
    interchange.  Initialisation happens at the real shapes (models._build_model_from_dataset builds the real-size twin on the host)."""
    d: int                # n_factors
    d_pad: int
    n_heads: int
    qk: int               # real head size of q / k (softmax stacks: n_factors // n_heads; STU: attention_dim)
    vo: int               # real head size of v / the attention output (STU: linear_hidden_dim)
    hd_pad: int           # the head size the kernels run, both kinds
    kind: str             # "mha" (SASRec / Pre-LN / LiGR blocks) | "stu"

    @property
    def row_cols(self) -> tp.Optional[tp.Tuple[int, int]]:
        return None if self.d_pad == self.d else (self.d_pad, self.d)

    @property
    def head_cols(self) -> tp.Optional[tp.Tuple[int, int]]:
        return None if self.hd_pad == self.vo else (self.hd_pad, self.vo)

    @staticmethod
    def make(n_factors: int, n_heads: int, kind: str = "mha", linear_hidden_dim: tp.Optional[int] = None,
             attention_dim: tp.Optional[int] = None) -> tp.Optional["DimPlan"]:
        """None when the kernels tile the model as it is."""
        up8 = lambda n: (n + 7) // 8 * 8  # noqa: E731
        hd = n_factors // n_heads
        if kind == "mha":
            if hd % 8 == 0 and hd <= 128:
                return None
            plan = DimPlan(n_factors, n_heads * up8(hd), n_heads, hd, hd, up8(hd), "mha")
        else:
            lin = hd if linear_hidden_dim is None else int(linear_hidden_dim)
            att = hd if attention_dim is None else int(attention_dim)
            if lin == att and lin % 8 == 0 and lin <= 128 and n_factors % 4 == 0:
                return None
            plan = DimPlan(n_factors, n_factors if n_factors % 4 == 0 else up8(n_factors), n_heads, att, lin, up8(max(lin, att)), "stu")
        if plan.hd_pad > 128:
            raise NotImplementedError(f"the HIP attention kernels hold a head of at most 128 columns; got a head size of {max(plan.qk, plan.vo)} "
                                      f"(n_factors={n_factors}, n_heads={n_heads}): use more heads")
        return plan

    # real position -> padded position, per axis kind
    def axis(self, kind: tp.Optional[str], n_pad: int) -> tp.Optional[torch.Tensor]:
        H, hp = self.n_heads, self.hd_pad
        heads = lambda r, base=0: (base + torch.arange(H)[:, None] * hp + torch.arange(r)[None, :]).reshape(-1)  # noqa: E731
        if kind is None:
            return None
        if kind == "D":
            return torch.arange(self.d)
        if kind == "F":           # hidden width of a feed-forward: multiplier x n_factors
            return torch.arange(n_pad // self.d_pad * self.d)
        if kind == "HV":
            return heads(self.vo)
        if kind == "QKV":         # nn.MultiheadAttention's packed in_proj: q | k | v, each H heads
            return torch.cat([heads(self.qk, i * H * hp) for i in range(3)])
        if kind == "UVQK":        # hstu.py:259-262: u | v (linear_hidden_dim) | q | k (attention_dim)
            return torch.cat([heads(self.vo, 0), heads(self.vo, H * hp), heads(self.qk, 2 * H * hp), heads(self.qk, 3 * H * hp)])
        raise ValueError(kind)


def _set_axes(plan: DimPlan, p: tp.Optional[torch.Tensor], *kinds: tp.Optional[str]) -> None:
    if p is None:
        return
    axes = tuple(plan.axis(k, int(n)) for k, n in zip(kinds, p.shape))
    if all(a is None or int(a.numel()) == int(n) for a, n in zip(axes, p.shape)):
        return        # nothing padded on any axis of this parameter
    p._rt_axes = axes                                                           # pylint: disable=protected-access
    p._rt_real_shape = tuple(int(n) if a is None else int(a.numel()) for a, n in zip(axes, p.shape))   # pylint: disable=protected-access


def unpad_tensor(t: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """The REAL entries of `t` (shaped like the padded parameter `like`); `t` itself when `like` carries no padding."""
    axes = getattr(like, "_rt_axes", None)
    if axes is None:
        return t
    for dim, idx in enumerate(axes):
        if idx is not None:
            t = t.index_select(dim, idx.to(t.device))
    return t


def pad_tensor(real: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """`real` (the parameter's real shape) placed into zeros of the padded parameter's shape; `real` itself without padding."""
    axes = getattr(like, "_rt_axes", None)
    if axes is None:
        return real
    if tuple(real.shape) != tuple(like._rt_real_shape):      # pylint: disable=protected-access
        raise ValueError(f"expected a tensor of shape {tuple(like._rt_real_shape)}, got {tuple(real.shape)}")   # pylint: disable=protected-access
    out = torch.zeros(tuple(like.shape), dtype=real.dtype, device=real.device)
    index = []
    for dim, (idx, n) in enumerate(zip(axes, like.shape)):
        ix = (torch.arange(n) if idx is None else idx).to(real.device)
        index.append(ix.view([-1 if i == dim else 1 for i in range(len(axes))]))
    out[tuple(index)] = real
    return out


def _unpad_state_hook(module: nn.Module, state_dict: tp.Dict[str, torch.Tensor], prefix: str, local_metadata: tp.Any) -> None:
    for name, p in module.named_parameters():
        if getattr(p, "_rt_axes", None) is not None and prefix + name in state_dict:
            state_dict[prefix + name] = unpad_tensor(state_dict[prefix + name], p)


def _pad_state_hook(module: nn.Module, state_dict: tp.Dict[str, torch.Tensor], prefix: str, *args: tp.Any) -> None:
    for name, p in module.named_parameters():
        t = state_dict.get(prefix + name)
        if t is not None and getattr(p, "_rt_axes", None) is not None and tuple(t.shape) != tuple(p.shape):
            state_dict[prefix + name] = pad_tensor(t, p)


def apply_dim_plan(backbone: "TransformerTorchBackbone", plan: DimPlan) -> None:
    """Mark a backbone BUILT AT THE PADDED SIZES (n_factors = plan.d_pad; STU head sizes = plan.hd_pad) as carrying zero columns: which
    entries of every parameter are real, the column patterns of its LayerNorms, the softmax scale of its real head size, the generic
    (op-by-op, padded-window) form of its blocks, and the state-dict hooks that translate to and from the real shapes."""
    row, head = plan.row_cols, plan.head_cols
    backbone.d_real = plan.d
    if backbone.pos_encoding_layer is not None:
        backbone.pos_encoding_layer.d_real = plan.d
    for m in backbone.modules():
        if isinstance(m, EmbeddingParams):                    # item ids, category values, positions: [n, n_factors]
            _set_axes(plan, m.weight, None, "D")
        elif isinstance(m, LayerNormParams):
            m.cols = row
            _set_axes(plan, m.weight, "D"); _set_axes(plan, m.bias, "D")
        elif isinstance(m, MultiheadAttnParams):
            m.scale = 1.0 / math.sqrt(plan.qk)
            _set_axes(plan, m.in_proj_weight, "QKV", "D"); _set_axes(plan, m.in_proj_bias, "QKV")
            _set_axes(plan, m.out_proj.weight, "D", "HV"); _set_axes(plan, m.out_proj.bias, "D")
        elif isinstance(m, (PointWiseFeedForward, SwigluFeedForward)):
            for lin in (m.ff_linear_1, getattr(m, "ff_linear_3", None)):
                if lin is not None:
                    _set_axes(plan, lin.weight, "F", "D"); _set_axes(plan, lin.bias, "F")
            _set_axes(plan, m.ff_linear_2.weight, "D", "F"); _set_axes(plan, m.ff_linear_2.bias, "D")
        if isinstance(m, LiGRLayer):
            for lin in (m.gating_linear_1, m.gating_linear_2):
                _set_axes(plan, lin.weight, "D", "D"); _set_axes(plan, lin.bias, "D")
        if isinstance(m, (SASRecTransformerLayer, PreLNTransformerLayer, LiGRLayer, STULayer)):
            m.generic = True
    for m in backbone.modules():      # (after the generic pass: the STU block's second LayerNorm runs over the heads' columns)
        if isinstance(m, STULayer):
            _set_axes(plan, m.uvqk_proj, "D", "UVQK")
            _set_axes(plan, m.output_mlp.weight, "D", "HV"); _set_axes(plan, m.output_mlp.bias, "D")
            ln = m.norm_attn_output
            ln.cols = head
            for t in (ln.weight, ln.bias):
                for attr in ("_rt_axes", "_rt_real_shape"):
                    if hasattr(t, attr):
                        delattr(t, attr)
                _set_axes(plan, t, "HV")
    backbone.register_state_dict_post_hook(_unpad_state_hook)
    backbone.register_load_state_dict_pre_hook(_pad_state_hook)
    backbone.dim_plan = plan
