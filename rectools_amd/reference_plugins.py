"""The module of plug-in classes a RecTools maintainer adds: the HIP engine behind the REFERENCE's own model classes.

Importable only where `rectools` itself is (the classes derive from the reference's base classes, as its pydantic config validation
requires: `transformers/base.py:58-186`).  Nothing in RecTools changes — the classes are selected by dotted path:

    rectools.models.SASRecModel(
        transformer_layers_type="rectools_amd.reference_plugins.HipSASRecTransformerLayers",      # sasrec.py:233
        similarity_module_type="rectools_amd.reference_plugins.HipDistanceSimilarityModule",      # similarity.py:67
        get_trainer_func=...  # any Trainer that trains on the GPU
    )

  * the layer stacks keep the reference's parameter names and shapes (checkpoints interchange) and its forward signature
    `(seqs [B, L, d], timeline_mask, attn_mask, key_padding_mask, batch=...)` (`net_blocks.py:154-185`); masks are not read — the
    kernels derive them from the item ids in `batch["x"]` (`torch_backbone.py:243-259`): a 2-D `attn_mask` means causal, a merged 3-D
    one causal + key padding (`torch_backbone.py:172-218`), `key_padding_mask` alone key padding;
  * every block is `torch.autograd.Function`s over the C ABI: Lightning (or any loop) drives them like any other module; weight
    gradients issued on the side stream are joined by an autograd-engine callback at the end of `backward()`;
  * the similarity module ranks through `HipRanker` (`rt_topk_score*`) instead of `TorchRanker` (`rank_torch.py:77-223`): same triplet.
"""
from __future__ import annotations

import typing as tp

import numpy as np
import torch
from rectools.models.nn.transformers.net_blocks import TransformerLayersBase as _RefLayersBase
from rectools.models.nn.transformers.similarity import DistanceSimilarityModule as _RefSimilarity

from . import nn as hnn
from . import ops
from .rank import HipRanker


class _ReferenceSignature:
    """forward() of the reference's `TransformerLayersBase` in front of a HIP layer stack (mixed in before it)."""

    def forward(self, seqs: torch.Tensor, timeline_mask: torch.Tensor, attn_mask: tp.Optional[torch.Tensor],      # type: ignore[override]
                key_padding_mask: tp.Optional[torch.Tensor], **kwargs: tp.Any) -> torch.Tensor:
        batch = kwargs.get("batch")
        if batch is None or "x" not in batch:
            raise ValueError("the HIP layer stacks derive their masks from the item ids: pass `batch=batch` (torch_backbone.py:259 does)")
        B, L, d = seqs.shape
        causal = attn_mask is not None
        keypad = key_padding_mask is not None or (attn_mask is not None and attn_mask.dim() == 3)
        if self.training:
            ops.RNG.next_step()       # a fresh set of dropout streams per training forward
        out = self._hip_forward(seqs.reshape(B * L, d).contiguous(), batch["x"].reshape(-1), B, L, causal, keypad, batch)
        return out.view(B, L, d)


def _plug(hip_cls: tp.Type[hnn.TransformerLayersBase], name: str, doc: str) -> tp.Type[_RefLayersBase]:
    def _hip_forward(self: tp.Any, seqs: torch.Tensor, ids: torch.Tensor, B: int, L: int, causal: bool, keypad: bool, batch: tp.Any) -> torch.Tensor:
        return hip_cls.forward(self, seqs, ids, B, L, causal, keypad, batch)

    return type(name, (_ReferenceSignature, hip_cls, _RefLayersBase), {"_hip_forward": _hip_forward, "__doc__": doc, "__module__": __name__})


HipSASRecTransformerLayers = _plug(hnn.SASRecTransformerLayers, "HipSASRecTransformerLayers",
                                   "`rectools_amd.nn.SASRecTransformerLayers` behind `SASRecTransformerLayers`' signature (sasrec.py:233-304).")
HipPreLNTransformerLayers = _plug(hnn.PreLNTransformerLayers, "HipPreLNTransformerLayers",
                                  "`rectools_amd.nn.PreLNTransformerLayers` behind `PreLNTransformerLayers`' signature (net_blocks.py:264-335).")
HipLiGRLayers = _plug(hnn.LiGRLayers, "HipLiGRLayers", "`rectools_amd.nn.LiGRLayers` behind `LiGRLayers`' signature (ligr.py:109-191).")
HipSTULayers = _plug(hnn.STULayers, "HipSTULayers", "`rectools_amd.nn.STULayers` behind `STULayers`' signature (hstu.py:298-399).")


class HipDistanceSimilarityModule(_RefSimilarity):
    """`DistanceSimilarityModule` (similarity.py:67-140) whose recommend step ranks with the exact top-k HIP kernels.  Logits for the
    losses stay the reference's own (`forward`): this class replaces `_recommend_u2i` only."""

    def _recommend_u2i(self, user_embs: torch.Tensor, item_embs: torch.Tensor, user_ids: np.ndarray, k: int,      # type: ignore[override]
                       sorted_item_ids_to_recommend: np.ndarray, ui_csr_for_filter: tp.Any) -> tp.Tuple[np.ndarray, np.ndarray, np.ndarray]:
        device = item_embs.device if item_embs.is_cuda else torch.device("cuda")
        ranker = HipRanker(self.distance.name.lower() if hasattr(self.distance, "name") else str(self.distance), device,
                           user_embs[user_ids], item_embs)
        rows, reco_ids, scores = ranker.rank(np.arange(len(user_ids)), k=k, filter_pairs_csr=ui_csr_for_filter,
                                             sorted_object_whitelist=sorted_item_ids_to_recommend)
        return np.asarray(user_ids)[rows], reco_ids, scores
