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
  * the similarity module ranks through `HipRanker` (`rt_topk_score*`) instead of `TorchRanker` (`rank_torch.py:77-223`): same triplet;
  * `HipTransformerLightningModule` (`lightning_module_type`) computes the training loss with the fused loss kernels — the survey's
    hotspot #1, `similarity.py:92-94` + `lightning.py:144-212`: no [B, L, 1 + N, d] gather, no [B, L, V] logits — and hands Lightning a
    `torch.optim.Optimizer` whose step is the fused flat Adam kernel (`lightning.py:214-218`);
  * `HipIdEmbeddingsItemNet` (`item_net_block_types`) hands the id table out as it is (`item_net.py:361-368` re-gathers it with an
    `arange` every forward); `HipCatalogUniformSampler` (`negative_sampler_type`) draws the negatives on the device
    (`negative_sampler.py:58-73` draws on the host inside the collate and ships them over PCIe).

All of them at once:

    rectools.models.SASRecModel(
        transformer_layers_type="rectools_amd.reference_plugins.HipSASRecTransformerLayers",
        similarity_module_type="rectools_amd.reference_plugins.HipDistanceSimilarityModule",
        lightning_module_type="rectools_amd.reference_plugins.HipTransformerLightningModule",
        item_net_block_types=("rectools_amd.reference_plugins.HipIdEmbeddingsItemNet",),
        negative_sampler_type="rectools_amd.reference_plugins.HipCatalogUniformSampler", ...)

What stays the reference's: the DataLoader + collate (`sasrec.py:86-166`; the engine's own models cut batches on the device,
`rectools_amd.models`), Lightning's loop, recommend()'s pandas glue.
"""
from __future__ import annotations

import typing as tp

import numpy as np
import torch
from rectools.models.nn.item_net import IdEmbeddingsItemNet as _RefIdItemNet
from rectools.models.nn.transformers.lightning import TransformerLightningModule as _RefLightning
from rectools.models.nn.transformers.negative_sampler import CatalogUniformSampler as _RefSampler
from rectools.models.nn.transformers.net_blocks import TransformerLayersBase as _RefLayersBase
from rectools.models.nn.transformers.similarity import DistanceSimilarityModule as _RefSimilarity

from . import checkpoint as ckpt
from . import lightning as hl
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


class FlatAdamOptimizer(torch.optim.Optimizer):
    """`torch.optim.Adam(lr, betas)` of `configure_optimizers` (lightning.py:214-218) as ONE fused kernel: a `torch.optim.Optimizer` façade
    over `lightning.FlatAdam` (the parameters move into one flat fp32 buffer — their `.data` become views of it —, the moments live in
    two more, `rt_adam_step_segments` reads every gradient through its own pointer).  `state_dict()` / `load_state_dict()` speak
    torch.optim.Adam's layout (per-parameter `exp_avg` / `exp_avg_sq` / `step`): Lightning checkpoints interchange with the stock run.
    Build it when the module is on its device (Lightning calls `configure_optimizers` after moving the module)."""

    def __init__(self, module: torch.nn.Module, lr: float, betas: tp.Tuple[float, float] = (0.9, 0.98), eps: float = 1e-8) -> None:
        params = [p for p in module.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0))
        self.flat = hl.FlatAdam(module, lr=lr, betas=tuple(betas), eps=eps)

    def step(self, closure: tp.Optional[tp.Callable[[], tp.Any]] = None) -> tp.Any:      # type: ignore[override]
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        group = self.param_groups[0]       # (a scheduler writes the learning rate here)
        self.flat.lr, self.flat.betas, self.flat.eps = float(group["lr"]), tuple(group["betas"]), float(group["eps"])
        self.flat.step()
        return loss

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.flat.zero_grad()

    def state_dict(self) -> tp.Dict[str, tp.Any]:
        return ckpt.adam_state_dict(self.flat)

    def load_state_dict(self, state_dict: tp.Dict[str, tp.Any]) -> None:
        ckpt.load_adam_state_dict(self.flat, state_dict)
        for group in self.param_groups:
            group["lr"], group["betas"], group["eps"] = self.flat.lr, tuple(self.flat.betas), self.flat.eps


class HipTransformerLightningModule(_RefLightning):
    """`TransformerLightningModule` (lightning.py:259-449) whose training step computes the loss with the fused kernels and whose
    optimiser is the fused Adam.  Everything else — hooks, logging, validation, recommend — is the reference's own code.

    The fused path needs what the kernels define: one of the four stock losses and a similarity module whose logits are the stock dot /
    cosine ones (`DistanceSimilarityModule` or `HipDistanceSimilarityModule`, `forward` not overridden); any other combination takes
    the reference's own `training_step`."""

    def _fused(self) -> bool:
        sim = self.torch_model.similarity_module
        return (self.loss in hl.LOSSES and isinstance(sim, _RefSimilarity) and type(sim).forward is _RefSimilarity.forward
                and type(sim)._get_pos_neg_logits is _RefSimilarity._get_pos_neg_logits      # pylint: disable=protected-access
                and type(sim)._get_full_catalog_logits is _RefSimilarity._get_full_catalog_logits)      # pylint: disable=protected-access

    def training_step(self, batch: tp.Dict[str, torch.Tensor], batch_idx: int) -> torch.Tensor:
        item_embs = self.torch_model.item_model.get_all_embeddings()
        if not (self._fused() and item_embs.is_cuda and item_embs.dtype == torch.float32 and item_embs.shape[1] % 4 == 0):
            return super().training_step(batch, batch_idx)
        session_embs = self.torch_model.encode_sessions(batch, item_embs)            # torch_backbone.py:290-292
        B, L, d = session_embs.shape
        cosine = str(getattr(self.torch_model.similarity_module.distance, "value", self.torch_model.similarity_module.distance)) == "cosine"
        n_extra = len(self.item_extra_tokens)
        loss, _ = hl.fused_loss(item_embs.contiguous(), session_embs.reshape(B * L, d), batch["y"], batch["yw"], batch.get("negatives"),
                                self.loss, cosine, float(self.logits_t), float(self.gbce_t), n_extra)
        self.log(self.train_loss_name, loss, on_step=False, on_epoch=True, prog_bar=self.verbose > 0)
        return loss

    def configure_optimizers(self) -> torch.optim.Optimizer:      # type: ignore[override]
        if self.optimizer is None:
            if next(self.torch_model.parameters()).is_cuda:
                self.optimizer = FlatAdamOptimizer(self.torch_model, lr=self.lr, betas=tuple(self.adam_betas))
            else:
                return super().configure_optimizers()
        return self.optimizer


class HipIdEmbeddingsItemNet(_RefIdItemNet):
    """`IdEmbeddingsItemNet` (item_net.py:236-281) whose catalog matrix is the embedding table itself: `get_all_embeddings()` of the
    reference builds `arange(n_items)` on the host, ships it and gathers every row on every forward (item_net.py:44-52, 361-368).  The PAD
    row keeps `padding_idx` semantics — no gradient reaches it.  Same parameter name (`ids_emb.weight`): checkpoints interchange."""

    def get_all_embeddings(self) -> torch.Tensor:
        return hl._PadRowNoGrad.apply(self.ids_emb.weight)      # pylint: disable=protected-access


class HipCatalogUniformSampler(_RefSampler):
    """`CatalogUniformSampler` (negative_sampler.py:49-73) drawing on the device: `rt_sample_negatives` (Philox4x32-10) fills the
    [B, L | 1, N] tensor in HBM — called from the reference's collate (use `dataloader_num_workers=0`: a forked worker has no HIP
    context), the tensor is already where Lightning would move it.  Batch c of a sampler seeded s is a function of (s, c); parity with
    the reference is distributional, as for the reference's own draws between two runs."""

    def __init__(self, n_negatives: int, seed: int = 0, **kwargs: tp.Any) -> None:
        super().__init__(n_negatives, **kwargs)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.calls = 0

    def get_negatives(self, batch_dict: tp.Dict, lowest_id: int, highest_id: int, session_len_limit: tp.Optional[int] = None,      # type: ignore[override]
                      **kwargs: tp.Any) -> torch.Tensor:
        x = batch_dict["x"]
        session_len = session_len_limit if session_len_limit is not None else x.shape[1]
        device = x.device if x.is_cuda else torch.device("cuda")
        out = torch.empty((x.shape[0], session_len, self.n_negatives), dtype=torch.int64, device=device)
        self.calls += 1
        ops._c("rt_sample_negatives", int(lowest_id), int(highest_id), out.numel(), self.seed, self.calls, out)      # pylint: disable=protected-access
        return out
