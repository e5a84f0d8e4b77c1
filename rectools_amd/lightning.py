"""Training / inference glue of the transformer models on the HIP engine — mirror of the reference's
`TransformerLightningModule` (rectools/models/nn/transformers/lightning.py:259-449) without PyTorch-Lightning.

  * `training_loss(batch)`      == `training_step` (lightning.py:311-321): logits / logits_t -> loss
  * `batch_logits(batch)`       == `get_batch_logits` (lightning.py:301-309) — parity / validation only
  * `FlatAdam`                  == `configure_optimizers` (lightning.py:214-218): dense Adam(lr, betas=(0.9,0.98)),
                                   as ONE fused kernel over a flat parameter buffer; with world_size > 1 the flat
                                   gradient is summed with ONE RCCL all-reduce (DDP's averaging is folded into the
                                   Adam kernel's grad_scale)
  * `xavier_normal_init`        == `_xavier_normal_init` (lightning.py:366-369)
"""
from __future__ import annotations

import os
import typing as tp

import torch
from torch import nn

from . import ops
from .nn import TransformerTorchBackbone
from .rank import Distance

Batch = tp.Dict[str, torch.Tensor]
LOSSES = ("softmax", "BCE", "gBCE", "sampled_softmax")


def requires_negatives(loss: str) -> tp.Optional[bool]:
    """lightning.py:115-124"""
    if loss == "softmax":
        return False
    if loss in ("BCE", "gBCE", "sampled_softmax"):
        return True
    return None


def xavier_normal_init(model: nn.Module) -> None:
    for _, param in model.named_parameters():
        if param.data.dim() > 1:
            torch.nn.init.xavier_normal_(param.data)


def gbce_beta(n_negatives: int, n_items: int, gbce_t: float) -> float:
    """lightning.py:170-177: alpha = N / (n_items - 1); beta = alpha * (t * (1 - 1/alpha) + 1/alpha)."""
    alpha = n_negatives / (n_items - 1)
    return alpha * (gbce_t * (1 - 1 / alpha) + 1 / alpha)


def fused_loss(table: torch.Tensor, sess2d: torch.Tensor, y: torch.Tensor, w: torch.Tensor, negatives: tp.Optional[torch.Tensor],
               loss: str, cosine: bool, logits_t: float, gbce_t: float, n_item_extra_tokens: int,
               n_targets: tp.Optional[int] = None) -> tp.Tuple[torch.Tensor, tp.Optional[torch.Tensor]]:
    """`get_batch_logits` + the loss calculator (lightning.py:144-212, 301-321) as the fused kernels: -> (loss, logits [M, 1 + N] / logits_t
    of the sampled losses | None).  table [V, d] (the catalog matrix of THIS step), sess2d [M, d] encoder rows, y / w [M], negatives
    [M, N].  softmax: `rt_gemm` + `rt_softmax_ce_rows` over the positions with a target; BCE / gBCE / sampled softmax:
    `rt_sampled_loss_*` (the [M, 1 + N, d] gather never exists).  Shared by `TransformerLossModule` and the reference-side
    `reference_plugins.HipTransformerLightningModule`."""
    y = y.reshape(-1)
    w = w.reshape(-1).contiguous()
    if loss == "softmax":
        if cosine:
            sess2d, table = ops.l2norm(sess2d), ops.l2norm(table)
        if n_targets is not None:   # the caller counted the targets on the host: no device -> host round trip for the size
            act = torch.nonzero_static(y, size=int(n_targets)).reshape(-1)
        else:
            act = torch.nonzero(y, as_tuple=False).reshape(-1)  # positions with a target (ignore_index = 0)
        return ops.softmax_loss(sess2d, table, act, y[act].contiguous(), w[act].contiguous(), logits_t), None
    kind = {"BCE": ops.LOSS_BCE, "gBCE": ops.LOSS_GBCE, "sampled_softmax": ops.LOSS_SAMPLED_SOFTMAX}[loss]
    beta = 0.0
    if loss == "gBCE":
        n_items = table.shape[0] - n_item_extra_tokens
        beta = gbce_beta(int(negatives.shape[-1]), n_items, gbce_t)
    return ops.sampled_loss(sess2d, table, y, negatives, w, kind, cosine, logits_t, beta)


class _PadRowNoGrad(torch.autograd.Function):
    """Identity on the catalog matrix whose backward zeroes row 0: `nn.Embedding(padding_idx=0)` (item_net.py:260-264) — the reference
    materialises the catalog THROUGH that lookup (`get_all_embeddings`, item_net.py:361-368), so the PAD row never receives a gradient,
    not even from full-catalog logits.  The fused losses zero the row themselves; a plugged similarity module reads the table through
    this node."""

    @staticmethod
    def forward(ctx, table):
        return table.view_as(table)

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        g[0] = 0
        return g


class TransformerLossModule(nn.Module):
    """Backbone + loss.  `n_item_extra_tokens` real-item offset is only needed for gBCE (lightning.py:202)."""

    def __init__(self, torch_model: TransformerTorchBackbone, loss: str = "softmax", n_negatives: tp.Optional[int] = None,
                 gbce_t: float = 0.2, logits_t: float = 1.0, n_item_extra_tokens: int = 1, **kwargs: tp.Any) -> None:
        """The leading arguments are this engine's; `**kwargs` takes the rest of the reference constructor's keyword set
        (lightning.py:75-91: model_config, dataset_schema, item_external_ids, item_extra_tokens, data_preparator, lr, verbose,
        train_loss_name, val_loss_name, adam_betas) — `models._build_model_from_dataset` instantiates the class given as
        `lightning_module_type` with exactly those keywords, so a subclass written against the reference's signature plugs in.
        They are kept as attributes (the optimiser and the loop live in `FlatAdam` / `models._TrainLoop`)."""
        super().__init__()
        if loss not in LOSSES and type(self)._loss_from_sessions is TransformerLossModule._loss_from_sessions:
            raise ValueError(f"loss {loss} is not supported")  # lightning.py:328 (a subclass with its own loss may name it freely)
        self.torch_model = torch_model
        self.loss = loss
        self.n_negatives = n_negatives
        self.gbce_t = gbce_t
        self.logits_t = logits_t
        self.n_item_extra_tokens = n_item_extra_tokens
        for name in ("model_config", "dataset_schema", "item_external_ids", "item_extra_tokens", "data_preparator", "lr", "verbose",
                     "train_loss_name", "val_loss_name", "adam_betas"):
            if name in kwargs:
                object.__setattr__(self, name, kwargs.pop(name))   # plain attributes (the data preparator is not a sub-module)
        self.extra_kwargs = kwargs

    @staticmethod
    def requires_negatives(loss: str) -> tp.Optional[bool]:
        """lightning.py:115-124 — consulted by the model before it builds the data preparator (transformers/base.py:354)."""
        return requires_negatives(loss)

    # LightningModule.log / log_dict as far as the engine's loop goes: a callback of a user-built Trainer (`models._trainer_plan`) logs
    # its metrics here; the loop copies them into the epoch's history record, the CSV row and `trainer.callback_metrics`
    def log(self, name: str, value: tp.Any, *args: tp.Any, **kwargs: tp.Any) -> None:
        metrics = self.__dict__.setdefault("logged_metrics", {})
        metrics[name] = float(value.detach()) if isinstance(value, torch.Tensor) else float(value)

    def log_dict(self, dictionary: tp.Mapping[str, tp.Any], *args: tp.Any, **kwargs: tp.Any) -> None:
        for name, value in dictionary.items():
            self.log(name, value)

    @property
    def cosine(self) -> bool:
        return self.torch_model.similarity_module.distance == Distance.COSINE

    def _encode(self, batch: Batch) -> tp.Tuple[torch.Tensor, torch.Tensor]:
        """-> (catalog matrix, session encodings).  The catalog matrix is taken ONCE per step and shared by the session
        encoder and the loss (torch_backbone.py:290-293): with a feature-aware item net each read is a fused pass with its
        own dropout mask."""
        table = self.torch_model.item_model.get_all_embeddings()
        return table, self.torch_model.encode_sessions(batch, table)

    def _loss_from_sessions(self, table: torch.Tensor, sess2d: torch.Tensor, y: torch.Tensor, w: torch.Tensor,
                            negatives: tp.Optional[torch.Tensor], n_targets: tp.Optional[int] = None
                            ) -> tp.Tuple[torch.Tensor, tp.Optional[torch.Tensor]]:
        return fused_loss(table, sess2d, y, w, negatives, self.loss, self.cosine, self.logits_t, self.gbce_t, self.n_item_extra_tokens,
                          n_targets)

    # ---- a plugged similarity module (`similarity_module_type`, transformers/base.py:415-421) -------------------------------------
    @property
    def similarity_is_stock(self) -> bool:
        """The fused loss kernels compute dot / cosine logits of the encoder's rows against the table's rows — what the stock
        `DistanceSimilarityModule` defines.  Any other module is called for its logits (`_loss_via_similarity`)."""
        from .nn import similarity_is_stock

        return similarity_is_stock(self.torch_model.similarity_module)

    def _logits_via_similarity(self, table: torch.Tensor, sess: torch.Tensor, y: torch.Tensor,
                               negatives: tp.Optional[torch.Tensor]) -> torch.Tensor:
        """`get_batch_logits` (lightning.py:301-309) with the similarity module's own `forward`: sess [B, L, d], y [B, L],
        negatives [B, L, N] -> [B, L, 1 + N] (sampled losses) or [B, L, V] (softmax), divided by logits_t."""
        sim = self.torch_model.similarity_module
        if table.is_leaf and table.requires_grad:      # an ids-only item net hands out the parameter itself (nn.SumOfEmbeddingsConstructor)
            table = _PadRowNoGrad.apply(table)
        if requires_negatives(self.loss) or (requires_negatives(self.loss) is None and negatives is not None):
            pos_neg = torch.cat([y.unsqueeze(-1), negatives], dim=-1)
            return sim(sess, table, pos_neg) / self.logits_t
        return sim(sess, table) / self.logits_t

    # the reference's loss calculators on materialised logits (lightning.py:144-212), device tensor ops under autograd: the path of a
    # plugged similarity module only — the stock step never materialises [B, L, V] or [B, L, 1 + N, d]
    @staticmethod
    def _calc_softmax_loss(logits: torch.Tensor, y: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        loss = torch.nn.functional.cross_entropy(logits.transpose(1, 2), y, ignore_index=0, reduction="none") * w
        return torch.sum(loss) / torch.sum((loss > 0).to(loss.dtype))

    @staticmethod
    def _calc_bce_loss(logits: torch.Tensor, y: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        mask = y != 0
        target = torch.zeros_like(logits)
        target[:, :, 0] = 1
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, target, reduction="none")
        loss = loss.mean(-1) * mask * w
        return torch.sum(loss) / torch.sum(mask)

    def _get_reduced_overconfidence_logits(self, logits: torch.Tensor, n_items: int, n_negatives: int) -> torch.Tensor:
        dtype = torch.float64                                             # lightning.py:167
        beta = gbce_beta(n_negatives, n_items, self.gbce_t)
        pos, neg = logits[:, :, 0:1].to(dtype), logits[:, :, 1:].to(dtype)
        eps = 1e-10
        p = torch.clamp(torch.sigmoid(pos), eps, 1 - eps)
        p = torch.clamp(p.pow(-beta), 1 + eps, torch.finfo(dtype).max)
        p = torch.clamp(torch.div(1, (p - 1)), eps, torch.finfo(dtype).max)
        return torch.cat([torch.log(p), neg], dim=-1)

    def _calc_loss_from_logits(self, logits: torch.Tensor, y: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        """`loss_calculator(logits, y, w)` (lightning.py:126-142).  Like the reference's `_calc_sampled_softmax_loss`
        (lightning.py:207-212) the sampled softmax swaps columns 0 and 1 of `logits` IN PLACE."""
        if self.loss == "softmax":
            return self._calc_softmax_loss(logits, y, w)
        if self.loss == "BCE":
            return self._calc_bce_loss(logits, y, w)
        if self.loss == "gBCE":
            n_items = int(self.torch_model.item_model.n_items) - self.n_item_extra_tokens
            return self._calc_bce_loss(self._get_reduced_overconfidence_logits(logits, n_items, int(logits.shape[-1]) - 1), y, w)
        if self.loss == "sampled_softmax":
            logits[:, :, [0, 1]] = logits[:, :, [1, 0]]
            return self._calc_softmax_loss(logits, (y != 0).long(), w)
        raise ValueError(f"loss {self.loss} is not supported")

    def _loss_via_similarity(self, table: torch.Tensor, sess: torch.Tensor, y: torch.Tensor, w: torch.Tensor,
                             negatives: tp.Optional[torch.Tensor]) -> tp.Tuple[torch.Tensor, torch.Tensor]:
        logits = self._logits_via_similarity(table, sess, y, negatives)
        return self._calc_loss_from_logits(logits, y, w), logits

    def training_loss(self, batch: Batch) -> torch.Tensor:
        table, sess = self._encode(batch)
        B, L, d = sess.shape
        if not self.similarity_is_stock and type(self)._loss_from_sessions is TransformerLossModule._loss_from_sessions:
            return self._loss_via_similarity(table, sess, batch["y"], batch["yw"], batch.get("negatives"))[0]
        loss, _ = self._loss_from_sessions(table, sess.view(B * L, d), batch["y"], batch["yw"], batch.get("negatives"))
        return loss

    def training_loss_packed(self, pbatch: Batch) -> torch.Tensor:
        """`training_loss` on a packed batch: x / y / yw / dist [Np], negatives [Np, N] (rows behind the last
        session: id 0, target 0 — the loss ignores them), cu [B+1], window.  Same value and gradients as the padded batch of the
        same sessions (tests/test_packed_gpu.py)."""
        table = self.torch_model.item_model.get_all_embeddings()
        B = int(pbatch["cu"].numel()) - 1
        sess = self.torch_model.encode_packed_train(pbatch["x"], pbatch["dist"], pbatch["cu"], B, int(pbatch["window"]), table,
                                                    rows_real=pbatch.get("n_rows"), cu_attn=pbatch.get("cu_attn"), ts=pbatch.get("ts"),
                                                    **({"n_prefixed": pbatch["n_prefixed"]} if pbatch.get("n_prefixed") is not None else {}))
        stock = type(self)._loss_from_sessions is TransformerLossModule._loss_from_sessions    # (a plugged loss keeps its own signature)
        kw = {"n_targets": pbatch["n_targets"]} if stock and pbatch.get("n_targets") is not None and self.loss == "softmax" else {}
        loss, _ = self._loss_from_sessions(table, sess, pbatch["y"], pbatch["yw"], pbatch.get("negatives"), **kw)
        return loss

    def validation_loss(self, batch: Batch) -> torch.Tensor:
        """Last position only (lightning.py:340-349): y, yw [B,1]; negatives [B,1,N]."""
        return self.validation_step(batch, want_logits=False)["loss"]

    def validation_step(self, batch: Batch, batch_idx: int = 0, want_logits: bool = True) -> tp.Dict[str, torch.Tensor]:
        """`validation_step` (lightning.py:336-359): {"loss", "pos_neg_logits" [B, 1 + N] | "logits" [B, V]} — the outputs a
        validation callback receives (`on_validation_batch_end`; examples/tutorials/utils.py:54-118 reads "logits").  Logits are
        divided by logits_t; with the sampled softmax columns 0 and 1 come swapped, as the reference hands them on (its loss
        calculator swaps them in place before the outputs are taken, lightning.py:209,346-352).  want_logits=False skips the
        [B, V] product of the softmax loss (nobody listens)."""
        table, sess = self._encode(batch)
        last = sess[:, -1:, :]
        y, w, negatives = batch["y"], batch["yw"], batch.get("negatives")
        sampled = negatives is not None and requires_negatives(self.loss) is not False
        key = "pos_neg_logits" if sampled else "logits"
        if type(self)._loss_from_sessions is not TransformerLossModule._loss_from_sessions:      # a plugged loss: its own values
            loss, logits = self._loss_from_sessions(table, last[:, 0, :], y, w, negatives)
            return {"loss": loss} if logits is None else {"loss": loss, key: logits}
        if not self.similarity_is_stock:
            loss, logits = self._loss_via_similarity(table, last, y, w, negatives)
            return {"loss": loss, key: logits.squeeze()}
        loss, logits = self._loss_from_sessions(table, last[:, 0, :], y, w, negatives)
        out = {"loss": loss}
        if sampled:
            if self.loss == "sampled_softmax":
                logits = logits.clone()
                logits[:, [0, 1]] = logits[:, [1, 0]]
            out[key] = logits.squeeze()
        elif want_logits:
            sim = self.torch_model.similarity_module
            out[key] = (sim(last[:, 0, :].contiguous(), table) / self.logits_t).squeeze()
        return out

    def batch_logits(self, batch: Batch) -> torch.Tensor:
        """[B, L, 1+N] (sampled losses) — the values the reference's get_batch_logits returns; parity checks only."""
        if self.loss == "softmax":
            raise NotImplementedError("full-catalog logits are never materialised for all positions; use training_loss")
        table, sess = self._encode(batch)
        B, L, d = sess.shape
        _, logits = self._loss_from_sessions(table, sess.view(B * L, d), batch["y"], batch["yw"], batch["negatives"])
        return logits.view(B, L, -1)


class RcclExchange:
    """The gradient exchange through the library's own RCCL entry points (`rt_dp_*`, include/rectools_hip.h) instead of
    torch.distributed's collectives: one `ncclAllReduce` on the stream the gradient kernels ran on.  The 128-byte unique id is
    made by rank 0 and handed to the other ranks through the already initialised torch.distributed group (any backend: it only
    carries the id); opt-in with `RT_DP_BACKEND=rccl`, the default exchange stays `dist.all_reduce` (which is RCCL too)."""

    def __init__(self, rank: int, world: int, device: tp.Optional[torch.device] = None) -> None:
        import ctypes

        from . import _lib

        self._lib = _lib.load()
        self.rank, self.world = rank, world
        if device is not None:   # ncclCommInitRank binds the CURRENT HIP device: make it the one the parameters live on
            torch.cuda.set_device(device)
        uid = ctypes.create_string_buffer(128)
        if rank == 0:
            _lib.check(self._lib.rt_dp_unique_id(uid), "rt_dp_unique_id")
        if world > 1:
            import torch.distributed as dist

            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=0)
            uid = ctypes.create_string_buffer(box[0], 128)
        comm = ctypes.c_void_p()
        status = self._lib.rt_dp_init(uid, rank, world, ctypes.byref(comm))
        if status != 0:
            raise _lib.HipLibraryError(f"rt_dp_init failed ({status}): {(self._lib.rt_dp_last_error() or b'').decode()}")
        self.comm = comm

    def _call(self, name: str, *args: tp.Any) -> None:
        from . import _lib

        try:
            ops._c(name, *args)
        except _lib.HipLibraryError as e:    # the exchange keeps its own error text (RCCL's, not the kernels')
            raise _lib.HipLibraryError(f"{e}; rt_dp_last_error: {(self._lib.rt_dp_last_error() or b'').decode()}") from e

    def all_reduce(self, buf: torch.Tensor) -> None:
        self._call("rt_dp_allreduce", self.comm, buf, buf.numel())

    def broadcast(self, buf: torch.Tensor, src: int = 0) -> None:
        self._call("rt_dp_broadcast", self.comm, buf, buf.numel(), src)

    def reduce_scatter(self, send: torch.Tensor, recv: torch.Tensor) -> None:
        self._call("rt_dp_reduce_scatter", self.comm, send, recv, recv.numel())

    def all_gather(self, send: torch.Tensor, recv: torch.Tensor) -> None:
        self._call("rt_dp_allgather", self.comm, send, recv, send.numel())

    def close(self) -> None:
        if self.comm is not None:
            torch.cuda.synchronize()
            self._lib.rt_dp_finalize(self.comm)
            self.comm = None

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:   # interpreter shutdown: the runtime may be gone already
            pass


FLAT_QUANTUM = 107_520    # lcm(1024, 840) floats: see FlatAdam.__init__


class FlatAdam:
    """All parameters and Adam moments live in flat fp32 buffers; one fused kernel per step.

    Gradients are NOT accumulated into a flat buffer on one GPU: `zero_grad()` drops the `.grad` references, autograd
    then hands over its own output tensors (no AccumulateGrad add, no zero fill) and `rt_adam_step_segments` reads each
    parameter's gradient through its own pointer.  With data parallelism the gradients are packed into `flat_g` for ONE
    all-reduce per step and the flat kernel runs on the reduced buffer.
    """

    def __init__(self, module: nn.Module, lr: float, betas: tp.Tuple[float, float] = (0.9, 0.98), eps: float = 1e-8) -> None:
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("no parameters to optimise")
        dev = params[0].device
        # 32-byte aligned segments (a weight's bf16 planes then start on 16 bytes: `ops.WeightPlanes.of`).  Large 2-D parameters (embedding tables) additionally own ZERO rows up to the next
        # multiple of 128 behind their data (never touched by Adam: their gradient is zero): kernels that tile the table in
        # 128-row blocks — the full-catalog softmax GEMMs — may then read `_rt_rows_padded` rows and take the exact-tile
        # LDS-DMA path instead of the ragged-edge kernel (ops._SoftmaxLoss).
        def seg_floats(p: torch.Tensor) -> int:
            if p.dim() == 2 and p.shape[0] >= 1024 and p.shape[1] % 4 == 0:
                return (p.shape[0] + 127) // 128 * 128 * p.shape[1]
            return (p.numel() + 7) // 8 * 8

        sizes = [seg_floats(p) for p in params]
        self.n_used = sum(sizes)
        # tail of zeros up to a multiple of lcm(1024, 840) = 107,520 floats (430 KB): the sharded exchange cuts the buffers into
        # world_size equal, 16-byte aligned slices, and this length is divisible by every world size up to 8 (and 10, 12, 14, 15, 16 ...)
        total = (self.n_used + FLAT_QUANTUM - 1) // FLAT_QUANTUM * FLAT_QUANTUM
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(total, dtype=torch.float32, device=dev)
        self._flat_g: tp.Optional[torch.Tensor] = None   # allocated on first use (data-parallel runs only)
        self._g_views: tp.Optional[tp.List[torch.Tensor]] = None
        self.params = params
        self._offsets: tp.List[int] = []
        ofs = 0
        for p, sz in zip(params, sizes):
            view = self.flat_p[ofs:ofs + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = None
            if sz > (p.numel() + 7) // 8 * 8:
                p._rt_rows_padded = sz // p.shape[1]
            self._offsets.append(ofs)
            ofs += sz
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        self.exchange: tp.Optional[RcclExchange] = None   # set by use_rccl_exchange(): rt_dp_* instead of torch.distributed
        # RT_DP_EXCHANGE = allreduce | sharded | auto (sharded from 256 MB of gradient per step: the table-dominated configurations)
        import os

        mode = os.environ.get("RT_DP_EXCHANGE", "auto")
        # what the configuration ASKS for; whether a step takes it is decided at step time, when the world size is known
        # (`_use_sharded`: a world size that does not cut the buffer into aligned equal slices falls back to the all-reduce)
        self.sharded = mode == "sharded" or (mode == "auto" and total * 4 >= (256 << 20))
        self._g_shard: tp.Optional[torch.Tensor] = None
        # set by a sharded step: (world, rank) whose slice of the moments is current on this rank — the other slices are STALE until
        # `consolidate_moments()` (collective) has run; checkpoints refuse to be written from partial moments
        self.partial_moments: tp.Optional[tp.Tuple[int, int]] = None
        # bench.py --gpus N: device-side duration of every step's gradient exchange (pack + collective[s]), HIP events on the step's stream
        self.exchange_events: tp.Optional[tp.List[tp.Tuple[tp.Any, tp.Any]]] = None
        # the EARLY bucket of the all-reduce exchange (`set_early_bucket`): first parameter index / float offset of the flat buffer's tail
        # whose gradients exist before the backward pass ends; `_early`: the exchange in flight between `begin_early_exchange` and `step`
        self.early_first: tp.Optional[int] = None
        self.early_from: tp.Optional[int] = None
        self._early: tp.Optional[tp.Tuple[tp.Any, tp.List[torch.Tensor], tp.Any]] = None
        self._xs: tp.Optional["torch.cuda.Stream"] = None     # the exchange's own stream (made on first use)
        self.early_enabled = os.environ.get("RT_DP_EARLY", "1") != "0"
        self.early_stats = {"started": 0, "redone": 0}

    def use_rccl_exchange(self, rank: int, world: int) -> None:
        """Route the gradient all-reduce and the parameter broadcast through `rt_dp_*` (collective: every rank calls it)."""
        self.exchange = RcclExchange(rank, world, self.flat_p.device)

    @property
    def flat_g(self) -> torch.Tensor:
        if self._flat_g is None:
            self._flat_g = torch.zeros_like(self.flat_p)
        return self._flat_g

    def broadcast_parameters(self, src: int = 0, force: bool = False) -> None:
        """Data-parallel start: every replica takes rank `src`'s parameters and moments (what DDP does when it wraps a
        module).  One broadcast per flat buffer; a no-op without an initialised process group."""
        import torch.distributed as dist

        if self.exchange is not None:
            for buf in (self.flat_p, self.m, self.v):
                self.exchange.broadcast(buf, src)
        elif dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
            for buf in (self.flat_p, self.m, self.v):
                dist.broadcast(buf, src=src)

    def zero_grad(self) -> None:
        if self._early is not None:      # an early exchange nobody finished (the step was abandoned): let it land, drop its result
            work, _, done = self._early
            self._early = None
            if work is not None:
                work.wait()
            elif done is not None:
                torch.cuda.current_stream(self.flat_p.device).wait_event(done)
        for p in self.params:
            p.grad = None  # the next backward's gradient tensors are adopted as they are
        ops.clear_step_expectations()

    def gather_gradients(self, first: int = 0, last: tp.Optional[int] = None) -> torch.Tensor:
        """Pack the per-parameter gradients into the flat buffer (zeros where a parameter got none): ONE multi-tensor copy
        launch for all of them instead of a copy kernel per parameter (28 launches at C2 right before the collective).
        first / last: parameters [first, last) only (the two buckets of the early exchange)."""
        fg = self.flat_g
        if self._g_views is None:
            self._g_views = [fg[ofs:ofs + p.numel()].view_as(p) for p, ofs in zip(self.params, self._offsets)]
            # large tables: from the next step on the sampled losses write the table's gradient INTO its segment (`ops._TABLE_GRAD_HOME`)
            # — the copy below then skips it (1 GB per step at C4, 27 MB at C2)
            import weakref

            me = weakref.ref(self)      # (the registry must not keep the buffers of a dead optimiser alive)

            def home(ofs: int, shape: torch.Size, n: int) -> tp.Optional[torch.Tensor]:
                opt = me()
                return None if opt is None or opt._flat_g is None else opt._flat_g[ofs:ofs + n].view(shape)

            for p, ofs in zip(self.params, self._offsets):
                if p.dim() == 2 and p.is_cuda and p.numel() >= (1 << 20):
                    key = p.data_ptr()
                    ops._TABLE_GRAD_HOME[key] = (lambda ofs=ofs, shape=p.shape, n=p.numel(): home(ofs, shape, n))
                    weakref.finalize(self, ops._TABLE_GRAD_HOME.pop, key, None)
        views, grads = [], []
        for p, view in zip(self.params[first:last], self._g_views[first:last]):
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() == view.data_ptr() and p.grad.shape == view.shape and p.grad.is_contiguous():
                continue                                # produced in place
            else:
                views.append(view)
                grads.append(p.grad if p.grad.dtype == torch.float32 else p.grad.float())
        if views:
            torch._foreach_copy_(views, grads)   # pylint: disable=protected-access
        return fg

    # ---- the early bucket of the all-reduce exchange ---------------------------------------------------------------------------------
    def set_early_bucket(self, late: tp.Iterable[torch.Tensor]) -> None:
        """`late`: the parameters whose gradients are complete only when the backward pass ENDS — the input embeddings (item net,
        positions): the lookup's backward is the last node.  Every parameter behind the last of them in the flat buffer (the block
        weights: 13 % of C2's gradient bytes) forms the EARLY bucket: `begin_early_exchange` starts its all-reduce while the
        backward pass runs on, `step` exchanges the rest and runs Adam on the early bucket under that second collective — what DDP's
        bucketed all-reduce does behind the reference's `Trainer.fit` (transformers/base.py:367-380).  The sharded exchange (table-
        dominated gradients: the early bucket is < 1 % of the bytes there) keeps its single reduce-scatter."""
        late_ids = {id(p) for p in late}
        last = max((i for i, p in enumerate(self.params) if id(p) in late_ids), default=-1)
        if 0 <= last < len(self.params) - 1:
            self.early_first, self.early_from = last + 1, self._offsets[last + 1]
        else:
            self.early_first = self.early_from = None

    def _exchange_stream(self) -> "torch.cuda.Stream":
        if self._xs is None:
            self._xs = torch.cuda.Stream(device=self.flat_p.device)
        return self._xs

    def begin_early_exchange(self, world_size: int, force: bool = False) -> bool:
        """Called when every gradient of the early bucket EXISTS (a hook on the gradient of the blocks' input: the autograd engine has run
        the blocks' nodes and their accumulators, the lookup's backward comes next).  Packs the bucket and starts its sum all-reduce
        on the exchange stream, ordered behind the weight-gradient side streams (`ops.side_streams_reach`) — the calling stream is
        not held up.  COLLECTIVE: every rank takes the same decision (it depends on the configuration only).  -> started?"""
        import torch.distributed as dist

        if (self.early_from is None or not self.early_enabled or (world_size <= 1 and not force) or self._early is not None
                or self._use_sharded(world_size) or self.partial_moments is not None):      # (force: the one-rank RCCL smoke test)
            return False
        params = self.params[self.early_first:]
        if any(p.grad is None for p in params):      # (a frozen / unused parameter: the step's own pack writes its zeros)
            return False
        fg = self.flat_g
        cuda = fg.is_cuda
        done = None
        if cuda:
            xs = self._exchange_stream()
            xs.wait_stream(torch.cuda.current_stream(fg.device))
            ops.side_streams_reach(xs)
            ctx: tp.Any = torch.cuda.stream(xs)
        else:
            import contextlib

            ctx = contextlib.nullcontext()
        with ctx:
            self.gather_gradients(self.early_first, None)
            seg = fg[self.early_from:]
            if self.exchange is not None:
                self.exchange.all_reduce(seg)
                work = None
            else:
                work = dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True)
            if cuda:
                done = xs.record_event()
        self._early = (work, [p.grad for p in params], done)
        self.early_stats["started"] += 1
        return True

    def _finish_early(self, world_size: int, hyper: tp.Tuple) -> None:
        """The step behind `begin_early_exchange`: pack and exchange the LATE bucket (asynchronously), wait for the early one, Adam on
        the early bucket while the late collective runs, wait, Adam on the late bucket."""
        import torch.distributed as dist

        work_b, grads_b, done_b = self._early      # type: ignore[misc]
        self._early = None
        fg, ef = self.flat_g, self.early_from
        cuda = fg.is_cuda
        params_b = self.params[self.early_first:]
        stale = any(p.grad is not g for p, g in zip(params_b, grads_b))      # (somebody replaced a gradient behind the hook: e.g. a second backward)
        self.gather_gradients(0, self.early_first)
        late = fg[:ef]
        work_t, done_t = None, None
        if self.exchange is not None:      # one communicator: its collectives stay on ONE stream, in issue order
            xs = self._exchange_stream()
            xs.wait_stream(torch.cuda.current_stream(fg.device))
            with torch.cuda.stream(xs):
                self.exchange.all_reduce(late)
                done_t = xs.record_event()
        else:
            work_t = dist.all_reduce(late, op=dist.ReduceOp.SUM, async_op=True)
        if work_b is not None:
            work_b.wait()
        elif done_b is not None:
            torch.cuda.current_stream(fg.device).wait_event(done_b)
        if stale:      # exchange the bucket again from the gradients as they are now (the first sum is overwritten by the pack)
            self.early_stats["redone"] += 1
            if work_t is not None:
                work_t.wait()
            elif done_t is not None:
                torch.cuda.current_stream(fg.device).wait_event(done_t)
            self.gather_gradients(self.early_first, None)
            if self.exchange is not None:
                self.exchange.all_reduce(fg[ef:])
            else:
                dist.all_reduce(fg[ef:], op=dist.ReduceOp.SUM)
            self._adam_flat(self.flat_p, fg, self.m, self.v, hyper)
            return
        self._adam_flat(self.flat_p[ef:], fg[ef:], self.m[ef:], self.v[ef:], hyper)
        if work_t is not None:
            work_t.wait()
        elif done_t is not None:
            torch.cuda.current_stream(fg.device).wait_event(done_t)
        self._adam_flat(self.flat_p[:ef], late, self.m[:ef], self.v[:ef], hyper)

    def reduce_gradients(self, world_size: int = 1, force: bool = False) -> float:
        """Data-parallel gradient exchange: ONE sum all-reduce of the flat gradient buffer (RCCL over xGMI on GPUs,
        gloo in the CPU tests).  Returns the scale that turns the sum into DDP's mean; it is folded into the Adam
        kernel instead of a separate pass over the buffer.  `force` runs the collective with a single rank too (the
        RCCL smoke test of a one-GPU box)."""
        if world_size <= 1 and not force:
            return 1.0
        if self.exchange is not None:
            self.exchange.all_reduce(self.gather_gradients())
            return 1.0 / max(world_size, 1)
        import torch.distributed as dist

        dist.all_reduce(self.gather_gradients(), op=dist.ReduceOp.SUM)
        return 1.0 / max(world_size, 1)

    def _adam_flat(self, p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, hyper: tp.Tuple) -> None:
        """`rt_adam_step` over (a slice of) the flat buffers (the seam the gloo tests replace with a torch restatement)."""
        ops._c("rt_adam_step", p, g, m, v, p.numel(), *hyper)

    def shard_bounds(self, world_size: int, rank: int) -> tp.Tuple[int, int]:
        n = self.flat_p.numel() // world_size
        return rank * n, (rank + 1) * n

    def step_sharded(self, world_size: int, rank: int) -> None:
        """The exchange that scales when the gradient is dominated by a large table: reduce-scatter the flat gradient —
        every rank receives the SUM of its 1/world slice only —, Adam on that slice of (p, m, v) (the moments of the other slices are
        never touched on this rank), all-gather of the updated parameter slices.  Same link traffic as the all-reduce
        (2 (N-1)/N of the buffer), 1/N of the 5-stream optimiser pass per GPU.  Replicas stay bit-identical: every rank receives the
        same parameter bytes.  gloo (CPU tests, ranks sharing a GPU) has no reduce-scatter: all-reduce + slice there."""
        import torch.distributed as dist

        n_flat = self.flat_p.numel()
        if world_size < 1 or n_flat % world_size != 0 or (n_flat // world_size) % 4 != 0:
            raise ValueError(f"sharded exchange: world size {world_size} does not cut the flat buffer ({self.flat_p.numel()} floats) into "
                             f"equal 16-byte aligned slices")
        if self.partial_moments is not None and self.partial_moments != (world_size, rank):
            raise RuntimeError(f"sharded exchange: the moments on this rank are current for (world, rank) = {self.partial_moments} only; "
                               f"call consolidate_moments() on every rank before changing the process group")
        self.partial_moments = (world_size, rank)
        lo, hi = self.shard_bounds(world_size, rank)
        fg = self.gather_gradients()
        if self._g_shard is None or self._g_shard.numel() != hi - lo:
            self._g_shard = torch.empty(hi - lo, dtype=torch.float32, device=fg.device)
        if self.exchange is not None:
            self.exchange.reduce_scatter(fg, self._g_shard)
        elif dist.get_backend() == "nccl":
            dist.reduce_scatter_tensor(self._g_shard, fg, op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(fg, op=dist.ReduceOp.SUM)
            self._g_shard.copy_(fg[lo:hi])
        self.step_count += 1
        hyper = (self.step_count, float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), 1.0 / world_size)
        self._adam_flat(self.flat_p[lo:hi], self._g_shard, self.m[lo:hi], self.v[lo:hi], hyper)
        if self.exchange is not None:
            self.exchange.all_gather(self.flat_p[lo:hi], self.flat_p)
        elif dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(self.flat_p, self.flat_p[lo:hi].clone())
        else:
            parts = [torch.empty(hi - lo, dtype=torch.float32, device=fg.device) for _ in range(world_size)]
            dist.all_gather(parts, self.flat_p[lo:hi].clone())
            for r, part in enumerate(parts):
                self.flat_p[r * (hi - lo):(r + 1) * (hi - lo)].copy_(part)

    def _use_sharded(self, world_size: int) -> bool:
        """Does a step over `world_size` ranks take the sharded exchange?  Asked for (`self.sharded`) AND the world cuts the flat
        buffers into equal slices whose bounds are 16-byte aligned; otherwise the all-reduce exchange (always valid) is used."""
        n = self.flat_p.numel()
        return bool(self.sharded) and world_size > 1 and n % world_size == 0 and (n // world_size) % 4 == 0

    def consolidate_moments(self) -> None:
        """COLLECTIVE (every rank of the process group the sharded steps ran in): gather the moment slices so that every rank holds
        the whole of (m, v) again.  `fit()` / `fit_partial()` call it after their last step, while the process group is alive, so that
        saving a checkpoint afterwards is a purely local operation (`if rank == 0: model.save_to_checkpoint(...)` cannot deadlock)."""
        if self.partial_moments is None:
            return
        import torch.distributed as dist

        world_size, rank = self.partial_moments
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() != world_size or dist.get_rank() != rank:
            raise RuntimeError(f"the Adam moments on this rank are partial (slice {rank} of {world_size}, sharded exchange) and the process "
                               f"group they were sharded over is gone: they cannot be gathered any more")
        lo, hi = self.shard_bounds(world_size, rank)
        for buf in (self.m, self.v):
            if dist.get_backend() == "nccl":
                dist.all_gather_into_tensor(buf, buf[lo:hi].clone())
            else:
                parts = [torch.empty(hi - lo, dtype=torch.float32, device=buf.device) for _ in range(world_size)]
                dist.all_gather(parts, buf[lo:hi].clone())
                for r, part in enumerate(parts):
                    buf[r * (hi - lo):(r + 1) * (hi - lo)].copy_(part)
        self.partial_moments = None

    def full_moments(self, world_size: int = 1, rank: int = 0) -> tp.Tuple[torch.Tensor, torch.Tensor]:
        """(m, v) over the whole flat buffer.  After sharded steps a rank holds only its own slice: this gathers the others
        (COLLECTIVE in that case — prefer `consolidate_moments()` at a point every rank reaches) and raises when it cannot."""
        if self.partial_moments is None:
            return self.m, self.v
        self.consolidate_moments()
        return self.m, self.v

    def step(self, world_size: int = 1, flat: bool = False) -> None:
        """One Adam step.  world_size > 1 (or flat=True): pack -> all-reduce -> Adam over the flat buffers (or, `self.sharded`, the
        reduce-scatter / sharded Adam / all-gather exchange); otherwise the segmented kernel reads every gradient through its own
        pointer."""
        if self.flat_p.is_cuda:
            ops.join_side_streams()   # weight gradients may still be in flight on the wgrad stream
        timed = self.exchange_events is not None and self.flat_p.is_cuda and world_size > 1
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self._use_sharded(world_size):
            import torch.distributed as dist

            self.step_sharded(world_size, dist.get_rank())
            if timed:      # (reduce-scatter + sharded Adam + all-gather: the sharded exchange has no seam between them)
                e1.record()
                self.exchange_events.append((e0, e1))
            return
        if self.partial_moments is not None:     # an all-reduce step after sharded ones needs whole moments on every rank
            self.consolidate_moments()
        if self._early is not None:              # the early bucket is in flight (begin_early_exchange): the two-bucket step
            self.step_count += 1
            self._finish_early(world_size, (self.step_count, float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                            1.0 / max(world_size, 1)))
            if timed:
                e1.record()
                self.exchange_events.append((e0, e1))
            return
        flat = flat or world_size > 1
        scale = self.reduce_gradients(world_size, force=flat)
        if timed:
            e1.record()
            self.exchange_events.append((e0, e1))
        self.step_count += 1
        hyper = (self.step_count, float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(scale))
        if flat:
            self._adam_flat(self.flat_p, self.flat_g, self.m, self.v, hyper)
            return
        import ctypes

        grads = []
        for p in self.params:  # parameters without a gradient are skipped, as torch.optim.Adam does
            g = p.grad
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous() or g.device != self.flat_p.device):
                g = g.to(device=self.flat_p.device, dtype=torch.float32).contiguous()
            grads.append(g)
        n = len(self.params)
        offsets = (ctypes.c_int64 * n)(*self._offsets)
        lens = (ctypes.c_int64 * n)(*[p.numel() for p in self.params])
        ptrs = (ctypes.c_void_p * n)(*[None if g is None else g.data_ptr() for g in grads])
        ops._c("rt_adam_step_segments", self.flat_p, self.m, self.v, n, offsets, lens, ptrs, *hyper)

    def state_dict(self) -> tp.Dict[str, tp.Any]:
        if self.partial_moments is not None:
            raise RuntimeError("FlatAdam.state_dict(): the moments are partial (sharded exchange); call consolidate_moments() on every rank first")
        return {"m": self.m.clone(), "v": self.v.clone(), "step": self.step_count, "lr": self.lr, "betas": self.betas,
                "eps": self.eps}

    def load_state_dict(self, sd: tp.Dict[str, tp.Any]) -> None:
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self.step_count = int(sd["step"])
        self.partial_moments = None


class NativeSasrecStep:
    """The stock packed SASRec training step issued by ONE compiled call (`rt_sasrec_step_run`, csrc/rt_step.hip) instead of five
    autograd nodes, ~22 ctypes calls and 18 `torch.empty` per step: same entry points, same order, same streams, same dropout draws
    as `TransformerLossModule.training_loss_packed` + `loss.backward()` + `FlatAdam.step()` (lightning.py:311-321 of the reference) —
    losses and parameters agree to the run-to-run noise of either path (tests/test_native_step_gpu.py), the autograd path is the
    cross-check and what every other configuration runs.  RT_NATIVE_STEP=0 switches it off.

    `plan(lm, opt)` returns None unless the model is exactly what the sequence restates: stock loss module / backbone / similarity,
    `SASRecTransformerLayers` whose blocks take the native executor, an ids-only item net whose table is a leaf parameter, the stock
    positional encoding, a sampled loss (BCE / gBCE / sampled softmax), one rank, every optimised parameter on that path.  The
    gradients live in the step's arena (`p.grad` stays None)."""

    def __init__(self) -> None:
        self.desc: tp.Any = None

    @staticmethod
    def plan(lm: "TransformerLossModule", opt: "FlatAdam") -> tp.Optional["NativeSasrecStep"]:
        import ctypes

        from . import _lib
        from . import nn as hnn

        if os.environ.get("RT_NATIVE_STEP", "1") == "0" or not opt.flat_p.is_cuda or not ops.native_block_enabled():
            return None
        tl_cls, bb_cls = TransformerLossModule, TransformerTorchBackbone
        tm = lm.torch_model
        if (type(lm).training_loss_packed is not tl_cls.training_loss_packed or type(lm)._loss_from_sessions is not tl_cls._loss_from_sessions
                or lm.loss not in ("BCE", "gBCE", "sampled_softmax") or not lm.similarity_is_stock):
            return None
        if (type(tm).encode_packed_train is not bb_cls.encode_packed_train or type(tm)._watch_input_gradient is not bb_cls._watch_input_gradient
                or not tm._fused_pos() or tm.d_real is not None or not tm.use_causal_attn):
            return None
        tl = tm.transformer_layers
        if type(tl).forward_packed_train is not hnn.SASRecTransformerLayers.forward_packed_train or tl.last_layernorm.cols is not None:
            return None
        blocks = list(tl.transformer_blocks)
        if not blocks or len(blocks) > 16:
            return None
        for blk in blocks:
            if (type(blk).forward_packed_train is not hnn.SASRecTransformerLayer.forward_packed_train or not blk.packed_ok()
                    or blk.multi_head_attn.scale != 0.0 or blk.q_layer_norm.cols is not None or blk.ff_layer_norm.cols is not None
                    or blk.p != blocks[0].p or blk.multi_head_attn.n_heads != blocks[0].multi_head_attn.n_heads
                    or blk.feed_forward.ff_linear_1.weight.shape != blocks[0].feed_forward.ff_linear_1.weight.shape):
                return None
        im = tm.item_model
        if type(im) is not hnn.SumOfEmbeddingsConstructor or im._ids_at is None or im._cat_at is not None or im._more_at:
            return None
        table = im.table
        pos = tm.pos_encoding_layer.pos_emb.weight if tm.pos_encoding_layer.pos_emb is not None else None
        if not (isinstance(table, nn.Parameter) and table.is_leaf and table.requires_grad and table.is_contiguous()):
            return None
        lnf = tl.last_layernorm

        def block_params(blk: tp.Any) -> tp.List[torch.Tensor]:
            mha, ff = blk.multi_head_attn, blk.feed_forward
            return [blk.q_layer_norm.weight, blk.q_layer_norm.bias, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias,
                    blk.ff_layer_norm.weight, blk.ff_layer_norm.bias, ff.ff_linear_1.weight, ff.ff_linear_1.bias, ff.ff_linear_2.weight,
                    ff.ff_linear_2.bias]

        role_of: tp.Dict[int, int] = {id(table): 0, id(lnf.weight): 2, id(lnf.bias): 3}
        if pos is not None:
            role_of[id(pos)] = 1
        per_block = [block_params(b) for b in blocks]
        for b, ps in enumerate(per_block):
            for j, p in enumerate(ps):
                role_of[id(p)] = 16 + 12 * b + j
        if len(role_of) != 3 + (pos is not None) + 12 * len(blocks):       # (a parameter shared between two roles)
            return None
        roles = [role_of.get(id(p)) for p in opt.params]
        if any(r is None for r in roles) or len(set(roles)) != len(role_of):  # every optimised parameter is on this path, and all of the path is optimised
            return None
        if any(p.dtype != torch.float32 or not p.is_contiguous() or p.device != opt.flat_p.device for p in opt.params):
            return None
        d = int(table.shape[1])
        me = NativeSasrecStep()
        me.lm, me.opt, me.tm, me.tl, me.blocks, me.table, me.pos, me.lnf = lm, opt, tm, tl, blocks, table, pos, lnf
        me.per_block = per_block
        n = len(opt.params)
        me.seg_offsets = (ctypes.c_int64 * n)(*opt._offsets)
        me.seg_lens = (ctypes.c_int64 * n)(*[p.numel() for p in opt.params])
        me.seg_role = (ctypes.c_int32 * n)(*roles)
        me.blk_arr = (_lib.SasrecBlock * len(blocks))()
        s = me.desc = _lib.SasrecStep()
        s.n_blocks, s.V, s.d, s.dff, s.H = len(blocks), int(table.shape[0]), d, int(per_block[0][8].shape[0]), int(blocks[0].multi_head_attn.n_heads)
        s.pad_keys = int(not tm.use_key_padding_mask)
        s.loss = {"BCE": ops.LOSS_BCE, "gBCE": ops.LOSS_GBCE, "sampled_softmax": ops.LOSS_SAMPLED_SOFTMAX}[lm.loss]
        s.pos_rows = 0 if pos is None else int(pos.shape[0])
        s.eps_last = float(lnf.eps)
        s.blocks = ctypes.addressof(me.blk_arr)
        s.n_seg = n
        s.seg_offsets, s.seg_lens, s.seg_role = ctypes.addressof(me.seg_offsets), ctypes.addressof(me.seg_lens), ctypes.addressof(me.seg_role)
        me.arena: tp.Optional[torch.Tensor] = None
        me._reserve = 0
        me.steps = 0          # steps this plan has issued (bench.py's `roofline.step_issue`)
        me._bytes = getattr(_lib.load(), "rt_sasrec_step_arena_bytes")
        me.upstream = torch.ones((1,), dtype=torch.float32, device=opt.flat_p.device)
        s.upstream = me.upstream.data_ptr()
        me._bound: tp.Optional[tp.Tuple] = None
        me._fn = getattr(_lib.load(), "rt_sasrec_step_run")
        return me

    def _bind(self) -> None:
        """Parameter / optimiser pointers (re-read when the flat buffers or the planes moved: a checkpoint load, `.to()`)."""
        opt, s = self.opt, self.desc
        planes = self.tl._fresh_planes(refresh=False)
        key = (opt.flat_p.data_ptr(), opt.m.data_ptr(), opt.v.data_ptr(), self.table.data_ptr(), None if planes is None else planes.planes.data_ptr())
        if key == self._bound:
            return
        s.table, s.pos = self.table.data_ptr(), (None if self.pos is None else self.pos.data_ptr())
        s.lnf_w, s.lnf_b = self.lnf.weight.data_ptr(), self.lnf.bias.data_ptr()
        s.flat_p, s.adam_m, s.adam_v = opt.flat_p.data_ptr(), opt.m.data_ptr(), opt.v.data_ptr()
        if planes is not None:
            s.planes_src, s.planes_n, s.planes, s.planes_stride = planes.lo, planes.n, planes.planes.data_ptr(), planes.stride
        else:
            s.planes_src, s.planes_n, s.planes, s.planes_stride = None, 0, None, 0
        for b, (blk, ps) in enumerate(zip(self.blocks, self.per_block)):
            c = self.blk_arr[b]
            (c.ln1_w, c.ln1_b, c.in_w, c.in_b, c.out_w, c.out_b, c.ln2_w, c.ln2_b, c.w1, c.b1, c.w2, c.b2) = [t.data_ptr() for t in ps]
            c.eps1, c.eps2 = float(blk.q_layer_norm.eps), float(blk.ff_layer_norm.eps)
            c.in_wp = c.out_wp = c.w1_wp = c.w2_wp = None
            c.wp_stride = 0
            if planes is not None:
                ptrs = [planes.of(ps[i]) for i in (2, 4, 8, 10)]
                if all(q is not None for q in ptrs):
                    c.in_wp, c.out_wp, c.w1_wp, c.w2_wp = ptrs
                    c.wp_stride = planes.stride
        self._planes = planes
        self._bound = key

    def ready(self, batch: Batch) -> bool:
        """Per step: training mode, the default two-stream issue without instrumentation, whole moments, a batch of the stock shape."""
        lm, opt = self.lm, self.opt
        return (lm.training and ops._TIMING is None and ops._side_enabled() and opt.partial_moments is None and opt._early is None
                and batch.get("negatives") is not None and batch.get("n_rows") is not None and batch.get("ts") is None
                and batch.get("n_prefixed") is None and int(batch["x"].shape[0]) % 128 == 0
                and (self.pos is None or int(self.pos.shape[0]) == int(batch["window"])))

    def forward_backward(self, batch: Batch) -> torch.Tensor:
        """Phase 1 (forward, loss, backward) on torch's current stream -> the loss (a device scalar)."""
        from . import _lib

        self._bind()
        s, tm, lm = self.desc, self.tm, self.lm
        x, neg = batch["x"], batch["negatives"]
        rows = int(x.shape[0])
        B = int(batch["cu"].numel()) - 1
        s.rows, s.B, s.window = rows, B, int(batch["window"])
        cu_attn = batch.get("cu_attn")
        if cu_attn is not None:      # the unused tail of the row block rides along as one more session of the attention
            s.cu_attn, s.B_attn, s.rows_real = cu_attn.data_ptr(), B + 1, rows
        else:
            s.cu_attn, s.B_attn, s.rows_real = batch["cu"].data_ptr(), B, int(batch["n_rows"])
        s.ids, s.dist, s.y, s.yw, s.cu = x.data_ptr(), batch["dist"].data_ptr(), batch["y"].data_ptr(), batch["yw"].data_ptr(), batch["cu"].data_ptr()
        if neg.numel() % rows != 0 or not neg.is_contiguous() or not batch["yw"].is_contiguous():
            raise ValueError("native step: negatives / weights are not contiguous [rows, N] / [rows] tensors")
        s.neg, s.n_neg = neg.data_ptr(), neg.numel() // rows
        s.cosine, s.logits_t = int(lm.cosine), float(lm.logits_t)
        s.gbce_beta = gbce_beta(int(s.n_neg), int(s.V) - lm.n_item_extra_tokens, lm.gbce_t) if lm.loss == "gBCE" else 0.0
        s.emb_scale = tm._pos_scale(self.table)
        p_emb = float(tm.dropout_rate) if tm.training else 0.0
        p_blk = float(self.blocks[0].p) if self.blocks[0].training else 0.0
        s.p_emb, s.p_blk = p_emb, p_blk
        # the dropout draws of the autograd nodes, in their order: the lookup, then (attention, hidden, output) of every block
        s.seed_emb, s.sid_emb = ops.RNG.next() if p_emb > 0 else (0, 0)
        for b in range(len(self.blocks)):
            c = self.blk_arr[b]
            if p_blk > 0:
                s0, sid = ops.RNG.next()
                c.seed_attn = (s0 + 0xD1B54A32D192ED03 * sid) & 0xFFFFFFFFFFFFFFFF
                c.seed_h, c.sid_h = ops.RNG.next()
                c.seed_o, c.sid_o = ops.RNG.next()
            else:
                c.seed_attn = c.seed_h = c.sid_h = c.seed_o = c.sid_o = 0
        s.wgrad_splits = ops._wgrad_splits(rows)
        import ctypes

        need = int(self._bytes(ctypes.byref(s)))
        if need <= 0:
            raise _lib.HipLibraryError("rt_sasrec_step_arena_bytes: invalid step description")
        if self.arena is None or need > int(self.arena.numel()):
            # grows to the largest batch met (models._TrainLoop starts with the epoch's largest: `reserve`); the old block goes back to
            # the allocator first — everything that read it was issued on this stream or joined into it (the previous step's Adam)
            self.arena = None
            self.arena = torch.empty((max(need, self._reserve),), dtype=torch.uint8, device=self.opt.flat_p.device)
            s.arena, s.arena_bytes = self.arena.data_ptr(), int(self.arena.numel())
        out = torch.empty((2,), dtype=torch.float32, device=self.opt.flat_p.device)
        s.loss_out = out.data_ptr()
        self._run(1)
        self.steps += 1
        return out[0]

    def gradients(self) -> tp.List[tp.Optional[torch.Tensor]]:
        """The gradients phase 1 left in the arena, one per `opt.params` entry (views; valid until the next `forward_backward`).
        Weight gradients may still be in flight on the library's side stream: this joins it into torch's current stream first.
        For checks against the oracle (tests, `__graft_entry__.smoke`) — the step itself hands the same pointers to Adam."""
        import ctypes

        from . import _lib

        assert self.arena is not None, "no step has run"
        ops._c("rt_side_join")      # (the C call forked the side stream itself: Python's keep-alive lists, which `ops.join_side_streams` consults, are empty)
        n = len(self.opt.params)
        ptrs = (ctypes.c_void_p * n)()
        _lib.check(_lib.load().rt_sasrec_step_grad_ptrs(ctypes.byref(self.desc), ptrs), "rt_sasrec_step_grad_ptrs")
        base = self.arena.data_ptr()
        out: tp.List[tp.Optional[torch.Tensor]] = []
        for p, q in zip(self.opt.params, ptrs):
            if not q:
                out.append(None)
                continue
            off = int(q) - base
            out.append(self.arena[off:off + 4 * p.numel()].view(torch.float32).view_as(p))
        return out

    def adam(self) -> None:
        """Phase 2: join the weight-gradient stream, one segmented Adam launch over the flat buffers."""
        opt, s = self.opt, self.desc
        opt.step_count += 1
        s.adam_step, s.lr, s.beta1, s.beta2, s.adam_eps = opt.step_count, float(opt.lr), float(opt.betas[0]), float(opt.betas[1]), float(opt.eps)
        self._run(2)

    def _run(self, phase: int) -> None:
        import ctypes

        from . import _lib

        _lib.check(self._fn(ctypes.byref(self.desc), phase, _lib.current_stream()), "rt_sasrec_step_run")
