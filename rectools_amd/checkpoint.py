"""Checkpoints in PyTorch Lightning's dict layout — read and written without Lightning.

The reference persists a fitted transformer model as the bytes of `Trainer.save_checkpoint` (`transformers/base.py:
656-676`) and restores it from that dict (`_model_from_checkpoint`, `base.py:591-654`; `load_from_checkpoint`,
`base.py:678-711`; `load_weights_from_checkpoint`, `base.py:713-724`).  The parts of the dict that carry the MODEL are

    "state_dict"        {"torch_model.<parameter or buffer name>": tensor}      (names: SURVEY.md Appendix B)
    "optimizer_states"  [torch.optim.Adam.state_dict()]   state indexed by parameter position, exp_avg / exp_avg_sq / step
    "hyper_parameters"  the lightning module's constructor arguments (`lightning.py:75-121`): model_config,
                        dataset_schema, item_external_ids, item_extra_tokens, lr, gbce_t, loss, verbose,
                        train_loss_name, val_loss_name, adam_betas, logits_t
    "epoch", "global_step"

and those are what this module reads and writes.  "loops" / "callbacks" / "lr_schedulers" hold the state of Lightning's
own Trainer objects; they are ignored when reading, and written empty — a checkpoint written here can be consumed by the
reference through `load_weights_from_checkpoint` / plain `torch.load` + `load_state_dict`, not through a Trainer resume.

Dotted class paths inside `model_config` are translated between the two packages (`rectools.models.nn...` <->
`rectools_amd...`), so a checkpoint of the reference's `SASRecModel` loads into this engine's `SASRecModel` and back.

Parity note: `pytorch_lightning` is not installed in the build image, so the fixture that pins the reader
(tests/golden/ckpt_*.ckpt) is assembled from the reference's own objects — `lightning_model.state_dict()`, its
`torch.optim.Adam.state_dict()` and its constructor arguments — by tests/golden/make_golden_transformer.py, not by
`Trainer.save_checkpoint` itself.
"""
from __future__ import annotations

import typing as tp

import numpy as np
import torch

LIGHTNING_LAYOUT_VERSION = "2.5.0"   # value written under "pytorch-lightning_version"; the reader accepts any

_REF = "rectools.models.nn."
CLASS_PATHS: tp.Dict[str, str] = {
    _REF + "transformers.sasrec.SASRecModel": "rectools_amd.models.SASRecModel",
    _REF + "transformers.bert4rec.BERT4RecModel": "rectools_amd.models.BERT4RecModel",
    _REF + "transformers.hstu.HSTUModel": "rectools_amd.models.HSTUModel",
    "rectools.models.SASRecModel": "rectools_amd.models.SASRecModel",
    "rectools.models.BERT4RecModel": "rectools_amd.models.BERT4RecModel",
    "rectools.models.HSTUModel": "rectools_amd.models.HSTUModel",
    _REF + "transformers.sasrec.SASRecDataPreparator": "rectools_amd.data_preparator.SASRecDataPreparator",
    _REF + "transformers.bert4rec.BERT4RecDataPreparator": "rectools_amd.data_preparator.BERT4RecDataPreparator",
    _REF + "transformers.negative_sampler.CatalogUniformSampler": "rectools_amd.data_preparator.CatalogUniformSampler",
    _REF + "transformers.sasrec.SASRecTransformerLayers": "rectools_amd.nn.SASRecTransformerLayers",
    _REF + "transformers.net_blocks.PreLNTransformerLayers": "rectools_amd.nn.PreLNTransformerLayers",
    _REF + "transformers.ligr.LiGRLayers": "rectools_amd.nn.LiGRLayers",
    _REF + "transformers.hstu.STULayers": "rectools_amd.nn.STULayers",
    _REF + "transformers.similarity.DistanceSimilarityModule": "rectools_amd.nn.DistanceSimilarityModule",
    _REF + "transformers.net_blocks.LearnableInversePositionalEncoding": "rectools_amd.nn.LearnableInversePositionalEncoding",
    _REF + "transformers.torch_backbone.TransformerTorchBackbone": "rectools_amd.nn.TransformerTorchBackbone",
    _REF + "transformers.lightning.TransformerLightningModule": "rectools_amd.lightning.TransformerLossModule",
    _REF + "transformers.utils.leave_one_out_mask": "rectools_amd.utils.leave_one_out_mask",
    _REF + "item_net.IdEmbeddingsItemNet": "rectools_amd.nn.IdEmbeddingsItemNet",
    _REF + "item_net.CatFeaturesItemNet": "rectools_amd.nn.CatFeaturesItemNet",
    _REF + "item_net.SumOfEmbeddingsConstructor": "rectools_amd.nn.SumOfEmbeddingsConstructor",
}
_BACK = {v: k for k, v in CLASS_PATHS.items() if not k.startswith("rectools.models.SASRec") and not k.startswith("rectools.models.BERT")
         and not k.startswith("rectools.models.HSTU")}
STATE_PREFIX = "torch_model."
ENGINE_ONLY_PARAMS = ("seed", "csv_log_dir")     # model constructor arguments the reference's config classes do not have


def translate_config(config: tp.Any, to_reference: bool = False) -> tp.Any:
    """Rewrite every dotted class path of a (nested) model config from one package to the other; unknown paths — user
    classes, trainer factories — are kept as they are."""
    table = _BACK if to_reference else CLASS_PATHS
    if isinstance(config, dict):
        return {k: translate_config(v, to_reference) for k, v in config.items()}
    if isinstance(config, (list, tuple)):
        return type(config)(translate_config(v, to_reference) for v in config)
    if isinstance(config, str):
        return table.get(config, config)
    return config


# ---- torch.optim.Adam state <-> flat moments ----------------------------------------------------------------
def adam_state_dict(opt: tp.Any) -> tp.Dict[str, tp.Any]:
    """`torch.optim.Adam.state_dict()` of a `FlatAdam`: per-parameter exp_avg / exp_avg_sq / step, in parameter order."""
    state: tp.Dict[int, tp.Dict[str, torch.Tensor]] = {}
    if opt.step_count > 0:
        if getattr(opt, "partial_moments", None) is not None:
            # sharded exchange: this rank's moments are current on its own slice only.  fit() / fit_partial() gather them after their
            # last step (FlatAdam.consolidate_moments, collective), so a checkpoint written after training is local; reaching this
            # point means the caller stepped the optimiser by hand — writing the stale slices would corrupt a later fit_partial
            raise RuntimeError("the Adam moments on this rank are partial (sharded data-parallel exchange): call "
                               "model.optimizer.consolidate_moments() on EVERY rank (collective) before saving a checkpoint")
        from .nn import unpad_tensor    # (parameters of a `nn.DimPlan` model carry zero columns: checkpoints speak the real shapes)

        m, v = opt.m, opt.v
        for i, (p, ofs) in enumerate(zip(opt.params, opt._offsets)):   # pylint: disable=protected-access
            n = p.numel()
            state[i] = {"step": torch.tensor(float(opt.step_count)),
                        "exp_avg": unpad_tensor(m[ofs:ofs + n].detach().reshape(p.shape), p).cpu().clone(),
                        "exp_avg_sq": unpad_tensor(v[ofs:ofs + n].detach().reshape(p.shape), p).cpu().clone()}
    group = {"lr": float(opt.lr), "betas": tuple(float(b) for b in opt.betas), "eps": float(opt.eps), "weight_decay": 0,
             "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
             "fused": None, "params": list(range(len(opt.params)))}
    return {"state": state, "param_groups": [group]}


def load_adam_state_dict(opt: tp.Any, sd: tp.Dict[str, tp.Any], names: tp.Optional[tp.Sequence[str]] = None,
                         opt_names: tp.Optional[tp.Sequence[str]] = None) -> None:
    """Fill a `FlatAdam`'s moments from a `torch.optim.Adam.state_dict()`.

    torch indexes the state by parameter POSITION.  `names` = the parameter names in the writer's order (the order of
    the parameter keys in the checkpoint's own state_dict — torch walks modules identically for `state_dict()` and
    `parameters()`), `opt_names` = this optimiser's parameter names: the state is then matched by name.  Without names
    both orders are assumed equal."""
    from .nn import pad_tensor

    groups = sd.get("param_groups", [])
    order = [i for g in groups for i in g["params"]] or sorted(sd["state"])
    if len(order) != len(opt.params):
        raise ValueError(f"optimizer state holds {len(order)} parameters, the model has {len(opt.params)}")
    if names is not None and opt_names is not None:
        if sorted(names) != sorted(opt_names):
            raise ValueError("optimizer state: parameter names of the checkpoint and of the model differ")
        where = {n: i for i, n in enumerate(opt_names)}
        target = [where[n] for n in names]
    else:
        target = list(range(len(order)))
    steps = []
    opt.m.zero_(); opt.v.zero_()
    for pos, key in enumerate(order):
        st = sd["state"].get(key)
        if st is None:
            continue
        p, ofs = opt.params[target[pos]], opt._offsets[target[pos]]   # pylint: disable=protected-access
        real_shape = tuple(getattr(p, "_rt_real_shape", p.shape))
        if tuple(st["exp_avg"].shape) != real_shape:
            raise ValueError(f"optimizer state {key}: shape {tuple(st['exp_avg'].shape)} != parameter shape {real_shape}")
        opt.m[ofs:ofs + p.numel()].copy_(pad_tensor(st["exp_avg"], p).reshape(-1))
        opt.v[ofs:ofs + p.numel()].copy_(pad_tensor(st["exp_avg_sq"], p).reshape(-1))
        steps.append(int(float(st["step"])))
    # one step counter for the whole model: torch keeps one per parameter, equal unless a parameter never got a gradient
    opt.step_count = max(steps) if steps else 0
    opt.partial_moments = None     # whole moments were just written on this rank
    if groups:
        opt.lr = float(groups[0].get("lr", opt.lr))
        opt.betas = tuple(groups[0].get("betas", opt.betas))
        opt.eps = float(groups[0].get("eps", opt.eps))


# ---- writer ---------------------------------------------------------------------------------------------------
def _plain(v: tp.Any) -> tp.Any:
    return v.item() if isinstance(v, np.generic) else v


def to_checkpoint(model: tp.Any, reference_paths: bool = True) -> tp.Dict[str, tp.Any]:
    """Lightning-layout checkpoint dict of a fitted model (tensors on CPU).  reference_paths=True writes the
    reference's class paths into `model_config`, which is what the reference's `from_config` can import."""
    lm, opt, dp = model.lightning_model, model.optimizer, model.data_preparator
    if lm is None or opt is None:
        raise RuntimeError("only a fitted (or built) model has a checkpoint")
    config = model.get_config(simple_types=True)
    engine_params = {}
    if reference_paths:
        config = translate_config(config, to_reference=True)
        # the reference's config classes forbid unknown fields (`extra="forbid"`, models/base.py:74-80): constructor arguments only this
        # engine knows travel beside the checkpoint's own keys, not inside `model_config` — found by the reference actually loading an
        # engine-written file (tests/test_reference_live.py): `seed` made `load_from_checkpoint` fail validation
        for key in ENGINE_ONLY_PARAMS:
            if key in config:
                engine_params[key] = config.pop(key)
    hyper = {
        "model_config": config, "dataset_schema": model.dataset_schema,
        # the array itself, as Lightning's `save_hyperparameters` keeps what `_init_lightning_model` was handed (base.py:459-473): the
        # reference's loader feeds it straight to `IdMap(item_external_ids)`, which needs an ndarray (a list has no `.size`)
        "item_external_ids": np.array(dp.item_id_map.external_ids, copy=True),
        "item_extra_tokens": tuple(dp.item_extra_tokens), "lr": model.lr, "gbce_t": model.gbce_t, "loss": model.loss,
        "verbose": model.verbose, "train_loss_name": model.train_loss_name, "val_loss_name": model.val_loss_name,
        "adam_betas": tuple(opt.betas), "logits_t": lm.logits_t,
    }
    return {
        "epoch": int(model.epochs_done), "global_step": int(opt.step_count), "pytorch-lightning_version": LIGHTNING_LAYOUT_VERSION,
        "state_dict": {STATE_PREFIX + k: v.detach().cpu().clone() for k, v in lm.torch_model.state_dict().items()},
        "loops": {}, "callbacks": {}, "optimizer_states": [adam_state_dict(opt)], "lr_schedulers": [],
        "hparams_name": "kwargs", "hyper_parameters": hyper,
        "rectools_amd": {"history": list(model.history), "model_params": engine_params},     # extra key: Lightning ignores what it does not know
    }


# ---- reader ---------------------------------------------------------------------------------------------------
def item_net_schema(dataset_schema: tp.Dict[str, tp.Any]) -> tp.List[tp.Dict[str, tp.Any]]:
    """Shapes of the item-net blocks a dataset schema implies (`from_dataset_schema`, item_net.py:193-228,283-300):
    id embeddings always; a category block when the item features are sparse with categorical columns."""
    blocks: tp.List[tp.Dict[str, tp.Any]] = [{"kind": "ids"}]
    feats = (dataset_schema.get("items") or {}).get("features")
    if feats and feats.get("kind") == "sparse" and len(feats.get("cat_feature_indices", [])) > 0:
        blocks.append({"kind": "cat", "nnz": int(feats["cat_n_stored_values"]),
                       "n_cat_feature_values": len(feats["cat_feature_indices"])})
    return blocks


def strip_state_dict(state_dict: tp.Dict[str, torch.Tensor]) -> tp.Dict[str, torch.Tensor]:
    out = {}
    for k, v in state_dict.items():
        if not k.startswith(STATE_PREFIX):
            raise KeyError(f"unexpected key {k!r} in checkpoint state_dict (expected the '{STATE_PREFIX}' prefix)")
        out[k[len(STATE_PREFIX):]] = v
    return out
