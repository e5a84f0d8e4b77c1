"""Golden vectors for the transformer fit() hot path, produced by the UNMODIFIED reference modules
(`TransformerTorchBackbone`, `SASRecTransformerLayers`, `PreLNTransformerLayers`, `LiGRLayers`, `STULayers`,
`DistanceSimilarityModule`, `TransformerLightningModule` losses, `torch.optim.Adam`).

Called from tests/golden/make_golden.py (which installs the import shims first).  For every variant we store
the config, the full state_dict (reference parameter names), one training batch, and the
reference's logits / loss / per-parameter gradients / parameters after one Adam step / last-slot session
embeddings in eval mode.  dropout_rate = 0 everywhere (dropout streams cannot match).
"""
from __future__ import annotations

import json
import os
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _variants():
    base = dict(V=50, B=4, L=8, d=16, H=2, n_blocks=2, N=3, loss="softmax", dist="dot", logits_t=1.0,
                causal=True, keypad=False, layers="sasrec", n_extra=1, gbce_t=0.2, lr=1e-3, use_scale=False,
                layer_kwargs={}, weights="ones", seed=32)
    out = {}

    def add(name, **kw):
        c = dict(base)
        c.update(kw)
        out[name] = c

    add("sasrec_softmax_dot")
    add("sasrec_softmax_cos", dist="cosine", weights="rand")
    add("sasrec_bce_dot", loss="BCE", weights="rand")
    add("sasrec_gbce_dot", loss="gBCE")
    add("sasrec_gbce_cos", loss="gBCE", dist="cosine")
    add("sasrec_sampled_dot", loss="sampled_softmax", N=5)
    add("sasrec_sampled_cos_t", loss="sampled_softmax", dist="cosine", logits_t=0.05)
    add("sasrec_keypad_causal", keypad=True)
    add("sasrec_mid", V=300, B=5, L=40, d=64, H=4, loss="sampled_softmax", N=7, store_logits=True)
    add("bert4rec_softmax", layers="preln", causal=False, keypad=True, n_extra=2)
    add("bert4rec_bce", layers="preln", causal=False, keypad=True, n_extra=2, loss="BCE")
    add("ligr_swiglu_sampled", layers="ligr", loss="sampled_softmax",
        layer_kwargs=dict(ff_factors_multiplier=4, ff_activation="swiglu", bias_in_ff=False))
    add("ligr_gelu_softmax", layers="ligr", layer_kwargs=dict(ff_factors_multiplier=2, ff_activation="gelu",
                                                               bias_in_ff=True))
    add("ligr_relu_keypad", layers="ligr", keypad=True, layer_kwargs=dict(ff_factors_multiplier=4,
                                                                          ff_activation="relu", bias_in_ff=False))
    for rt, rp in ((True, True), (True, False), (False, True), (False, False)):
        add(f"hstu_time{int(rt)}_pos{int(rp)}", layers="stu", dist="cosine", loss="sampled_softmax", N=4,
            use_scale=True, logits_t=0.05, rel_time=rt, rel_pos=rp)
    add("hstu_softmax_dot", layers="stu", dist="dot", loss="softmax", use_scale=True, rel_time=True, rel_pos=True)
    # feature-aware item net: IdEmbeddingsItemNet + CatFeaturesItemNet (item_net.py:60-233), F (feature, value) pairs
    add("sasrec_catfeat_sampled", loss="sampled_softmax", N=5, cat=dict(F=13, max_per_item=4))
    add("sasrec_catfeat_softmax_cos", dist="cosine", weights="rand", cat=dict(F=7, max_per_item=3))
    add("bert4rec_catfeat_bce", layers="preln", causal=False, keypad=True, n_extra=2, loss="BCE", cat=dict(F=9, max_per_item=5))
    add("sasrec_catfeat_mid", V=300, B=5, L=40, d=64, H=4, loss="gBCE", N=7, cat=dict(F=40, max_per_item=6))
    return out


def make_cat_structure(cfg, gen: torch.Generator):
    """Item -> category-value structure in the layout CatFeaturesItemNet.from_dataset produces (item_net.py:171-181):
    CSR over the catalog, extra-token rows empty (data_preparator.py:203-208), indices ascending inside a row.  Some
    items carry no category at all; value 0 is popular (tags every third item)."""
    n_tokens, F, mx = cfg["V"] + cfg["n_extra"], cfg["cat"]["F"], cfg["cat"]["max_per_item"]
    rows = []
    for i in range(n_tokens):
        if i < cfg["n_extra"]:
            rows.append([])
            continue
        n = int(torch.randint(0, mx + 1, (1,), generator=gen))
        vals = set(torch.randint(0, F, (n,), generator=gen).tolist())
        if i % 3 == 0:
            vals.add(0)
        rows.append(sorted(vals))
    lens = torch.tensor([len(r) for r in rows], dtype=torch.long)
    offsets = torch.cumsum(lens, 0) - lens
    inputs = torch.tensor([v for r in rows for v in r], dtype=torch.long)
    return inputs, lens, offsets


def build_reference(cfg):
    """Instantiate the reference torch modules directly (no Dataset needed)."""
    from rectools.models.nn.item_net import CatFeaturesItemNet, IdEmbeddingsItemNet, SumOfEmbeddingsConstructor
    from rectools.models.nn.transformers.hstu import STULayers
    from rectools.models.nn.transformers.lightning import TransformerLightningModule
    from rectools.models.nn.transformers.ligr import LiGRLayers
    from rectools.models.nn.transformers.net_blocks import LearnableInversePositionalEncoding, PreLNTransformerLayers
    from rectools.models.nn.transformers.sasrec import SASRecTransformerLayers
    from rectools.models.nn.transformers.similarity import DistanceSimilarityModule
    from rectools.models.nn.transformers.torch_backbone import TransformerTorchBackbone

    n_tokens = cfg["V"] + cfg["n_extra"]
    blocks = [IdEmbeddingsItemNet(cfg["d"], n_tokens, 0.0)]
    if cfg.get("cat"):
        inputs, lens, offsets = make_cat_structure(cfg, torch.Generator().manual_seed(cfg["seed"] + 7))
        blocks.append(CatFeaturesItemNet(emb_bag_inputs=inputs, input_lengths=lens, offsets=offsets,
                                         n_cat_feature_values=cfg["cat"]["F"], n_factors=cfg["d"], dropout_rate=0.0))
    item_model = SumOfEmbeddingsConstructor(n_tokens, blocks)
    pos = LearnableInversePositionalEncoding(True, cfg["L"], cfg["d"], use_scale_factor=cfg["use_scale"])
    kind = cfg["layers"]
    if kind == "sasrec":
        layers = SASRecTransformerLayers(n_blocks=cfg["n_blocks"], n_factors=cfg["d"], n_heads=cfg["H"], dropout_rate=0.0)
    elif kind == "preln":
        layers = PreLNTransformerLayers(n_blocks=cfg["n_blocks"], n_factors=cfg["d"], n_heads=cfg["H"], dropout_rate=0.0)
    elif kind == "ligr":
        layers = LiGRLayers(n_blocks=cfg["n_blocks"], n_factors=cfg["d"], n_heads=cfg["H"], dropout_rate=0.0,
                            **cfg["layer_kwargs"])
    elif kind == "stu":
        hd = cfg["d"] // cfg["H"]
        layers = STULayers(n_blocks=cfg["n_blocks"], n_factors=cfg["d"], n_heads=cfg["H"], linear_hidden_dim=cfg.get("linear_hidden_dim", hd),
                           attention_dim=cfg.get("attention_dim", hd), session_max_len=cfg["L"], relative_time_attention=cfg["rel_time"],
                           relative_pos_attention=cfg["rel_pos"], attn_dropout_rate=0.0, dropout_rate=0.0)
    else:
        raise ValueError(kind)
    sim = DistanceSimilarityModule(distance=cfg["dist"])
    backbone = TransformerTorchBackbone(
        n_heads=cfg["H"], dropout_rate=0.0, item_model=item_model, pos_encoding_layer=pos,
        transformer_layers=layers, similarity_module=sim, use_causal_attn=cfg["causal"],
        use_key_padding_mask=cfg["keypad"],
    )
    extra = ["PAD"] if cfg["n_extra"] == 1 else ["PAD", "MASK"]
    dp = types.SimpleNamespace(n_negatives=cfg["N"], item_extra_tokens=extra)
    lm = TransformerLightningModule(
        torch_model=backbone, model_config={}, dataset_schema={}, item_external_ids=[], item_extra_tokens=extra,
        data_preparator=dp, lr=cfg["lr"], gbce_t=cfg["gbce_t"], loss=cfg["loss"], logits_t=cfg["logits_t"],
    )
    return lm


def make_batch(cfg, gen: torch.Generator):
    B, L, V, ne = cfg["B"], cfg["L"], cfg["V"], cfg["n_extra"]
    x = torch.zeros(B, L, dtype=torch.int64)
    y = torch.zeros(B, L, dtype=torch.int64)
    lens = torch.randint(1, L + 1, (B,), generator=gen)
    lens[0] = L  # one full row
    if B > 1:
        lens[1] = 1  # one nearly-empty row
    for b in range(B):
        n = int(lens[b])
        seq = torch.randint(ne, V + ne, (n + 1,), generator=gen)
        x[b, L - n:] = seq[:-1]
        y[b, L - n:] = seq[1:]
    if cfg["layers"] == "preln":
        # BERT4Rec-style batch: some inputs replaced by MASK (id 1), targets only at masked slots
        x2, y2 = x.clone(), torch.zeros_like(y)
        for b in range(B):
            n = int(lens[b])
            full = torch.randint(ne, V + ne, (n,), generator=gen)
            m = torch.rand(n, generator=gen) < 0.4
            m[-1] = True
            inp = full.clone()
            inp[m] = 1
            x2[b] = 0
            x2[b, L - n:] = inp
            y2[b, L - n:] = torch.where(m, full, torch.zeros_like(full))
        x, y = x2, y2
    if cfg["weights"] == "ones":
        yw = (y != 0).float()
    else:
        yw = (y != 0).float() * (0.5 + torch.rand(B, L, generator=gen))
    batch = {"x": x, "y": y, "yw": yw}
    if cfg["loss"] != "softmax":
        batch["negatives"] = torch.randint(ne, V + ne, (B, L, cfg["N"]), generator=gen)
    if cfg["layers"] == "stu":
        gaps = torch.randint(1, 10_000_000, (B, L + 1), generator=gen)
        gaps[:, 2] = 0  # a zero gap (bucket of max(1,|0|) -> 0)
        ts = torch.cumsum(gaps, dim=1) + 1_500_000_000
        # left-fill padded slots with the first real timestamp (sasrec.py:109-116)
        for b in range(B):
            n = int(lens[b])
            ts[b, : L - n] = ts[b, L - n]
        batch["unix_ts"] = ts
    return batch


def make_transformer(only: str = "") -> None:
    """`only`: substring filter on the variant name (existing fixtures are not rewritten when new variants are added)."""
    from oracle import ref_shims

    for name, cfg in _variants().items():
        if only and only not in name:
            continue
        ref_shims.seed_all(cfg["seed"])
        lm = build_reference(cfg)
        lm._xavier_normal_init()
        # make LayerNorm / bias parameters non-trivial so their gradients and use are exercised
        g = torch.Generator().manual_seed(cfg["seed"] + 1)
        with torch.no_grad():
            for n, p in lm.torch_model.named_parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
        batch = make_batch(cfg, g)
        state0 = {k: v.detach().clone() for k, v in lm.torch_model.state_dict().items()}

        lm.train()
        b_in = {k: v.clone() for k, v in batch.items()}
        logits = lm.get_batch_logits(b_in).detach().clone()
        opt = lm.configure_optimizers()
        opt.zero_grad()
        loss = lm.training_step({k: v.clone() for k, v in batch.items()}, 0)
        loss.backward()
        grads = {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
                 for n, p in lm.torch_model.named_parameters()}
        opt.step()
        state1 = {k: v.detach().clone() for k, v in lm.torch_model.state_dict().items()}
        # second step on the same batch (exercises Adam moments beyond the first update)
        opt.zero_grad()
        loss2 = lm.training_step({k: v.clone() for k, v in batch.items()}, 1)
        loss2.backward()
        opt.step()
        state2 = {k: v.detach().clone() for k, v in lm.torch_model.state_dict().items()}

        # eval-mode session embeddings with the ORIGINAL weights (recommend path, lightning.py:392-396)
        lm.torch_model.load_state_dict(state0)
        lm.torch_model.eval()
        with torch.no_grad():
            item_embs = lm.torch_model.item_model.get_all_embeddings()
            enc = lm.torch_model.encode_sessions({k: v.clone() for k, v in batch.items()}, item_embs)

        out = {"config": np.array(json.dumps(cfg))}
        for k, v in state0.items():
            out["p0/" + k] = v.numpy()
        for k, v in grads.items():
            out["g/" + k] = v.numpy()
        for k, v in state1.items():
            out["p1/" + k] = v.numpy()
        for k, v in state2.items():
            out["p2/" + k] = v.numpy()
        for k, v in batch.items():
            out["b/" + k] = v.numpy()
        if logits.numel() <= 20000 or cfg.get("store_logits"):
            out["logits"] = logits.numpy()
        out["loss"] = np.array(loss.item(), dtype=np.float64)
        out["loss2"] = np.array(loss2.item(), dtype=np.float64)
        out["enc"] = enc.numpy()
        np.savez_compressed(os.path.join(HERE, f"transformer_{name}.npz"), **out)
        print(f"transformer {name}: loss={loss.item():.6f} loss2={loss2.item():.6f} params={len(state0)}")


def make_collate() -> None:
    """Collate known-answers: the reference's SASRec / BERT4Rec collate functions on seeded inputs."""
    from rectools.models.nn.transformers.bert4rec import BERT4RecDataPreparator
    from rectools.models.nn.transformers.negative_sampler import CatalogUniformSampler
    from rectools.models.nn.transformers.sasrec import SASRecDataPreparator

    from oracle import ref_shims

    out = {}
    sessions = [
        ([3, 5, 2, 7, 9, 4, 6], [1.0, 1.0, 2.0, 1.0, 1.0, 0.5, 1.0], [10, 20, 30, 40, 50, 60, 70]),
        ([8, 2], [1.0, 3.0], [100, 200]),
        ([4, 4, 6, 5], [1.0, 1.0, 1.0, 1.0], [5, 6, 7, 8]),
    ]
    L = 5
    for n_neg, tag in ((None, "sasrec_noneg"), (2, "sasrec_neg2")):
        for ts in (False, True):
            ref_shims.seed_all(32)
            dp = SASRecDataPreparator(
                session_max_len=L, batch_size=4, dataloader_num_workers=0, n_negatives=n_neg,
                negative_sampler=CatalogUniformSampler(n_negatives=n_neg) if n_neg else None, add_unix_ts=ts,
            )
            dp.item_id_map = types.SimpleNamespace(size=12)
            sfx = "_ts" if ts else ""
            # train sessions carry L+1 items at most (train_session_max_len_addition = 1, sasrec.py:84)
            batch = [(s[-(L + 1):], w[-(L + 1):], {"unix_ts": t[-(L + 1):]}) for s, w, t in sessions]
            b = dp._collate_fn_train(batch)
            for k, v in b.items():
                out[f"{tag}{sfx}/train/{k}"] = v.numpy()
            # recommend sessions: history (+ dummy target item when timestamps are used, sasrec.py:154-163)
            if ts:
                rb = [(s + [0], w + [0.0], {"unix_ts": t + [t[-1] + 5]}) for s, w, t in sessions]
            else:
                rb = [(s, w, {}) for s, w, t in sessions]
            br = dp._collate_fn_recommend(rb)
            for k, v in br.items():
                out[f"{tag}{sfx}/recommend/{k}"] = v.numpy()
    # SASRec validation collate (sasrec.py:118-147): zero-weight interactions are the input, the first weighted one the target
    val_sessions = [
        ([3, 5, 2, 7, 9, 4, 6], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 2.0], list(range(10, 80, 10))),   # L + 2 items: the longest a val session gets
        ([8, 2], [0.0, 1.0], [100, 200]),
        ([4, 4, 6, 5, 9], [0.0, 0.0, 0.0, 3.0, 1.0], [5, 6, 7, 8, 9]),
    ]
    for ts in (False, True):
        ref_shims.seed_all(32)
        dp = SASRecDataPreparator(session_max_len=L, batch_size=4, dataloader_num_workers=0, n_negatives=None, add_unix_ts=ts)
        dp.item_id_map = types.SimpleNamespace(size=12)
        b = dp._collate_fn_val([(s, w, {"unix_ts": t}) for s, w, t in val_sessions])
        for k, v in b.items():
            out[f"sasrec_val{'_ts' if ts else ''}/{k}"] = v.numpy()
    out["val_sessions"] = np.array(json.dumps(val_sessions))

    # BERT4Rec (bert4rec.py:109-193): masking draws come from np.random, in the order rand(len) per session then one
    # randint per randomly replaced element — a seeded restatement must land on the same batch
    from rectools.models.nn.transformers.bert4rec import MASKING_VALUE
    from rectools.models.nn.transformers.constants import PADDING_VALUE

    bert_sessions = [
        ([3, 5, 2, 7, 9], [1.0, 1.0, 2.0, 1.0, 0.5]),
        ([8, 2], [1.0, 3.0]),
        ([4, 4, 6, 5, 9], [1.0, 1.0, 1.0, 1.0, 1.0]),
        ([10, 11, 7], [2.0, 1.0, 1.0]),
    ]
    for mask_prob, seed in ((0.5, 32), (0.9, 7)):
        dp = BERT4RecDataPreparator(session_max_len=L, batch_size=4, dataloader_num_workers=0, n_negatives=None,
                                    train_min_user_interactions=2, mask_prob=mask_prob)
        dp.item_id_map = types.SimpleNamespace(size=12)
        dp.extra_token_ids = {PADDING_VALUE: 0, MASKING_VALUE: 1}
        ref_shims.seed_all(seed)
        b = dp._collate_fn_train([(list(s), list(w), {}) for s, w in bert_sessions])
        for k, v in b.items():
            out[f"bert4rec_p{mask_prob}_s{seed}/train/{k}"] = v.numpy()
    dp = BERT4RecDataPreparator(session_max_len=L, batch_size=4, dataloader_num_workers=0, n_negatives=None,
                                train_min_user_interactions=2, mask_prob=0.5)
    dp.item_id_map = types.SimpleNamespace(size=12)
    dp.extra_token_ids = {PADDING_VALUE: 0, MASKING_VALUE: 1}
    long_sessions = bert_sessions + [([2, 3, 4, 5, 6, 7, 8], [1.0] * 7)]
    b = dp._collate_fn_recommend([(list(s), list(w), {}) for s, w in long_sessions])
    out["bert4rec/recommend/x"] = b["x"].numpy()
    b = dp._collate_fn_val([(list(s), list(w), {}) for s, w, _ in val_sessions])
    for k, v in b.items():
        out[f"bert4rec/val/{k}"] = v.numpy()
    out["bert_sessions"] = np.array(json.dumps(bert_sessions))
    out["bert_long_sessions"] = np.array(json.dumps(long_sessions))
    out["sessions"] = np.array(json.dumps(sessions))
    out["L"] = np.array(L)
    np.savez_compressed(os.path.join(HERE, "collate_golden.npz"), **out)
    print("collate: ok", sorted(k for k in out if k.startswith("bert") or "val" in k))


# ------------------------------------------------------------------------------------------------------------------
# Checkpoint fixtures: the reference's models trained through the shimmed Trainer, stored in the
# layout of `Trainer.save_checkpoint` — assembled from the reference's own objects because pytorch_lightning itself is
# not installed: state_dict = lightning_model.state_dict(), optimizer_states = [its torch.optim.Adam.state_dict()],
# hyper_parameters = the constructor arguments `save_hyperparameters(ignore=[torch_model, data_preparator])` records
# (lightning.py:75-121).  "expected" (not a Lightning key) holds the reference's own recommend() / recommend_to_items()
# frames for the loaded model to reproduce.
# ------------------------------------------------------------------------------------------------------------------
def checkpoint_frames():
    import pandas as pd

    interactions = pd.DataFrame(
        [[10, 13, 1, "2021-11-30"], [10, 11, 1, "2021-11-29"], [10, 12, 1, "2021-11-29"], [30, 11, 1, "2021-11-27"],
         [30, 12, 2, "2021-11-26"], [30, 15, 1, "2021-11-25"], [40, 11, 1, "2021-11-25"], [40, 17, 1, "2021-11-26"],
         [50, 16, 1, "2021-11-25"], [10, 14, 1, "2021-11-28"], [10, 16, 1, "2021-11-27"], [20, 13, 9, "2021-11-28"]],
        columns=["user_id", "item_id", "weight", "datetime"])
    features = pd.DataFrame(
        [[11, "f1", "f1val1"], [11, "f2", "f2val1"], [12, "f1", "f1val1"], [12, "f2", "f2val2"], [13, "f1", "f1val1"],
         [13, "f2", "f2val3"], [11, "f3", 0], [12, "f3", 1], [13, "f3", 2], [16, "f3", 6], [14, "f2", "f2val1"], [17, "f2", "f2val3"]],
        columns=["id", "feature", "value"])
    return interactions, features


def make_checkpoints() -> None:
    import pandas as pd
    from rectools.dataset import Dataset
    from rectools.dataset.context import get_context
    from rectools.models import BERT4RecModel, HSTUModel, SASRecModel
    from rectools.models.nn.item_net import IdEmbeddingsItemNet
    from rectools.models.nn.transformers.ligr import LiGRLayers

    from oracle import ref_shims

    interactions, features = checkpoint_frames()
    cases = {
        "sasrec_catfeat": (SASRecModel, dict(n_factors=32, n_blocks=2, n_heads=2, session_max_len=4, lr=0.01, batch_size=4, epochs=3,
                                             loss="sampled_softmax", n_negatives=3, use_key_padding_mask=True, deterministic=True),
                           True),
        "bert4rec_ids": (BERT4RecModel, dict(n_factors=32, n_blocks=1, n_heads=2, session_max_len=4, lr=0.01, batch_size=4, epochs=2,
                                             loss="softmax", mask_prob=0.5, deterministic=True,
                                             item_net_block_types=(IdEmbeddingsItemNet,)), False),
        "hstu_time_pos": (HSTUModel, dict(n_factors=32, n_blocks=2, n_heads=2, session_max_len=4, lr=0.01, batch_size=4, epochs=3,
                                          loss="sampled_softmax", n_negatives=3, deterministic=True, relative_time_attention=True,
                                          relative_pos_attention=True, item_net_block_types=(IdEmbeddingsItemNet,)), False),
        "esasrec_ligr": (SASRecModel, dict(n_factors=32, n_blocks=2, n_heads=2, session_max_len=4, lr=0.01, batch_size=4, epochs=2,
                                           loss="gBCE", n_negatives=2, deterministic=True, transformer_layers_type=LiGRLayers,
                                           transformer_layers_kwargs=dict(ff_factors_multiplier=4, ff_activation="swiglu", bias_in_ff=False),
                                           item_net_block_types=(IdEmbeddingsItemNet,)), False),
    }
    users = [10, 30, 40]
    context_df = pd.DataFrame({"user_id": users, "datetime": ["2021-12-12", "2021-12-13", "2021-12-12"]})
    for name, (klass, kw, with_features) in cases.items():
        ref_shims.seed_all(32)
        ds = (Dataset.construct(interactions, item_features_df=features, cat_item_features=["f1", "f2"]) if with_features
              else Dataset.construct(interactions))
        model = klass(**kw)
        model.fit(ds)
        lm = model.lightning_model
        hyper = dict(model_config=lm.model_config, dataset_schema=lm.dataset_schema, item_external_ids=list(lm.item_external_ids),
                     item_extra_tokens=tuple(lm.item_extra_tokens), lr=lm.lr, gbce_t=lm.gbce_t, loss=lm.loss, verbose=lm.verbose,
                     train_loss_name=lm.train_loss_name, val_loss_name=lm.val_loss_name, adam_betas=tuple(lm.adam_betas),
                     logits_t=lm.logits_t)
        n_steps = int(next(iter(lm.optimizer.state_dict()["state"].values()))["step"])
        expected = {}
        for tag, rk in (("filter", dict(k=3, filter_viewed=True)), ("nofilter", dict(k=4, filter_viewed=False)),
                        ("whitelist", dict(k=2, filter_viewed=False, items_to_recommend=[11, 13, 17]))):
            ctx = get_context(context_df) if model.require_recommend_context else None
            r = model.recommend(users=users, dataset=ds, context=ctx, **rk)
            expected[tag] = {c: r[c].tolist() for c in r.columns}
        i2i = model.recommend_to_items(target_items=[11, 12], dataset=ds, k=2)
        expected["i2i"] = {c: i2i[c].tolist() for c in i2i.columns}
        ckpt = {
            "epoch": kw["epochs"], "global_step": n_steps, "pytorch-lightning_version": "shimmed",
            "state_dict": {k: v.detach().clone() for k, v in lm.state_dict().items()},
            "loops": {}, "callbacks": {}, "optimizer_states": [lm.optimizer.state_dict()], "lr_schedulers": [],
            "hparams_name": "kwargs", "hyper_parameters": hyper, "expected": expected,
        }
        torch.save(ckpt, os.path.join(HERE, f"ckpt_{name}.ckpt"))
        print(f"checkpoint {name}: {len(ckpt['state_dict'])} tensors, {n_steps} optimizer steps, "
              f"items {hyper['item_external_ids']}, reco {expected['filter']['item_id']}")
