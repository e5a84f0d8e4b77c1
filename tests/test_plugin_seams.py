"""The reference's plug-in seams (transformers/base.py:215-222,278-286,368-380,407,430,449): `pos_encoding_type`, `backbone_type`,
`lightning_module_type` are instantiated from the class handed over (as `transformer_layers_type` / `negative_sampler_type` are),
`get_trainer_func` is read as a plan for the engine's own loop (epoch counts, logger directory, user-defined callbacks); all of it
round-trips through `get_config` / `from_config`."""
import types
import warnings

import numpy as np
import pandas as pd
import pytest
import torch

from rectools_amd import lightning as hl
from rectools_amd import nn as hnn


class SinusoidPositionalEncoding(torch.nn.Module):
    """A user class written against the reference's `PositionalEncodingBase` (net_blocks.py:327-342): constructor
    (use_pos_emb, session_max_len, n_factors, **kwargs), forward([B, L, d]) -> [B, L, d]."""

    calls = 0

    def __init__(self, use_pos_emb, session_max_len, n_factors, amplitude=0.5, **kwargs):
        super().__init__()
        pos = torch.arange(session_max_len - 1, -1, -1, dtype=torch.float32)[:, None]
        freq = torch.exp(-torch.arange(n_factors, dtype=torch.float32)[None, :] / n_factors * 4.0)
        self.register_buffer("table", amplitude * torch.sin(pos * freq), persistent=False)
        self.use_pos_emb = use_pos_emb

    def forward(self, sessions):
        type(self).calls += 1
        return sessions + self.table[None] if self.use_pos_emb else sessions


class ScaledLossModule(hl.TransformerLossModule):
    """Subclass against the reference's constructor keywords (lightning.py:75-91) with its own loss hook."""

    def __init__(self, *args, loss_scale=1.0, **kwargs):
        super().__init__(*args, **kwargs)
        self.loss_scale = loss_scale
        self.seen = 0

    def training_loss(self, batch):
        self.seen += 1
        return super().training_loss(batch) * self.loss_scale


class TaggedBackbone(hnn.TransformerTorchBackbone):
    def __init__(self, *args, tag="none", **kwargs):
        super().__init__(*args, **kwargs)
        self.tag = tag


def test_seams_are_constructor_arguments_and_round_trip_through_configs():
    from rectools_amd.models import SASRecModel

    m = SASRecModel(n_factors=32, n_blocks=1, session_max_len=6, pos_encoding_type=SinusoidPositionalEncoding,
                    pos_encoding_kwargs={"amplitude": 0.25}, lightning_module_type=ScaledLossModule,
                    lightning_module_kwargs={"loss_scale": 2.0}, backbone_type=TaggedBackbone, backbone_kwargs={"tag": "mine"})
    cfg = m.get_config()
    assert cfg["pos_encoding_type"].endswith("test_plugin_seams.SinusoidPositionalEncoding")
    assert cfg["lightning_module_type"].endswith("test_plugin_seams.ScaledLossModule")
    assert cfg["backbone_type"].endswith("test_plugin_seams.TaggedBackbone") and cfg["backbone_kwargs"] == {"tag": "mine"}
    assert cfg["get_trainer_func"] is None
    m2 = SASRecModel.from_config(cfg)
    assert m2.pos_encoding_type is SinusoidPositionalEncoding and m2.lightning_module_type is ScaledLossModule
    assert m2.backbone_type is TaggedBackbone and m2.get_config() == cfg
    # defaults name the stock classes, as the reference's config does (transformers/base.py:215-222)
    d = SASRecModel().get_config()
    assert d["pos_encoding_type"] == "rectools_amd.nn.LearnableInversePositionalEncoding"
    assert d["lightning_module_type"] == "rectools_amd.lightning.TransformerLossModule"
    assert d["backbone_type"] == "rectools_amd.nn.TransformerTorchBackbone"


class EpochRecorder:
    """A user-defined callback, duck-typed against pytorch_lightning.Callback: records the hooks the loop reaches, logs a metric the
    way the reference's tutorial callbacks do (`pl_module.log_dict`), and asks for an early stop."""

    def __init__(self, stop_after=None):
        self.events, self.stop_after, self.val_batches = [], stop_after, 0

    def on_fit_start(self, trainer, pl_module):
        self.events.append("fit_start")

    def on_train_start(self, trainer, pl_module):
        self.events.append("train_start")

    def on_train_epoch_start(self, trainer, pl_module):
        self.events.append("epoch_start")

    def on_validation_batch_end(self, trainer, pl_module, outputs, batch, batch_idx, dataloader_idx=0):
        assert torch.isfinite(outputs["loss"]) and batch["x"].is_cuda and pl_module.item_embs.shape[0] == pl_module.torch_model.item_model.n_items
        self.val_batches += 1

    def on_validation_epoch_end(self, trainer, pl_module):
        pl_module.log_dict({"recall@10": 0.25 + 0.01 * len(self.events)}, on_step=False, on_epoch=True, prog_bar=True)

    def on_train_epoch_end(self, trainer, pl_module):
        self.events.append("epoch_end")
        done = self.events.count("epoch_end")
        assert "train_loss" in trainer.callback_metrics
        if self.stop_after is not None and done >= self.stop_after:
            trainer.should_stop = True

    def on_train_end(self, trainer, pl_module):
        self.events.append("train_end")


_RECORDER = EpochRecorder()


def duck_trainer(max_epochs=3, min_epochs=1, log_dir=None, callbacks=None):
    """What a `get_trainer_func` returns, as far as the engine reads it (no pytorch_lightning in this image)."""
    logger = types.SimpleNamespace(log_dir=log_dir) if log_dir else None
    return types.SimpleNamespace(max_epochs=max_epochs, min_epochs=min_epochs, callbacks=list(callbacks if callbacks is not None else [_RECORDER]),
                                 logger=logger, enable_progress_bar=False, callback_metrics={}, should_stop=False)


def test_get_trainer_func_is_kept_in_configs():
    from rectools_amd.models import BERT4RecModel, SASRecModel

    m = SASRecModel(get_trainer_func=duck_trainer, get_trainer_func_kwargs={"max_epochs": 2})
    cfg = m.get_config()
    assert cfg["get_trainer_func"].endswith("test_plugin_seams.duck_trainer") and cfg["get_trainer_func_kwargs"] == {"max_epochs": 2}
    assert SASRecModel.from_config(cfg).get_trainer_func is duck_trainer
    assert BERT4RecModel.from_config({"get_trainer_func": "test_plugin_seams.duck_trainer"}).get_trainer_func is duck_trainer
    SASRecModel(get_trainer_func=None, get_trainer_func_kwargs=None)


@pytest.mark.gpu
def test_get_trainer_func_drives_the_engine_loop(tmp_path):
    """transformers/base.py:367-380: a user-built Trainer.  Epoch counts, the logger's directory and user-defined callbacks are honoured
    by the engine's own loop; the rest is named in ONE warning per fit; `should_stop` ends training once min_epochs are done."""
    from rectools_amd.dataset import Dataset
    from rectools_amd.models import SASRecModel
    from rectools_amd.utils import leave_one_out_mask

    ds = Dataset.construct(_interactions())
    rec = EpochRecorder(stop_after=2)
    common = dict(n_factors=32, n_blocks=1, n_heads=2, session_max_len=4, lr=0.01, batch_size=4, epochs=7, seed=32, dropout_rate=0.0,
                  get_val_mask_func=leave_one_out_mask)
    model = SASRecModel(get_trainer_func=duck_trainer,
                        get_trainer_func_kwargs=dict(max_epochs=5, min_epochs=3, log_dir=str(tmp_path / "logs"), callbacks=[rec]), **common)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        model.fit(ds)
    msgs = [str(w.message) for w in caught if "get_trainer_func" in str(w.message)]
    assert len(msgs) == 1 and "max_epochs=5" in msgs[0] and "accelerator" in msgs[0]
    # asked to stop after 2 epochs, min_epochs = 3: three epochs ran (not the model's `epochs` = 7, not max_epochs = 5)
    assert model.epochs_done == 3 and len(model.history) == 3 and rec.events.count("epoch_end") == 3
    assert rec.events[:3] == ["fit_start", "train_start", "epoch_start"] and rec.events[-1] == "train_end" and rec.val_batches > 0
    assert all("recall@10" in h and "val_loss" in h for h in model.history)      # what the callback logged lands in the epoch records
    assert model.log_path is not None and model.log_path.startswith(str(tmp_path / "logs"))
    assert model.fit_trainer.callback_metrics["train_loss"].ndim == 0
    # fit_partial reads the factory again and takes its epoch counts from the call (transformers/base.py:527-529)
    rec.stop_after, n_before = None, rec.events.count("epoch_end")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model.fit_partial(ds, min_epochs=1, max_epochs=2)
    assert model.epochs_done == 5 and rec.events.count("epoch_end") == n_before + 2
    # a checkpoint keeps the factory's dotted path when it imports
    clone = SASRecModel.loads(model.dumps())
    assert clone.get_trainer_func is duck_trainer


def test_a_plugged_lightning_module_decides_whether_negatives_are_sampled():
    from rectools_amd.models import SASRecModel

    class PairwiseModule(hl.TransformerLossModule):
        @staticmethod
        def requires_negatives(loss):
            return True if loss == "my_pairwise" else hl.requires_negatives(loss)

    m = SASRecModel(loss="my_pairwise", n_negatives=7, lightning_module_type=PairwiseModule)
    assert m.data_preparator.n_negatives == 7 and m.data_preparator.negative_sampler is not None
    with pytest.raises(ValueError):
        SASRecModel(loss="my_pairwise")      # the stock module does not know it (lightning.py:115-124)


def test_stock_positional_encoding_forward_matches_the_reference_formula():
    pe = hnn.LearnableInversePositionalEncoding(True, 5, 8, use_scale_factor=True)
    torch.nn.init.normal_(pe.pos_emb.weight)
    x = torch.randn(2, 5, 8)
    want = x * 8 ** 0.5 + pe.pos_emb.weight[torch.arange(4, -1, -1)][None]      # net_blocks.py:388-399
    torch.testing.assert_close(pe(x), want)


def _interactions():
    return pd.DataFrame(
        [[10, 13, 1, "2021-11-30"], [10, 11, 1, "2021-11-29"], [10, 12, 1, "2021-11-29"], [30, 11, 1, "2021-11-27"],
         [30, 12, 2, "2021-11-26"], [30, 15, 1, "2021-11-25"], [40, 11, 1, "2021-11-25"], [40, 17, 1, "2021-11-26"],
         [50, 16, 1, "2021-11-25"], [10, 14, 1, "2021-11-28"], [10, 16, 1, "2021-11-27"], [20, 13, 9, "2021-11-28"]],
        columns=["user_id", "item_id", "weight", "datetime"])


@pytest.mark.gpu
def test_plugged_classes_are_the_ones_that_run():
    from rectools_amd.dataset import Dataset
    from rectools_amd.models import SASRecModel

    ds = Dataset.construct(_interactions())
    common = dict(n_factors=32, n_blocks=1, n_heads=2, session_max_len=4, lr=0.01, batch_size=4, epochs=2, loss="sampled_softmax",
                  n_negatives=3, seed=32, dropout_rate=0.0)
    SinusoidPositionalEncoding.calls = 0
    plugged = SASRecModel(pos_encoding_type=SinusoidPositionalEncoding, lightning_module_type=ScaledLossModule,
                          lightning_module_kwargs={"loss_scale": 2.0, "logits_t": 0.5}, backbone_type=TaggedBackbone,
                          backbone_kwargs={"tag": "mine"}, **common).fit(ds)
    lm = plugged.lightning_model
    assert isinstance(lm, ScaledLossModule) and lm.seen > 0 and lm.loss_scale == 2.0 and lm.logits_t == 0.5
    assert lm.lr == 0.01 and lm.loss == "sampled_softmax" and lm.data_preparator is plugged.data_preparator   # reference keyword set
    assert isinstance(lm.torch_model, TaggedBackbone) and lm.torch_model.tag == "mine"
    assert isinstance(lm.torch_model.pos_encoding_layer, SinusoidPositionalEncoding) and SinusoidPositionalEncoding.calls > 0
    assert not any("pos_emb" in n for n, _ in lm.torch_model.named_parameters())       # no learnable rows: the plugged class has none
    reco = plugged.recommend(users=np.array([10, 30, 40]), dataset=ds, k=3, filter_viewed=True)
    assert len(reco) > 0 and np.isfinite(reco["score"]).all()
    # the loss the plugged module reports is what the epoch log carries: twice the stock module's on the same weights
    stock = SASRecModel(pos_encoding_type=SinusoidPositionalEncoding, lightning_module_kwargs={"logits_t": 0.5}, **common)
    stock._build_model_from_dataset(ds)
    plugged2 = SASRecModel(pos_encoding_type=SinusoidPositionalEncoding, lightning_module_type=ScaledLossModule,
                           lightning_module_kwargs={"loss_scale": 2.0, "logits_t": 0.5}, **common)
    plugged2._build_model_from_dataset(ds)
    plugged2.torch_model.load_state_dict(stock.torch_model.state_dict())
    loop_a, loop_b = stock.training_loop(), plugged2.training_loop()
    stock.lightning_model.train(); plugged2.lightning_model.train()
    loop_a.begin_epoch(0); loop_b.begin_epoch(0)
    la, lb = float(loop_a.step()), float(loop_b.step())
    assert abs(lb - 2.0 * la) <= 1e-5 * abs(lb)
    # persistence keeps the plugged classes
    clone = SASRecModel.loads(plugged.dumps())
    assert isinstance(clone.lightning_model, ScaledLossModule) and isinstance(clone.torch_model, TaggedBackbone)
    pd.testing.assert_frame_equal(clone.recommend(users=np.array([10, 30, 40]), dataset=ds, k=3, filter_viewed=True), reco)


class ShiftedBackbone(hnn.TransformerTorchBackbone):
    """Overrides the reference-shaped encoder hook (torch_backbone.py:220-260)."""
    calls = 0

    def encode_sessions(self, batch, item_embs=None):
        ShiftedBackbone.calls += 1
        return super().encode_sessions(batch, item_embs)


@pytest.mark.gpu
def test_overridden_hooks_run_where_the_stack_would_otherwise_pack(monkeypatch):
    """packed training calls `training_loss_packed` / `encode_packed_train`, packed recommend `encode_last_packed` — a
    subclass overriding `training_loss` or `encode_sessions` was bypassed whenever the stack packs (stock positional encoding, head
    size 32 / 64).  With such a subclass plugged in the loop must keep the padded path, on which the override runs."""
    from rectools_amd.dataset import Dataset
    from rectools_amd.models import SASRecModel

    ds = Dataset.construct(_interactions())
    common = dict(n_factors=64, n_blocks=1, n_heads=2, session_max_len=4, lr=0.01, batch_size=4, epochs=1, loss="sampled_softmax",
                  n_negatives=3, seed=32, dropout_rate=0.0)      # head size 32, stock positional encoding: the stack packs
    stock = SASRecModel(**common)
    stock._build_model_from_dataset(ds)
    assert stock.training_loop().packed
    plugged = SASRecModel(lightning_module_type=ScaledLossModule, lightning_module_kwargs={"loss_scale": 3.0}, **common)
    plugged._build_model_from_dataset(ds)
    plugged.torch_model.load_state_dict(stock.torch_model.state_dict())
    loop_b = plugged.training_loop()
    assert not loop_b.packed
    monkeypatch.setenv("RT_PACKED_TRAIN", "0")       # the stock model on the padded window too: same batches, same negatives
    loop_a = stock.training_loop()
    monkeypatch.delenv("RT_PACKED_TRAIN")
    assert not loop_a.packed
    stock.lightning_model.train(); plugged.lightning_model.train()
    loop_a.begin_epoch(0); loop_b.begin_epoch(0)
    la, lb = float(loop_a.step()), float(loop_b.step())
    assert plugged.lightning_model.seen == 1 and abs(lb - 3.0 * la) <= 2e-5 * abs(lb)
    # a backbone that overrides encode_sessions: training AND recommend() go through it
    ShiftedBackbone.calls = 0
    m = SASRecModel(backbone_type=ShiftedBackbone, **common)
    m._build_model_from_dataset(ds)
    assert not m.training_loop().packed
    m.fit(ds)
    n_train = ShiftedBackbone.calls
    assert n_train > 0
    m.torch_model.eval()
    with torch.no_grad():
        assert not m.torch_model.can_encode_packed(64, 4)
    reco = m.recommend(users=np.array([10, 30, 40]), dataset=ds, k=3, filter_viewed=True)
    assert len(reco) > 0


@pytest.mark.gpu
def test_custom_positional_encoding_equals_the_fused_stock_path_when_it_restates_it():
    """A subclass that overrides forward() with the stock formula takes the modular path (embed -> forward -> dropout) and must
    give the encodings of the fused `rt_embed_fwd` path."""
    class Restated(hnn.LearnableInversePositionalEncoding):
        def forward(self, sessions):
            return super().forward(sessions)

    torch.manual_seed(0)
    V, L, d, H, B = 50, 6, 32, 2, 5
    def make(pe_cls):
        torch.manual_seed(1)
        im = hnn.SumOfEmbeddingsConstructor(V, [hnn.IdEmbeddingsItemNet(d, V, 0.0)])
        return hnn.TransformerTorchBackbone(H, 0.0, im, pe_cls(True, L, d, use_scale_factor=True), hnn.SASRecTransformerLayers(1, d, H, 0.0),
                                            hnn.DistanceSimilarityModule(), True, False).cuda().eval()
    a, b = make(hnn.LearnableInversePositionalEncoding), make(Restated)
    b.load_state_dict(a.state_dict())
    assert a._fused_pos() and not b._fused_pos()
    x = torch.randint(0, V, (B, L)).cuda()
    x[:, :2] = 0
    with torch.no_grad():
        torch.testing.assert_close(a.encode_sessions({"x": x}), b.encode_sessions({"x": x}), rtol=1e-5, atol=1e-6)


class HashedRowsItemNet(torch.nn.Module):
    """A plugged item-net block written against the reference's `ItemNetBase` (item_net.py:26-57): `from_dataset`, `get_all_embeddings`."""

    def __init__(self, n_items, n_factors, buckets=7):
        super().__init__()
        self.table = torch.nn.Parameter(torch.randn(buckets, n_factors) * 0.1)
        self.register_buffer("bucket_of", torch.arange(n_items) % buckets, persistent=False)

    @classmethod
    def from_dataset(cls, dataset, n_factors, dropout_rate, **kwargs):
        return cls(dataset.item_id_map.size, n_factors)

    def get_all_embeddings(self):
        return self.table[self.bucket_of]


@pytest.mark.gpu
def test_item_net_is_the_sum_of_whatever_blocks_it_is_given():
    """item_net.py:463-482 sums the block list as it is: two id blocks, a plugged block — the catalog matrix, the training gradients of
    every block and recommend() follow."""
    from rectools_amd.dataset import Dataset
    from rectools_amd.models import SASRecModel

    ds = Dataset.construct(_interactions())
    model = SASRecModel(n_factors=32, n_blocks=1, n_heads=2, session_max_len=4, lr=0.01, batch_size=4, epochs=2, seed=3, dropout_rate=0.0,
                        item_net_block_types=(hnn.IdEmbeddingsItemNet, hnn.IdEmbeddingsItemNet, HashedRowsItemNet))
    model._build_model_from_dataset(ds)
    im = model.torch_model.item_model
    a, b, c = (im.item_net_blocks[i] for i in range(3))
    want = a.ids_emb.weight + b.ids_emb.weight + c.get_all_embeddings()
    torch.testing.assert_close(im.get_all_embeddings(), want, rtol=0, atol=1e-6)
    before = [p.detach().clone() for p in (a.ids_emb.weight, b.ids_emb.weight, c.table)]
    model._run_epochs(0, 2)
    model.is_fitted = True
    for p0, p in zip(before, (a.ids_emb.weight, b.ids_emb.weight, c.table)):
        assert not torch.equal(p0, p.detach())                  # every block of the sum trains
    torch.testing.assert_close(a.ids_emb.weight - before[0], b.ids_emb.weight - before[1], rtol=1e-5, atol=1e-7)   # same gradient, same Adam path
    reco = model.recommend(users=np.array([10, 30, 40]), dataset=ds, k=3, filter_viewed=True)
    assert len(reco) > 0 and np.isfinite(reco["score"]).all()


# ---- similarity_module_type (transformers/base.py:415-421; the contract: similarity.py:26-64) -----------------------------------------
class TemperatureSimilarity(hnn.DistanceSimilarityModule):
    """A user's similarity module: the stock distances times a LEARNED scalar temperature, its own ranker scores scaled alike."""
    u2i_calls = 0

    def __init__(self, distance="dot", init_temperature=1.0, **kwargs):
        super().__init__(distance, **kwargs)
        self.log_t = torch.nn.Parameter(torch.tensor(float(np.log(init_temperature))))

    def forward(self, session_embs, item_embs, candidate_item_ids=None):
        return super().forward(session_embs, item_embs, candidate_item_ids) * torch.exp(self.log_t)

    def _recommend_u2i(self, user_embs, item_embs, user_ids, k, sorted_item_ids_to_recommend, ui_csr_for_filter):
        type(self).u2i_calls += 1
        u, i, s = super()._recommend_u2i(user_embs, item_embs, user_ids, k, sorted_item_ids_to_recommend, ui_csr_for_filter)
        return u, i, s * float(torch.exp(self.log_t))


class RenamedSimilarity(hnn.DistanceSimilarityModule):
    """Overrides nothing the reference calls: still served by the fused kernels."""
    note = "mine"


def test_only_a_module_that_restates_the_stock_methods_is_fused():
    assert hnn.similarity_is_stock(hnn.DistanceSimilarityModule("cosine"))
    assert hnn.similarity_is_stock(RenamedSimilarity("dot"))
    assert not hnn.similarity_is_stock(TemperatureSimilarity("dot", 2.0))

    class Tower(hnn.DistanceSimilarityModule):
        def item_tower_forward(self, item_embs):
            return item_embs * 2.0

    assert not hnn.similarity_is_stock(Tower())
    assert not hnn.similarity_is_stock(torch.nn.Identity())


def _seam_case(loss, dist, seed, N=6, logits_t=1.0):
    from test_transformer_gpu import _random_case

    return _random_case("sasrec", loss, dist, 12, 32, 2, 3, 40, N, seed, logits_t=logits_t)


@pytest.mark.parametrize("loss,dist", [("softmax", "dot"), ("sampled_softmax", "cosine"), ("BCE", "dot"), ("gBCE", "cosine")])
def test_losses_on_materialised_logits_equal_the_oracle(loss, dist):
    """The path of a plugged similarity module, host side (pure tensor ops: runs without a GPU): the stock module's `forward` and the
    reference's loss calculators restated in `TransformerLossModule._calc_*` against the oracle's logits and losses."""
    from oracle import transformer_oracle as T
    from test_transformer_gpu import build_hip_model

    cfg, batch = _seam_case(loss, dist, 3, logits_t=0.5)
    torch.manual_seed(0)
    lm = build_hip_model(cfg, device="cpu")
    hl.xavier_normal_init(lm.torch_model)
    params = {k: v.detach().clone() for k, v in lm.torch_model.state_dict().items()}
    with torch.no_grad():
        sess = T.encode_sessions(cfg, params, batch)
        want_logits = T.batch_logits(cfg, params, batch)
        want = T.training_loss(cfg, params, batch)
        table = T.item_table(params)
        got, logits = lm._loss_via_similarity(table, sess, batch["y"], batch["yw"], batch.get("negatives"))
    if loss == "sampled_softmax":        # the reference's calculator swaps columns 0 and 1 in place (lightning.py:209)
        logits = torch.cat([logits[..., 1:2], logits[..., 0:1], logits[..., 2:]], -1)
    torch.testing.assert_close(logits, want_logits, rtol=1e-5, atol=1e-6)
    assert abs(float(got) - float(want)) <= 1e-6 * abs(float(want)) + 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("loss,dist", [("softmax", "dot"), ("sampled_softmax", "cosine"), ("gBCE", "dot")])
def test_a_learned_temperature_changes_the_loss_exactly_as_the_oracle_with_that_temperature(loss, dist):
    """`logits * tau` is the oracle's `logits / logits_t` with logits_t / tau: loss, every parameter gradient and d loss / d log tau
    (autograd through the oracle with logits_t as a tensor) — the plugged module's `forward` is what the training step calls."""
    from oracle import transformer_oracle as T
    from test_transformer_gpu import _close, build_hip_model

    tau, t0 = 1.7, 0.5
    cfg, batch = _seam_case(loss, dist, 5, logits_t=t0)
    torch.manual_seed(100)
    lm = build_hip_model(cfg)
    hl.xavier_normal_init(lm.torch_model)
    lm.torch_model.similarity_module = TemperatureSimilarity(dist, tau).cuda()
    assert not lm.similarity_is_stock
    params = {k: v.detach().cpu().clone() for k, v in lm.torch_model.state_dict().items() if "similarity_module" not in k}
    log_t = torch.tensor(float(np.log(tau)), requires_grad=True)
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in params.items()}
    loss_ref = T.training_loss(dict(cfg, logits_t=t0 / torch.exp(log_t)), p, batch)
    loss_ref.backward()
    dbatch = {k: v.cuda() for k, v in batch.items()}
    lm.train()
    lm.zero_grad()
    got = lm.training_loss(dbatch)
    got.backward()
    assert abs(float(got) - float(loss_ref)) <= 5e-5 * abs(float(loss_ref)) + 5e-6, (float(got), float(loss_ref))
    g_t = float(lm.torch_model.similarity_module.log_t.grad)
    assert abs(g_t - float(log_t.grad)) <= 2e-3 * abs(float(log_t.grad)) + 1e-6, (g_t, float(log_t.grad))
    for n, q in lm.torch_model.named_parameters():
        if "similarity_module" in n:
            continue
        ref = p[n].grad if p[n].grad is not None else torch.zeros_like(p[n])
        if n == "item_model.item_net_blocks.0.ids_emb.weight":
            ref = ref.clone(); ref[0] = 0        # the PAD row never receives a gradient (item_net.py:260-264)
        _close(q.grad, ref, 1e-2, 2e-5 if ref.abs().max() > 1e-6 else 1.0, f"grad {n}")
    # validation outputs (lightning.py:336-359): the logits a callback receives are the plugged module's
    lm.eval()
    vb = {"x": dbatch["x"], "y": dbatch["y"][:, -1:], "yw": dbatch["yw"][:, -1:]}
    if "negatives" in dbatch:
        vb["negatives"] = dbatch["negatives"][:, -1:, :]
    with torch.no_grad():
        out = lm.validation_step(vb, 0)
        want = T.batch_logits(dict(cfg, logits_t=t0 / tau), params, batch)[:, -1, :]
    key = "logits" if loss == "softmax" else "pos_neg_logits"
    assert set(out) == {"loss", key}
    if loss == "sampled_softmax":      # handed on as the reference hands them on: columns 0 and 1 swapped in place by its loss calculator
        want = torch.cat([want[:, 1:2], want[:, 0:1], want[:, 2:]], -1)
    _close(out[key], want, 5e-4, 5e-5, key)


class RecallAtK:
    """A validation callback written against the reference's contract the way its tutorial's is (examples/tutorials/utils.py:34-133):
    takes `outputs["logits"]` when the step hands them over, else asks `pl_module.torch_model.similarity_module(session_embs,
    pl_module.item_embs)`; masks the items of `batch["x"]`; logs the mean hit rate through `pl_module.log_dict`."""

    def __init__(self, k):
        self.k, self.hits, self.from_outputs, self.from_module = k, [], 0, 0

    def on_validation_batch_end(self, trainer, pl_module, outputs, batch, batch_idx, dataloader_idx=0):
        if "logits" in outputs:
            logits = outputs["logits"]
            self.from_outputs += 1
        else:
            last = pl_module.torch_model.encode_sessions(batch, pl_module.item_embs)[:, -1, :]
            logits = pl_module.torch_model.similarity_module(last, pl_module.item_embs)
            self.from_module += 1
        logits = logits.reshape(batch["x"].shape[0], -1)      # (`logits.squeeze()` of lightning.py:351 drops a batch of one)
        assert logits.shape[1] == pl_module.torch_model.item_model.n_items
        seen = torch.zeros_like(logits, dtype=torch.bool).scatter_(1, batch["x"], True)
        seen[:, 0] = True
        top = logits.masked_fill(seen, float("-inf")).topk(self.k).indices
        self.hits.append((top == batch["y"]).any(1).float())

    def on_validation_epoch_end(self, trainer, pl_module):
        pl_module.log_dict({f"recall@{self.k}": float(torch.cat(self.hits).mean())}, on_step=False, on_epoch=True)
        self.hits.clear()


@pytest.mark.gpu
@pytest.mark.parametrize("loss", ["softmax", "sampled_softmax"])
def test_a_recall_callback_reads_the_validation_logits(loss):
    from rectools_amd.dataset import Dataset
    from rectools_amd.models import SASRecModel
    from rectools_amd.utils import leave_one_out_mask

    ds = Dataset.construct(_interactions())
    cb = RecallAtK(3)
    model = SASRecModel(n_factors=32, n_blocks=1, n_heads=2, session_max_len=4, lr=0.01, batch_size=4, seed=32, dropout_rate=0.0, loss=loss,
                        n_negatives=3 if loss != "softmax" else 1, get_val_mask_func=leave_one_out_mask, get_trainer_func=duck_trainer,
                        get_trainer_func_kwargs=dict(max_epochs=2, callbacks=[cb]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model.fit(ds)
    assert (cb.from_outputs > 0) == (loss == "softmax") and (cb.from_module > 0) == (loss != "softmax")
    assert all(0.0 <= h["recall@3"] <= 1.0 for h in model.history) and len(model.history) == 2
    # the logits the callback saw rank like recommend() does: recall@n_items is 1 for every user whose target the model knows
    lm = model.lightning_model
    lm.eval()
    with torch.no_grad():
        table = lm.torch_model.item_model.get_all_embeddings()
        x = torch.tensor([[0, 0, 13, 11]], device=table.device)
        full = lm.torch_model.similarity_module(lm.torch_model.encode_sessions({"x": x}, table)[:, -1, :], table)
        want = lm.torch_model.encode_sessions({"x": x}, table)[:, -1, :] @ table.T
    torch.testing.assert_close(full, want, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_a_plugged_similarity_module_trains_and_recommends_through_its_own_methods():
    """End to end: the temperature is a parameter of the flat Adam buffer and moves; the step's loss is the stock model's at logits_t =
    1 / tau on the same weights; recommend() returns the module's own triplet (the stock items, scores times tau)."""
    from rectools_amd.dataset import Dataset
    from rectools_amd.models import SASRecModel

    ds = Dataset.construct(_interactions())
    common = dict(n_factors=64, n_blocks=1, n_heads=2, session_max_len=4, lr=0.01, batch_size=4, epochs=3, loss="sampled_softmax",
                  n_negatives=3, seed=32, dropout_rate=0.0)
    plugged = SASRecModel(similarity_module_type=TemperatureSimilarity, similarity_module_kwargs={"init_temperature": 2.0}, **common)
    plugged._build_model_from_dataset(ds)
    assert not plugged.training_loop().packed            # reference-shaped batches for a module written against the reference
    stock = SASRecModel(lightning_module_kwargs={"logits_t": 0.5}, **common)
    stock._build_model_from_dataset(ds)
    sd = {k: v for k, v in plugged.torch_model.state_dict().items() if "similarity_module" not in k}
    stock.torch_model.load_state_dict(sd)
    import os
    os.environ["RT_PACKED_TRAIN"] = "0"
    try:
        loop_a = stock.training_loop()
    finally:
        del os.environ["RT_PACKED_TRAIN"]
    loop_b = plugged.training_loop()
    stock.lightning_model.train(); plugged.lightning_model.train()
    loop_a.begin_epoch(0); loop_b.begin_epoch(0)
    la, lb = float(loop_a.step()), float(loop_b.step())
    assert abs(la - lb) <= 2e-5 * abs(la), (la, lb)
    sim = plugged.torch_model.similarity_module
    assert float(sim.log_t) != float(np.log(2.0))         # Adam moved it
    plugged._run_epochs(0, 3); plugged.is_fitted = True
    tau = float(torch.exp(sim.log_t))
    TemperatureSimilarity.u2i_calls = 0
    users = np.array([10, 30, 40])
    reco = plugged.recommend(users=users, dataset=ds, k=3, filter_viewed=True)
    assert TemperatureSimilarity.u2i_calls == 1
    twin = SASRecModel(**common)
    twin._build_model_from_dataset(ds)
    twin.torch_model.load_state_dict({k: v for k, v in plugged.torch_model.state_dict().items() if "similarity_module" not in k})
    twin.is_fitted = True
    want = twin.recommend(users=users, dataset=ds, k=3, filter_viewed=True)
    assert list(reco.columns) == list(want.columns) and (reco["user_id"].values == want["user_id"].values).all()
    assert (reco["item_id"].values == want["item_id"].values).all() and (reco["rank"].values == want["rank"].values).all()
    np.testing.assert_allclose(reco["score"].values, want["score"].values * tau, rtol=1e-5)
    # it survives persistence with its class and its learned value
    clone = SASRecModel.loads(plugged.dumps())
    assert isinstance(clone.torch_model.similarity_module, TemperatureSimilarity)
    pd.testing.assert_frame_equal(clone.recommend(users=users, dataset=ds, k=3, filter_viewed=True), reco)


# ---- round-5 advisor findings, host side ---------------------------------------------------------------------------------------------
def test_the_trainer_logger_directory_is_the_one_versions_are_numbered_in():
    """Lightning's CSVLogger: log_dir = save_dir/name/version_N.  `_log_epoch` numbers its own version directory, so the plan takes
    save_dir/name (or log_dir's parent), and the run's directory never becomes a hyper-parameter of the model."""
    from rectools_amd.models import SASRecModel

    def trainer_with(logger):
        return lambda: types.SimpleNamespace(max_epochs=1, min_epochs=1, callbacks=[], logger=logger, enable_progress_bar=False)

    cases = [
        (types.SimpleNamespace(save_dir="/tmp/x", name="lightning_logs", log_dir="/tmp/x/lightning_logs/version_3"), "/tmp/x/lightning_logs"),
        (types.SimpleNamespace(save_dir="/tmp/x", name="", log_dir="/tmp/x/version_0"), "/tmp/x"),
        (types.SimpleNamespace(log_dir="/tmp/y/z/version_12"), "/tmp/y/z"),
        (types.SimpleNamespace(log_dir="/tmp/plain"), "/tmp/plain"),
    ]
    for logger, want in cases:
        m = SASRecModel(get_trainer_func=trainer_with(logger))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            plan = m._trainer_plan()
        assert plan["log_dir"] == want, (plan["log_dir"], want)
        assert "csv_log_dir" not in m.get_config() or m.get_config()["csv_log_dir"] is None


def test_a_checkpointed_dataset_schema_reads_like_the_reference_object():
    from rectools_amd.models import _schema_view

    schema = {"n_interactions": 5, "items": {"n_hot": 7, "features": {"kind": "sparse", "cat_feature_indices": [0, 2], "names": [["f", 1]]}},
              "users": {"n_hot": 3, "features": None}}
    v = _schema_view(schema)
    assert v.items.n_hot == 7 and v.items.features.cat_feature_indices == [0, 2] and v.users.features is None
    assert v["items"]["features"]["kind"] == "sparse" and "users" in v and len(v) == 3 and v.get("nope") is None
    assert {k: v[k] for k in v.keys()}["n_interactions"] == 5                          # still readable as the mapping it was saved as
    with pytest.raises(AttributeError):
        v.items.nope


def test_a_tables_home_in_the_flat_gradient_buffer_is_handed_out_once_per_step():
    """Two loss nodes on one leaf table in one backward pass must not both write the table's segment of the flat gradient buffer
    (autograd would add two aliases of the same memory: 2 dB instead of dA + dB)."""
    from rectools_amd import ops

    table = torch.nn.Parameter(torch.zeros(8, 4))
    home = torch.zeros(8, 4)
    ops._TABLE_GRAD_HOME[table.data_ptr()] = lambda: home[:]
    try:
        ops.clear_step_expectations()
        a, b = ops._new_table_grad(table), ops._new_table_grad(table)
        assert a.data_ptr() == home.data_ptr() and b.data_ptr() != home.data_ptr()
        ops.clear_step_expectations()                                  # the next step's first node gets the home again
        assert ops._new_table_grad(table).data_ptr() == home.data_ptr()
    finally:
        ops._TABLE_GRAD_HOME.pop(table.data_ptr(), None)
        ops.clear_step_expectations()
