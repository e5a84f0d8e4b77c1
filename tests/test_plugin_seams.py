"""The reference's plug-in seams (transformers/base.py:215-222,278-286,368-380,407,430,449): `pos_encoding_type`, `backbone_type`,
`lightning_module_type` are instantiated from the class handed over (as `transformer_layers_type` / `negative_sampler_type` are),
`get_trainer_func` is refused unless None; all of it round-trips through `get_config` / `from_config`."""
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


def test_get_trainer_func_is_refused_not_swallowed():
    from rectools_amd.models import BERT4RecModel, SASRecModel

    with pytest.raises(NotImplementedError, match="get_trainer_func"):
        SASRecModel(get_trainer_func=lambda **kw: None)
    with pytest.raises(NotImplementedError, match="get_trainer_func"):
        BERT4RecModel.from_config({"get_trainer_func": "tests.test_plugin_seams.ScaledLossModule"})
    SASRecModel(get_trainer_func=None, get_trainer_func_kwargs=None)


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
