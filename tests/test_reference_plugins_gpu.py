"""The drop-in seam from the REFERENCE's side (`transformers/base.py:58-186`, `similarity.py:117-140`): the
unmodified `rectools.models.SASRecModel` / `HSTUModel` with the HIP plug-in classes selected by dotted path
(`rectools_amd.reference_plugins`) — fitted by the reference's own fit() (a Trainer on the GPU: `oracle/ref_shims` stands in for
pytorch_lightning, which this image lacks) and asked for recommendations through the reference's own recommend() — against the same
model run with the reference's stock classes, same seeds.  Dropout is off (the two implementations draw different masks); what remains
is fp32 summation order, so the trained weights agree to ~1e-4 and the frames hold the same items with the same scores.

Runs where the reference tree is: `/root/reference` in the build container, the staged copy `oracle/_ref` on the GPU box."""
import random

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import ref_shims

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_shims.reference_available(), reason="reference tree not present")]


@pytest.fixture(scope="module", autouse=True)
def _shims():
    ref_shims.install()


def _frames(seed=0, n_users=40, n_items=30):
    rng = np.random.default_rng(seed)
    rows = []
    for u in range(n_users):
        n = int(rng.integers(3, 14))
        ts = pd.Timestamp("2022-01-01") + pd.to_timedelta(np.cumsum(rng.integers(1, 200, n)), unit="h")
        rows.append(pd.DataFrame({"user_id": u + 1, "item_id": rng.integers(0, n_items, n) + 500, "weight": 1.0, "datetime": ts}))
    return pd.concat(rows, ignore_index=True)


def gpu_trainer(max_epochs=2):
    from pytorch_lightning import Trainer      # (the shim; a real Trainer takes the same arguments)

    return Trainer(max_epochs=max_epochs, min_epochs=max_epochs, accelerator="gpu", devices=1, enable_checkpointing=False, logger=False,
                   enable_progress_bar=False, enable_model_summary=False)


def _seed(seed=32):
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)


_P = "rectools_amd.reference_plugins."


def _run(model_cls, ds, users, plugged, context=None, everything=False, **kw):
    """plugged: dotted path of a HIP layer stack (+ the HIP ranker); everything: the lightning module (fused losses, fused Adam) and the
    id item net as well."""
    from rectools.models.nn.item_net import IdEmbeddingsItemNet

    extra = {"item_net_block_types": (IdEmbeddingsItemNet,)}
    if plugged:
        extra.update(transformer_layers_type=plugged, similarity_module_type=_P + "HipDistanceSimilarityModule")
    if everything:
        extra.update(lightning_module_type=_P + "HipTransformerLightningModule", item_net_block_types=(_P + "HipIdEmbeddingsItemNet",))
    extra.update(kw)
    _seed()
    model = model_cls.from_config(dict(dropout_rate=0.0, epochs=2, batch_size=8, session_max_len=8, lr=3e-3, deterministic=False,
                                       get_trainer_func=gpu_trainer, recommend_torch_device="cuda", **extra))
    model.fit(ds)
    reco = model.recommend(users, ds, k=5, filter_viewed=True, **({"context": context} if context is not None else {}))
    state = {k: v.detach().cpu() for k, v in model.lightning_model.torch_model.state_dict().items()}
    return model, reco, state


def _compare(a, b):
    (ma, ra, sa), (mb, rb, sb) = a, b
    assert list(sa) == list(sb)                                     # same parameter names: checkpoints interchange
    for k in sa:
        assert sa[k].shape == sb[k].shape, k
        if sa[k].is_floating_point():
            x, y = sb[k], sa[k]
            if k.endswith("in_proj_bias"):
                # the key bias shifts every logit of a query alike: its true gradient is ZERO, what each implementation computes is
                # rounding noise — which Adam normalises to steps of the order of lr with a random sign.  Compare the q and v thirds
                n = x.numel() // 3
                x, y = torch.cat([x[:n], x[2 * n:]]), torch.cat([y[:n], y[2 * n:]])
            torch.testing.assert_close(x, y, rtol=2e-3, atol=2e-4, msg=lambda m, k=k: f"{k} after training: {m}")
    assert len(ra) == len(rb) and (ra["user_id"].values == rb["user_id"].values).all()
    same = (ra["item_id"].values == rb["item_id"].values)
    assert same.mean() >= 0.97, f"only {same.mean():.3f} of the recommended items agree"      # (near-ties may swap neighbours)
    np.testing.assert_allclose(rb["score"].values[same], ra["score"].values[same], rtol=2e-3, atol=2e-3)
    # per user: the same item SETS except where a near-tie sits on the k-th place
    sets_a = ra.groupby("user_id")["item_id"].apply(frozenset)
    sets_b = rb.groupby("user_id")["item_id"].apply(frozenset)
    assert (sets_a == sets_b).mean() >= 0.9


@pytest.mark.parametrize("loss,keypad", [("softmax", False), ("sampled_softmax", True)])
def test_reference_sasrec_with_hip_plugins_trains_and_recommends_like_the_reference(loss, keypad):
    from rectools.dataset import Dataset
    from rectools.models import SASRecModel
    from rectools.models.nn.transformers.net_blocks import TransformerLayersBase

    from rectools_amd import nn as hnn
    from rectools_amd.reference_plugins import HipDistanceSimilarityModule, HipSASRecTransformerLayers

    assert issubclass(HipSASRecTransformerLayers, TransformerLayersBase) and issubclass(HipSASRecTransformerLayers, hnn.SASRecTransformerLayers)
    df = _frames()
    ds = Dataset.construct(df)
    users = np.unique(df["user_id"])
    kw = dict(n_factors=64, n_heads=2, n_blocks=2, loss=loss, n_negatives=5, use_key_padding_mask=keypad)
    ref = _run(SASRecModel, ds, users, None, **kw)
    hip = _run(SASRecModel, ds, users, "rectools_amd.reference_plugins.HipSASRecTransformerLayers", **kw)
    tm = hip[0].lightning_model.torch_model
    assert isinstance(tm.transformer_layers, HipSASRecTransformerLayers) and isinstance(tm.similarity_module, HipDistanceSimilarityModule)
    assert hip[0].get_config(simple_types=True)["transformer_layers_type"].endswith("reference_plugins.HipSASRecTransformerLayers")
    _compare(ref, hip)


def test_reference_hstu_with_hip_plugins_trains_and_recommends_like_the_reference():
    from rectools.dataset import Dataset
    from rectools.dataset.context import get_context
    from rectools.models import HSTUModel

    df = _frames(seed=2)
    ds = Dataset.construct(df)
    users = np.unique(df["user_id"])
    ctx = get_context(pd.DataFrame({"user_id": users, "datetime": pd.Timestamp("2023-03-01")}))
    kw = dict(n_factors=64, n_heads=2, n_blocks=2, loss="sampled_softmax", n_negatives=5)
    ref = _run(HSTUModel, ds, users, None, context=ctx, **kw)
    hip = _run(HSTUModel, ds, users, "rectools_amd.reference_plugins.HipSTULayers", context=ctx, **kw)
    _compare(ref, hip)


def test_reference_bert4rec_with_the_hip_pre_ln_stack():
    """bert4rec.py:204-452 with `HipPreLNTransformerLayers`: the reference's own collate masks the items (numpy draws: the same in both
    runs), the HIP stack encodes under the key-padding mask."""
    from rectools.dataset import Dataset
    from rectools.models import BERT4RecModel

    from rectools_amd.reference_plugins import HipPreLNTransformerLayers

    df = _frames(seed=4)
    ds = Dataset.construct(df)
    users = np.unique(df["user_id"])
    kw = dict(n_factors=64, n_heads=2, n_blocks=2, loss="softmax", mask_prob=0.3)
    ref = _run(BERT4RecModel, ds, users, None, **kw)
    hip = _run(BERT4RecModel, ds, users, _P + "HipPreLNTransformerLayers", **kw)
    assert isinstance(hip[0].lightning_model.torch_model.transformer_layers, HipPreLNTransformerLayers)
    _compare(ref, hip)


def test_reference_esasrec_with_the_hip_ligr_stack():
    """SASRecModel + `LiGRLayers` (ligr.py:109-191, the eSASRec configuration of BASELINE config 5) with `HipLiGRLayers`."""
    from rectools.dataset import Dataset
    from rectools.models import SASRecModel

    from rectools_amd.reference_plugins import HipLiGRLayers

    df = _frames(seed=5)
    ds = Dataset.construct(df)
    users = np.unique(df["user_id"])
    lkw = dict(ff_factors_multiplier=4, ff_activation="swiglu", bias_in_ff=False)
    kw = dict(n_factors=64, n_heads=2, n_blocks=2, loss="sampled_softmax", n_negatives=5, transformer_layers_kwargs=lkw)
    ref = _run(SASRecModel, ds, users, None, transformer_layers_type="rectools.models.nn.transformers.ligr.LiGRLayers", **kw)
    hip = _run(SASRecModel, ds, users, _P + "HipLiGRLayers", **kw)
    assert isinstance(hip[0].lightning_model.torch_model.transformer_layers, HipLiGRLayers)
    _compare(ref, hip)


@pytest.mark.parametrize("loss,dist", [("softmax", "dot"), ("sampled_softmax", "cosine"), ("gBCE", "dot")])
def test_every_plug_in_at_once(loss, dist):
    """Layer stack, ranker, lightning module (fused loss kernels + the fused Adam behind a torch.optim.Optimizer), id item net and the
    device-side sampler selected together on the reference's SASRecModel — against the reference's stock classes fed the SAME negatives
    (the sampler is plugged into both runs: the reference's own draws come from torch's global generator and cannot be replayed)."""
    from rectools.dataset import Dataset
    from rectools.models import SASRecModel

    from rectools_amd import reference_plugins as rp

    df = _frames(seed=6)
    ds = Dataset.construct(df)
    users = np.unique(df["user_id"])
    kw = dict(n_factors=64, n_heads=2, n_blocks=2, loss=loss, n_negatives=6, negative_sampler_type=_P + "HipCatalogUniformSampler",
              negative_sampler_kwargs={"seed": 11}, similarity_module_kwargs={"distance": dist}, lightning_module_kwargs={"logits_t": 0.5})
    ref = _run(SASRecModel, ds, users, None, **kw)
    hip = _run(SASRecModel, ds, users, _P + "HipSASRecTransformerLayers", everything=True, **kw)
    lm = hip[0].lightning_model
    assert isinstance(lm, rp.HipTransformerLightningModule) and isinstance(lm.optimizer, rp.FlatAdamOptimizer) and lm._fused()
    assert isinstance(lm.torch_model.item_model.item_net_blocks[0], rp.HipIdEmbeddingsItemNet)
    if loss != "softmax":
        assert isinstance(hip[0].data_preparator.negative_sampler, rp.HipCatalogUniformSampler) and hip[0].data_preparator.negative_sampler.calls > 0
    # the optimizer state the façade hands Lightning is torch.optim.Adam's: it loads into the stock run's optimizer and back
    sd = lm.optimizer.state_dict()
    ref[0].lightning_model.optimizer.load_state_dict(sd)
    lm.optimizer.load_state_dict(ref[0].lightning_model.optimizer.state_dict())
    _compare(ref, hip)
