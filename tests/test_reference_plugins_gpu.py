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


def _run(model_cls, ds, users, plugged, context=None, **kw):
    from rectools.models.nn.item_net import IdEmbeddingsItemNet

    extra = {}
    if plugged:
        extra = dict(transformer_layers_type=plugged, similarity_module_type="rectools_amd.reference_plugins.HipDistanceSimilarityModule")
    _seed()
    model = model_cls.from_config(dict(dropout_rate=0.0, epochs=2, batch_size=8, session_max_len=8, lr=3e-3, deterministic=False,
                                       item_net_block_types=(IdEmbeddingsItemNet,), get_trainer_func=gpu_trainer,
                                       recommend_torch_device="cuda", **extra, **kw))
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
