"""Lightning-layout checkpoints (rectools_amd/checkpoint.py).  Fixtures tests/golden/ckpt_*.ckpt hold
models the UNMODIFIED reference trained (through the shimmed Trainer), stored in the layout of `Trainer.save_checkpoint`,
plus the reference's own recommend() / recommend_to_items() frames for them.

CPU tests: config translation, dataset schema, torch.optim.Adam state exchange.  GPU tests: a reference checkpoint loads
into the HIP engine and reproduces the reference's recommendations; write -> read round trip; training continues."""
import importlib
import os

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import GOLDEN_DIR

CKPTS = ["sasrec_catfeat", "bert4rec_ids", "hstu_time_pos", "esasrec_ligr"]


def _load(name):
    return torch.load(os.path.join(GOLDEN_DIR, f"ckpt_{name}.ckpt"), map_location="cpu", weights_only=False)


def _frames():
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


def _dataset(name):
    from rectools_amd.dataset import Dataset

    interactions, features = _frames()
    if name == "sasrec_catfeat":
        return Dataset.construct(interactions, item_features_df=features, cat_item_features=["f1", "f2"])
    return Dataset.construct(interactions)


def _model_class(name):
    from rectools_amd.models import BERT4RecModel, HSTUModel, SASRecModel

    return {"sasrec_catfeat": SASRecModel, "bert4rec_ids": BERT4RecModel, "hstu_time_pos": HSTUModel, "esasrec_ligr": SASRecModel}[name]


def _context(model):
    """HSTU with relative time attention ranks "as of" a request time (hstu.py:719-729): one context row per user."""
    from rectools_amd.utils import get_context

    if not model.require_recommend_context:
        return None
    return get_context(pd.DataFrame({"user_id": [10, 30, 40], "datetime": ["2021-12-12", "2021-12-13", "2021-12-12"]}))


# ---- CPU -----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", CKPTS)
def test_reference_config_translates_to_importable_classes(name):
    from rectools_amd import checkpoint as ckpt

    ref_cfg = _load(name)["hyper_parameters"]["model_config"]
    cfg = ckpt.translate_config(ref_cfg)
    for key, value in cfg.items():
        paths = value if isinstance(value, list) else [value]
        for path in paths:
            if isinstance(path, str) and path.startswith("rectools"):
                assert path.startswith("rectools_amd."), f"{key}: {path} was not translated"
                module, attr = path.rsplit(".", 1)
                assert hasattr(importlib.import_module(module), attr), path
    back = ckpt.translate_config(cfg, to_reference=True)
    assert back == ref_cfg
    model = _model_class(name).from_config({k: v for k, v in cfg.items() if k != "cls"})
    assert model.n_factors == 32 and model.loss == ref_cfg["loss"] and model.session_max_len == 4
    assert [t.__name__ for t in model.item_net_block_types] == [p.rsplit(".", 1)[1] for p in ref_cfg["item_net_block_types"]]
    assert model.transformer_layers_type.__name__ == ref_cfg["transformer_layers_type"].rsplit(".", 1)[1]


@pytest.mark.parametrize("name", CKPTS)
def test_dataset_schema_and_item_ids_match_the_reference(name):
    """The host mirror (Dataset.construct -> process_dataset_train -> get_schema) yields what the reference recorded in
    its checkpoint: same item id order, same schema dict (dataset.py:139-174)."""
    from rectools_amd import checkpoint as ckpt

    hyper = _load(name)["hyper_parameters"]
    model = _model_class(name)(session_max_len=4, batch_size=4)
    model.data_preparator.process_dataset_train(_dataset(name))
    assert list(model.data_preparator.item_id_map.external_ids) == hyper["item_external_ids"]
    assert model.data_preparator.train_dataset.get_schema() == hyper["dataset_schema"]
    blocks = ckpt.item_net_schema(hyper["dataset_schema"])
    if name == "sasrec_catfeat":
        assert blocks == [{"kind": "ids"}, {"kind": "cat", "nnz": 8, "n_cat_feature_values": 4}]
    else:
        assert blocks == [{"kind": "ids"}]


def test_adam_state_exchange_with_torch_optim():
    """FlatAdam <-> torch.optim.Adam.state_dict(): moments of a real torch Adam (3 steps on CPU) are imported by name
    even when the two modules register their parameters in different orders, and exported back unchanged."""
    from rectools_amd import checkpoint as ckpt
    from rectools_amd.lightning import FlatAdam

    torch.manual_seed(0)

    class A(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.first = torch.nn.Linear(5, 3)
            self.second = torch.nn.Embedding(7, 5)

    class B(torch.nn.Module):     # same names, other registration order
        def __init__(self):
            super().__init__()
            self.second = torch.nn.Embedding(7, 5)
            self.first = torch.nn.Linear(5, 3)

    a, b = A(), B()
    b.load_state_dict(a.state_dict())
    topt = torch.optim.Adam(a.parameters(), lr=0.01, betas=(0.9, 0.98))
    for step in range(3):
        topt.zero_grad()
        (a.first(a.second(torch.tensor([1, 2, 6]))).square().sum() * (step + 1)).backward()
        topt.step()
    sd = topt.state_dict()
    flat = FlatAdam(b, lr=0.5)
    names_a = [n for n, _ in a.named_parameters()]
    names_b = [n for n, _ in b.named_parameters()]
    assert names_a != names_b
    ckpt.load_adam_state_dict(flat, sd, names_a, names_b)
    assert flat.step_count == 3 and flat.lr == 0.01 and tuple(flat.betas) == (0.9, 0.98)
    out = ckpt.adam_state_dict(flat)
    for i, n in enumerate(names_b):
        j = names_a.index(n)
        assert torch.equal(out["state"][i]["exp_avg"], sd["state"][j]["exp_avg"])
        assert torch.equal(out["state"][i]["exp_avg_sq"], sd["state"][j]["exp_avg_sq"])
        assert float(out["state"][i]["step"]) == 3.0
    fresh = torch.optim.Adam(b.parameters(), lr=0.01, betas=(0.9, 0.98))
    fresh.load_state_dict(out)          # torch accepts what we write
    with pytest.raises(ValueError):
        ckpt.load_adam_state_dict(flat, {"state": {}, "param_groups": [{"params": [0]}]})


# ---- GPU -----------------------------------------------------------------------------------------------------
def _assert_frame(got: pd.DataFrame, exp: dict, score_rtol=2e-4):
    assert list(got.columns) == list(exp.keys())
    for col, values in exp.items():
        if col == "score":
            np.testing.assert_allclose(got[col].values, np.asarray(values, np.float32), rtol=score_rtol, atol=2e-5)
        else:
            assert got[col].tolist() == values, col


@pytest.mark.gpu
@pytest.mark.parametrize("name", CKPTS)
def test_reference_checkpoint_loads_and_reproduces_reference_recommendations(name, tmp_path):
    from rectools_amd import checkpoint as ckpt

    ref = _load(name)
    path = os.path.join(GOLDEN_DIR, f"ckpt_{name}.ckpt")
    klass = _model_class(name)
    model = klass.load_from_checkpoint(path)
    assert model.is_fitted and model.epochs_done == ref["epoch"] and model.optimizer.step_count == ref["global_step"]
    sd = model.torch_model.state_dict()
    assert sorted(ckpt.STATE_PREFIX + k for k in sd) == sorted(ref["state_dict"])
    for k, v in ref["state_dict"].items():
        assert torch.equal(sd[k[len(ckpt.STATE_PREFIX):]].cpu(), v), k
    ds = _dataset(name)
    users, exp, ctx = [10, 30, 40], ref["expected"], _context(model)
    _assert_frame(model.recommend(users=users, dataset=ds, k=3, filter_viewed=True, context=ctx), exp["filter"])
    _assert_frame(model.recommend(users=users, dataset=ds, k=4, filter_viewed=False, context=ctx), exp["nofilter"])
    _assert_frame(model.recommend(users=users, dataset=ds, k=2, filter_viewed=False, items_to_recommend=[11, 13, 17], context=ctx),
                  exp["whitelist"])
    _assert_frame(model.recommend_to_items(target_items=[11, 12], dataset=ds, k=2), exp["i2i"])
    # Adam moments arrived (matched by name) and go back out bit-identically, in Lightning's layout
    out = ckpt.to_checkpoint(model)
    assert set(out) >= {"epoch", "global_step", "pytorch-lightning_version", "state_dict", "loops", "callbacks", "optimizer_states",
                        "lr_schedulers", "hyper_parameters"}
    assert out["hyper_parameters"]["model_config"]["transformer_layers_type"] == ref["hyper_parameters"]["model_config"]["transformer_layers_type"]
    assert out["hyper_parameters"]["dataset_schema"] == ref["hyper_parameters"]["dataset_schema"]
    assert isinstance(out["hyper_parameters"]["item_external_ids"], np.ndarray)         # what the reference's IdMap(...) takes
    assert list(out["hyper_parameters"]["item_external_ids"]) == list(ref["hyper_parameters"]["item_external_ids"])
    names_ref = [k for k in ref["state_dict"] if (k[len(ckpt.STATE_PREFIX):] in dict(model.torch_model.named_parameters()))]
    names_mine = [ckpt.STATE_PREFIX + n for n, _ in model.torch_model.named_parameters()]
    ref_state, my_state = ref["optimizer_states"][0]["state"], out["optimizer_states"][0]["state"]
    for i, n in enumerate(names_mine):
        j = names_ref.index(n)
        assert torch.equal(my_state[i]["exp_avg"], ref_state[j]["exp_avg"]), n
        assert torch.equal(my_state[i]["exp_avg_sq"], ref_state[j]["exp_avg_sq"]), n
    # write -> read round trip (file and pickle), then training continues from the restored state
    p2 = str(tmp_path / "again.ckpt")
    model.save_to_checkpoint(p2)
    again = klass.load_from_checkpoint(p2)
    _assert_frame(again.recommend(users=users, dataset=ds, k=3, filter_viewed=True, context=ctx), exp["filter"])
    clone = klass.loads(model.dumps())
    _assert_frame(clone.recommend(users=users, dataset=ds, k=3, filter_viewed=True, context=ctx), exp["filter"])
    before = {k: v.clone() for k, v in again.torch_model.state_dict().items()}
    again.fit_partial(ds, max_epochs=1)
    assert again.epochs_done == ref["epoch"] + 1 and again.optimizer.step_count > ref["global_step"]
    assert np.isfinite(again.history[-1]["train_loss"])
    assert any(not torch.equal(before[k], v) for k, v in again.torch_model.state_dict().items() if v.is_floating_point())
    # weights only, into another fitted model
    other = klass.load_from_checkpoint(p2, model_params_update={"lr": 0.5})
    assert other.lr == 0.5
    again.load_weights_from_checkpoint(path)
    _assert_frame(again.recommend(users=users, dataset=ds, k=3, filter_viewed=True, context=ctx), exp["filter"])


@pytest.mark.gpu
def test_unfitted_model_pickles_as_config():
    from rectools_amd.models import NotFittedError, SASRecModel

    m = SASRecModel.loads(SASRecModel(n_factors=32, loss="BCE").dumps())
    assert m.n_factors == 32 and m.loss == "BCE" and not m.is_fitted
    with pytest.raises(NotFittedError):
        m.recommend([1], None, 3, False)
