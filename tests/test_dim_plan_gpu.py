"""Sizes the kernels do not tile, on the GPU: the padded model (`nn.DimPlan`: zero columns behind every real one, column-aware LayerNorm
`rt_layernorm_*_cols`, softmax scale of the real head size `rt_mha_*_scaled`) against the ORACLE evaluated at the REAL sizes — eval
encodings, training loss, every parameter gradient (in the real shapes), zero gradients and zero Adam updates for every padded entry.
Cases: the reference's published HSTU configuration n_factors = 50 with 1 and 2 heads (`transformers_HSTU_tutorial.ipynb:470-484`; hstu.py:
558-607 accepts any n_factors % n_heads == 0), odd head sizes of the SASRec / BERT4Rec / LiGR stacks, STU stacks whose u / v and q / k
head sizes differ (hstu.py:186-221), a `num_buckets` other than 128 (hstu.py:63-78).  Same tolerances as tests/test_transformer_gpu.py.
"""
import numpy as np
import pandas as pd
import pytest
import torch

from oracle import transformer_oracle as T
from test_dim_plan import ODD
from test_transformer_gpu import _close, build_hip_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,case", ODD, ids=[c[0] for c in ODD])
def test_padded_model_equals_the_oracle_at_the_real_sizes(name, case):
    from rectools_amd import lightning as hl
    from rectools_amd.nn import pad_tensor, unpad_tensor

    cfg, batch = case
    torch.manual_seed(100)
    real = build_hip_model(cfg, device="cpu", real_size=True)      # initialised at the real shapes, as models._build_model_from_dataset does
    hl.xavier_normal_init(real.torch_model)
    with torch.no_grad():
        for p in real.torch_model.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    params = {k: v.detach().clone() for k, v in real.torch_model.state_dict().items()}
    lm = build_hip_model(cfg, params=params)
    assert lm.torch_model.dim_plan is not None
    loss_ref, g_ref = T.loss_and_grads(cfg, params, batch)
    with torch.no_grad():
        enc_ref = T.encode_sessions(cfg, params, batch)
    dbatch = {k: v.cuda() for k, v in batch.items()}
    lm.eval()
    with torch.no_grad():
        enc = lm.torch_model.encode_sessions(dbatch)
        last = lm.torch_model.encode_last(dbatch)
    d = cfg["d"]
    assert enc.shape[-1] == d or float(enc[..., d:].abs().max()) == 0.0     # the padded columns of the residual stream stay exact zeros
    _close(enc[..., :d], enc_ref, 5e-4, 5e-5, "encode_sessions")
    _close(last[:, :d], enc_ref[:, -1], 5e-4, 5e-5, "encode_last")
    lm.train()
    opt = hl.FlatAdam(lm.torch_model, lr=1e-2)
    opt.zero_grad()
    loss = lm.training_loss(dbatch)
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref)) <= 5e-5 * abs(float(loss_ref)) + 5e-6, (float(loss.detach()), float(loss_ref))
    for n, p in lm.torch_model.named_parameters():
        g = unpad_tensor(p.grad, p)
        _close(g, g_ref[n], 1e-2, 2e-5 if g_ref[n].abs().max() > 1e-6 else 1.0, f"grad {n}")
        if getattr(p, "_rt_axes", None) is not None:               # every gradient of a padded entry is an exact zero ...
            pad_only = 1.0 - pad_tensor(torch.ones(p._rt_real_shape, device=p.device), p)
            assert float((p.grad * pad_only).abs().max()) == 0.0, n
    before = {n: p.detach().clone() for n, p in lm.torch_model.named_parameters()}
    opt.step()
    torch.cuda.synchronize()
    for n, p in lm.torch_model.named_parameters():                 # ... and Adam leaves those entries at zero
        if getattr(p, "_rt_axes", None) is not None:
            pad_only = 1.0 - pad_tensor(torch.ones(p._rt_real_shape, device=p.device), p)
            assert float((p.detach() * pad_only).abs().max()) == 0.0, n
        assert not torch.equal(unpad_tensor(p.detach(), p), unpad_tensor(before[n], p)) or float(g_ref[n].abs().max()) == 0.0, n
    # the oracle's Adam on the oracle's gradients lands where the engine's step landed (real shapes)
    after_ref = T.AdamState(lr=1e-2).step(params, g_ref)
    got = lm.torch_model.state_dict()
    for n, _ in lm.torch_model.named_parameters():
        # (Adam's first step is lr * sign(g) wherever |g| >> eps: an entry whose true gradient is zero — the attention's key bias — moves
        # by +-lr on rounding noise in either implementation; compare where the gradient is a gradient)
        solid = g_ref[n].abs() > 1e-4 * float(g_ref[n].abs().max()) + 1e-7
        torch.testing.assert_close(got[n].cpu()[solid], after_ref[n][solid], rtol=2e-2, atol=2e-3, msg=lambda m, n=n: f"adam {n}: {m}")


def _frames(n_users=60, n_items=90, seed=0):
    rng = np.random.default_rng(seed)
    rows = []
    for u in range(n_users):
        n = int(rng.integers(3, 30))
        t0 = pd.Timestamp("2022-01-01") + pd.Timedelta(hours=int(rng.integers(0, 500)))
        ts = t0 + pd.to_timedelta(np.cumsum(rng.integers(1, 90, n)), unit="h")
        rows.append(pd.DataFrame({"user_id": u * 2 + 1, "item_id": rng.integers(0, n_items, n) + 1000, "weight": 1.0, "datetime": ts}))
    return pd.concat(rows, ignore_index=True)


@pytest.mark.parametrize("n_heads", [1, 2])
def test_hstu_tutorial_configuration_fits_recommends_and_round_trips(n_heads, tmp_path):
    """`HSTUModel(n_factors=50, n_heads=1 | 2)` — the configurations behind the reference's published HSTU numbers — through the public
    API: fit, recommend with a context, checkpoint in the real shapes, restored model recommends the same frame, fit_partial continues."""
    from rectools_amd.dataset import Dataset
    from rectools_amd.models import HSTUModel
    from rectools_amd.utils import get_context

    df = _frames()
    ds = Dataset.construct(df)
    model = HSTUModel(n_factors=50, n_heads=n_heads, n_blocks=2, session_max_len=16, batch_size=16, epochs=2, loss="sampled_softmax",
                      n_negatives=8, lr=1e-2, seed=5, dropout_rate=0.1, similarity_module_kwargs={"distance": "cosine"},
                      lightning_module_kwargs={"logits_t": 0.1})
    model.fit(ds)
    assert model.torch_model.dim_plan is not None and len(model.history) == 2
    assert all(np.isfinite(h["train_loss"]) for h in model.history)
    users = np.unique(df["user_id"])[:25]
    ctx = get_context(pd.DataFrame({"user_id": users, "datetime": pd.Timestamp("2023-06-01")}))
    reco = model.recommend(users, ds, k=5, filter_viewed=True, context=ctx)
    assert len(reco) == 25 * 5 and reco["score"].notna().all()
    sd = model.torch_model.state_dict()
    assert sd["item_model.item_net_blocks.0.ids_emb.weight"].shape[1] == 50
    assert sd["transformer_layers.stu_blocks.0.uvqk_proj"].shape == (50, 4 * 50)
    assert sd["transformer_layers.stu_blocks.0.norm_attn_output.weight"].shape == (50,)
    path = str(tmp_path / "hstu50.ckpt")
    model.save_to_checkpoint(path)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert ck["state_dict"]["torch_model.transformer_layers.stu_blocks.1.output_mlp.weight"].shape == (50, 50)
    assert ck["optimizer_states"][0]["state"][0]["exp_avg"].shape[1] == 50
    again = HSTUModel.load_from_checkpoint(path)
    reco2 = again.recommend(users, ds, k=5, filter_viewed=True, context=ctx)
    pd.testing.assert_frame_equal(reco, reco2)
    again.fit_partial(ds, max_epochs=1)
    assert again.epochs_done == 3


@pytest.mark.parametrize("cls_name,kw", [("SASRecModel", dict(n_factors=50, n_heads=2)), ("BERT4RecModel", dict(n_factors=36, n_heads=3)),
                                         ("SASRecModel", dict(n_factors=30, n_heads=1, loss="gBCE", n_negatives=4))])
def test_odd_sizes_through_the_model_api(cls_name, kw):
    from rectools_amd import models
    from rectools_amd.dataset import Dataset

    df = _frames(seed=3)
    ds = Dataset.construct(df)
    model = getattr(models, cls_name)(n_blocks=1, session_max_len=12, batch_size=16, epochs=2, lr=1e-2, seed=1, **kw)
    model.fit(ds)
    users = np.unique(df["user_id"])[:10]
    reco = model.recommend(users, ds, k=4, filter_viewed=False)
    assert len(reco) == 40 and reco["score"].notna().all()
    blob = model.dumps()
    reco2 = type(model).loads(blob).recommend(users, ds, k=4, filter_viewed=False)
    pd.testing.assert_frame_equal(reco, reco2)
