"""`nn.DimPlan` on the host: which sizes are padded and how, and that state dicts / Adam moments of a padded model speak the REAL shapes
(every real entry survives the round trip, every padded entry is zero).  What the padded model COMPUTES is compared with the oracle at
the real sizes on the GPU (tests/test_dim_plan_gpu.py)."""
import pytest
import torch

from test_transformer_gpu import _random_case, build_hip_model

ODD = [
    ("hstu_d50_h1", _random_case("stu", "sampled_softmax", "cosine", 24, 50, 1, 3, 90, 5, 31, logits_t=0.05)),     # the reference's
    ("hstu_d50_h2", _random_case("stu", "sampled_softmax", "cosine", 24, 50, 2, 3, 90, 5, 32, logits_t=0.05)),     # published HSTU sizes
    ("sasrec_d50_h2", _random_case("sasrec", "softmax", "dot", 20, 50, 2, 3, 80, 1, 33)),
    ("sasrec_d20_h1_gbce", _random_case("sasrec", "gBCE", "cosine", 17, 20, 1, 3, 80, 4, 34)),
    ("bert_d36_h3", _random_case("preln", "softmax", "dot", 21, 36, 3, 3, 70, 1, 35, causal=False, keypad=True)),
    ("ligr_d40_h2", _random_case("ligr", "sampled_softmax", "cosine", 19, 40, 2, 3, 70, 6, 36, logits_t=0.1,
                                 layer_kwargs=dict(ff_factors_multiplier=4, ff_activation="swiglu", bias_in_ff=False))),
    ("stu_lin12_att20", _random_case("stu", "BCE", "dot", 22, 32, 2, 3, 70, 3, 37, linear_hidden_dim=12, attention_dim=20)),
    ("stu_lin24_att8_buckets16", _random_case("stu", "sampled_softmax", "cosine", 18, 32, 2, 3, 70, 3, 38, linear_hidden_dim=24,
                                              attention_dim=8, num_buckets=16)),
]


def test_plan_sizes():
    from rectools_amd.nn import DimPlan

    assert DimPlan.make(256, 4) is None and DimPlan.make(64, 4) is None and DimPlan.make(512, 4) is None
    assert DimPlan.make(256, 4, "stu") is None and DimPlan.make(100, 1, "stu", 8, 8) is None
    p = DimPlan.make(50, 1, "stu")
    assert (p.d, p.d_pad, p.qk, p.vo, p.hd_pad) == (50, 56, 50, 50, 56) and p.row_cols == (56, 50) and p.head_cols == (56, 50)
    p = DimPlan.make(50, 2, "stu")
    assert (p.d_pad, p.qk, p.hd_pad) == (56, 25, 32)
    p = DimPlan.make(50, 2)
    assert (p.d_pad, p.hd_pad) == (64, 32) and p.row_cols == (64, 50)
    p = DimPlan.make(64, 2, "stu", 24, 40)
    assert (p.d_pad, p.qk, p.vo, p.hd_pad) == (64, 40, 24, 40) and p.row_cols is None and p.head_cols == (40, 24)
    with pytest.raises(NotImplementedError, match="128"):
        DimPlan.make(260, 2)


@pytest.mark.parametrize("name,case", ODD, ids=[c[0] for c in ODD])
def test_state_dict_speaks_the_real_shapes(name, case):
    from rectools_amd import lightning as hl
    from rectools_amd.nn import unpad_tensor

    cfg, _ = case
    torch.manual_seed(3)
    real = build_hip_model(cfg, device="cpu", real_size=True)
    hl.xavier_normal_init(real.torch_model)
    with torch.no_grad():
        for p in real.torch_model.parameters():
            if p.dim() == 1:
                p.add_(torch.rand_like(p) + 0.5)        # (no zeros among the real entries)
    want = {k: v.clone() for k, v in real.torch_model.state_dict().items()}
    padded = build_hip_model(cfg, params=want, device="cpu")
    plan = padded.torch_model.dim_plan
    assert padded.torch_model.d_real == cfg["d"] and plan.d == cfg["d"]
    got = padded.torch_model.state_dict()
    assert list(got) == list(want)
    n_padded = 0
    for k in want:
        assert got[k].shape == want[k].shape, k
        assert torch.equal(got[k], want[k]), k
    for n, p in padded.torch_model.named_parameters():
        if getattr(p, "_rt_axes", None) is not None:
            n_padded += 1
            assert tuple(p._rt_real_shape) == tuple(want[n].shape)
            assert int((p != 0).sum()) == want[n].numel(), n          # everything outside the real entries is zero
            assert torch.equal(unpad_tensor(p.detach(), p), want[n])
        else:
            assert p.shape == want[n].shape, n
    assert n_padded > 0
    # the lightning module's prefixed state dict goes through the same hooks
    full = padded.state_dict()
    assert all(full["torch_model." + k].shape == want[k].shape for k in want)


def test_adam_state_of_a_padded_model_is_written_and_read_in_the_real_shapes():
    from rectools_amd import checkpoint as ckpt
    from rectools_amd import lightning as hl

    cfg, _ = ODD[1][1]
    torch.manual_seed(4)
    lm = build_hip_model(cfg, device="cpu")
    opt = hl.FlatAdam(lm.torch_model, lr=1e-3)
    g = torch.Generator().manual_seed(9)
    opt.step_count = 3
    real_m = {}
    for (n, p), ofs in zip(lm.torch_model.named_parameters(), opt._offsets):
        shape = tuple(getattr(p, "_rt_real_shape", p.shape))
        real_m[n] = (torch.randn(shape, generator=g), torch.rand(shape, generator=g))
    names = [n for n, _ in lm.torch_model.named_parameters()]
    sd = {"state": {i: {"step": torch.tensor(3.0), "exp_avg": real_m[n][0], "exp_avg_sq": real_m[n][1]} for i, n in enumerate(names)},
          "param_groups": [{"lr": 1e-3, "betas": (0.9, 0.98), "eps": 1e-8, "params": list(range(len(names)))}]}
    ckpt.load_adam_state_dict(opt, sd, names, names)
    back = ckpt.adam_state_dict(opt)
    for i, n in enumerate(names):
        assert torch.equal(back["state"][i]["exp_avg"], real_m[n][0]), n
        assert torch.equal(back["state"][i]["exp_avg_sq"], real_m[n][1]), n
    # padded entries of the moments are zero
    for (n, p), ofs in zip(lm.torch_model.named_parameters(), opt._offsets):
        assert int((opt.m[ofs:ofs + p.numel()] != 0).sum()) <= real_m[n][0].numel()
