"""GPU tests of the packed (padding-free) BERT4Rec path: the bidirectional packed attention
(`rt_mha_varlen_bidir_*`) against torch autograd on every session's own keys (what the reference's key-padding mask leaves,
torch_backbone.py:254 / bert4rec.py:200), `rt_collate_packed_bert` against the padded device collate fed the same draws, the packed
Pre-LN stack against the padded one (loss, gradients, recommend encodings), and the product loop with and without it."""
import math

import numpy as np
import pandas as pd
import pytest
import torch

from test_packed_gpu import _drop_mask

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("H,hd,p", [(2, 32, 0.0), (2, 64, 0.25), (1, 32, 0.3), (4, 64, 0.0)])
def test_packed_qkv_attention_pair_equals_autograd_per_session(H, hd, p, causal):
    """Forward (dropout, lse) and backward of `ops.mha_varlen_qkv` on one packed [Np, 3d] projection: every query against the keys of
    its own session (all of them, or the causal prefix), same dropout masks; dq / dk / dv land in the column blocks of one buffer."""
    from rectools_amd import ops

    torch.manual_seed(11 * H + hd + causal)
    window, d = 96, H * hd
    lens = [1, 96, 32, 33, 64, 7, 95, 50, 17, 16]
    B, N = len(lens), sum(lens)
    Np = (N + 127) // 128 * 128
    qkv0 = torch.randn(Np, 3 * d) * 0.7
    gout = torch.randn(Np, d); gout[N:] = 0
    cu = torch.tensor(np.r_[0, np.cumsum(lens)], dtype=torch.int64).cuda()
    ops.RNG.seed, ops.RNG.step = 777, 2
    ops.RNG._stream = 0
    s0, sid = (ops.RNG.seed + 0x9E3779B97F4A7C15 * ops.RNG.step) & 0xFFFFFFFFFFFFFFFF, 1
    seed = (s0 + 0xD1B54A32D192ED03 * sid) & 0xFFFFFFFFFFFFFFFF

    qd = qkv0.cuda().requires_grad_(True)
    out = ops.mha_varlen_qkv(qd, cu, B, H, window, causal, p)
    out.backward(gout.cuda())

    x = qkv0.double().requires_grad_(True)
    outs, r0 = [], 0
    for b, n in enumerate(lens):
        qh, kh, vh = (x[r0:r0 + n, c * d:(c + 1) * d].view(n, H, hd).transpose(0, 1) for c in range(3))
        sc = qh @ kh.transpose(-1, -2) / math.sqrt(hd)
        if causal:
            sc = sc.masked_fill(~torch.tril(torch.ones(n, n, dtype=torch.bool)), float("-inf"))
        pr = torch.softmax(sc, -1)
        mask = torch.stack([torch.from_numpy(_drop_mask(seed, b * H + h, n, n, p)) for h in range(H)])
        outs.append(((pr * mask) @ vh).transpose(0, 1).reshape(n, d))
        r0 += n
    ref = torch.cat(outs)
    (ref * gout[:N].double()).sum().backward()
    tol = dict(rtol=3e-4, atol=3e-5)
    torch.testing.assert_close(out[:N].detach().cpu().double(), ref.detach(), **tol)
    torch.testing.assert_close(qd.grad[:N].cpu().double(), x.grad[:N], **tol)
    assert float(qd.grad[N:].abs().max()) == 0.0 and float(out[N:].detach().abs().max()) == 0.0          # the unused tail stays zero
    inf = ops.mha_varlen_qkv_infer(qkv0.cuda(), cu, B, H, window, causal)
    if p == 0:
        torch.testing.assert_close(inf[:N].cpu().double(), ref.detach(), **tol)


@pytest.mark.parametrize("H,hd,window,lens", [(4, 64, 200, [200, 1, 199, 200, 31, 128]),          # BASELINE config 3's head and window
                                               (2, 128, 200, [200, 65, 1, 33, 130]),                 # a 128-column head (eSASRec's)
                                               (2, 64, 400, [400, 257, 64, 399, 5])])                # a window the whole-image kernels refused
def test_bidir_attention_at_the_catalog_window_and_its_limits(H, hd, window, lens):
    """The streamed kernels (K4v3) serve head sizes 32 / 64 / 128 at any window — forward values and (through autograd of the fp64
    restatement) all three gradients; head sizes they do not tile are refused with a status, not served some other way."""
    from rectools_amd import ops

    torch.manual_seed(0)
    d = H * hd
    N = sum(lens); Np = (N + 127) // 128 * 128
    qkv = (torch.randn(Np, 3 * d) * 0.5).cuda().requires_grad_(True)
    cu = torch.tensor(np.r_[0, np.cumsum(lens)], dtype=torch.int64).cuda()
    out = ops.mha_varlen_qkv(qkv, cu, len(lens), H, window, False, 0.0)
    gout = torch.randn(Np, d).cuda()
    gout[N:] = 0
    out.backward(gout)
    x = qkv.detach().cpu().double().requires_grad_(True)
    r0, refs = 0, []
    for n in lens:
        qh, kh, vh = (x[r0:r0 + n, c * d:(c + 1) * d].view(n, H, hd).transpose(0, 1) for c in range(3))
        refs.append((torch.softmax(qh @ kh.transpose(-1, -2) / hd ** 0.5, -1) @ vh).transpose(0, 1).reshape(n, d))
        r0 += n
    ref = torch.cat(refs)
    torch.testing.assert_close(out[:N].detach().cpu().double(), ref.detach(), rtol=3e-4, atol=3e-5)
    ref.backward(gout[:N].cpu().double())
    g, gr = qkv.grad[:N].cpu().double(), x.grad[:N]
    torch.testing.assert_close(g, gr, rtol=3e-4, atol=3e-4 * float(gr.abs().max()))
    assert ops.mha_bidir_supported(4, 256, 200) and ops.mha_bidir_supported(2, 256, 200) and ops.mha_bidir_supported(4, 256, 4000)
    assert not ops.mha_bidir_supported(16, 256, 200) and not ops.mha_bidir_supported(1, 256, 200)          # heads of 16 / 256 columns
    bad = torch.zeros(128, 3 * 16, device="cuda")
    with pytest.raises(NotImplementedError):
        ops.mha_varlen_qkv_infer(bad, cu[:2].clone(), 1, 1, 64, False)        # hd = 16


@pytest.mark.parametrize("train", [True, False])
def test_collate_packed_bert_equals_the_padded_collate(train):
    """`rt_collate_packed_bert` against `rt_collate` modes 3 / 4 (pinned to the reference's collate in tests/test_host_path.py) fed
    the same draws: session b's packed rows are the non-pad columns of row b, bit for bit."""
    from rectools_amd import ops

    rng = np.random.default_rng(9)
    L, V, mask_id = 50, 1000, 1
    lens = np.r_[rng.integers(1, 3 * L, 300), 1, L - 1, L, L + 1, 2]
    offsets_h = np.r_[0, np.cumsum(lens)].astype(np.int64)
    offsets = torch.tensor(offsets_h).cuda()
    items = torch.tensor(rng.integers(2, V, int(lens.sum())), dtype=torch.int64).cuda()
    weights = torch.tensor(rng.random(int(lens.sum())).astype(np.float32) + 0.5).cuda()
    idx_h = rng.permutation(len(lens))[:200].astype(np.int64)
    idx = torch.tensor(idx_h).cuda()
    B = len(idx_h)
    slen = offsets_h[idx_h + 1] - offsets_h[idx_h]
    n_h = np.minimum(slen, L) if train else np.minimum(slen, L - 1) + 1
    cu_h = np.r_[0, np.cumsum(n_h)].astype(np.int64)
    N = int(cu_h[-1]); rows = (N + 127) // 128 * 128
    cu = torch.tensor(cu_h).cuda()
    probs = torch.rand(B, L, device="cuda")
    rand_ids = torch.randint(2, V, (B, L), device="cuda")
    xp = torch.empty(B, L, dtype=torch.int64, device="cuda")
    yp, wp = (torch.empty_like(xp), torch.empty(B, L, device="cuda")) if train else (None, None)
    ops._c("rt_collate", offsets, items, weights, None, idx, B, L, 3 if train else 4, probs if train else None,
           rand_ids if train else None, 0.3, mask_id, xp, yp, wp, None)
    got = ops.collate_packed_bert(offsets, items, weights, idx, cu, rows, L, train, mask_id, probs, rand_ids, 0.3)
    x, dist = got[0], got[-1]
    if train:      # the same draws through a slot table: every session reads the row it is pointed at
        perm = torch.randperm(B, device="cuda")
        again = ops.collate_packed_bert(offsets, items, weights, idx, cu, rows, L, True, mask_id, probs[perm], rand_ids[perm], 0.3,
                                        draw_rows=torch.argsort(perm))
        assert all(torch.equal(a, b) for a, b in zip(got, again))
    keep = torch.arange(L, device="cuda")[None, :] >= (L - torch.tensor(n_h, device="cuda"))[:, None]      # the non-pad columns
    assert torch.equal(x[:N], xp[keep])
    assert torch.equal(dist[:N], (L - 1 - torch.arange(L, device="cuda"))[None, :].expand(B, L)[keep])
    assert int(x[N:].abs().sum()) == 0 and int(dist[N:].abs().sum()) == 0
    if train:
        y, yw = got[1], got[2]
        assert torch.equal(y[:N], yp[keep]) and torch.equal(yw[:N], wp[keep])
        assert int((y[:N] != 0).sum()) > 0 and int((x[:N] == mask_id).sum()) > 0                 # the draws did pick positions
        assert int(y[N:].abs().sum()) == 0 and float(yw[N:].abs().sum()) == 0
    else:
        assert bool((x[cu[1:] - 1] == mask_id).all())


def _bert_stack(V, L, d, H, n_blocks, p, causal=False):
    from rectools_amd import nn as hnn

    item_model = hnn.SumOfEmbeddingsConstructor(V, [hnn.IdEmbeddingsItemNet(d, V, 0.0)])
    return hnn.TransformerTorchBackbone(H, p, item_model, hnn.LearnableInversePositionalEncoding(True, L, d),
                                        hnn.PreLNTransformerLayers(n_blocks, d, H, p), hnn.DistanceSimilarityModule(), causal, True)


@pytest.mark.parametrize("loss,causal", [("softmax", False), ("sampled_softmax", False), ("softmax", True)])
def test_packed_bert_loss_and_gradients_equal_the_padded_batch(loss, causal):
    """`training_loss_packed` on the Pre-LN stack with key-padding masks against `training_loss` on the padded batch of the same
    sessions (dropout 0): the loss and every parameter gradient; with and without the tail riding along as one more session."""
    from rectools_amd import lightning as hl

    torch.manual_seed(4)
    V, L, d, H, n_neg = 200, 48, 64, 2, 5
    lm = hl.TransformerLossModule(_bert_stack(V, L, d, H, 2, 0.0, causal), loss, n_neg).cuda().train()
    for prm in lm.parameters():
        if prm.dim() == 1:
            torch.nn.init.normal_(prm, std=0.3)
    assert lm.torch_model.transformer_layers.packed_ok(d, L, causal, True) and not lm.torch_model.transformer_layers.packed_ok(d, L, causal, False)
    rng = np.random.default_rng(0)
    lens = np.r_[rng.integers(1, L + 1, 40), 1, L, L - 1, 33, 32]
    B = len(lens)
    xp = torch.zeros(B, L, dtype=torch.int64); yp = torch.zeros_like(xp); wp = torch.zeros(B, L)
    for b, n in enumerate(lens):
        xp[b, L - n:] = torch.tensor(rng.integers(1, V, n))
        yp[b, L - n:] = torch.tensor(rng.integers(2, V, n) * (rng.random(n) < 0.3))
        wp[b, L - n:] = torch.tensor(rng.random(n).astype(np.float32) + 0.5)
    xp, yp, wp = xp.cuda(), yp.cuda(), wp.cuda()
    real = xp != 0
    N = int(real.sum()); tail = (N + 127) // 128 * 128 - N
    pad = lambda t: torch.nn.functional.pad(t, (0, tail))   # noqa: E731
    cu = torch.tensor(np.r_[0, np.cumsum(lens)], dtype=torch.int64).cuda()
    dist_p = (L - 1 - torch.arange(L, device="cuda"))[None, :].expand(B, L)
    neg_p = torch.tensor(rng.integers(1, V, (B, L, n_neg)), dtype=torch.int64).cuda()
    padded = {"x": xp, "y": yp, "yw": wp}
    packed = {"x": pad(xp[real]), "y": pad(yp[real]), "yw": pad(wp[real]), "dist": pad(dist_p[real]), "cu": cu, "window": L}
    if loss != "softmax":
        padded["negatives"], packed["negatives"] = neg_p, pad(neg_p[real].t()).t().contiguous()
    lp = lm.training_loss(padded); lp.backward()
    g_padded = {k: v.grad.clone() for k, v in lm.named_parameters() if v.grad is not None}
    variants = [dict(packed)]
    if 0 < tail <= L:
        variants.append(dict(packed, n_rows=N, cu_attn=torch.cat([cu, cu[-1:] + tail])))
    for pb in variants:
        for v in lm.parameters():
            v.grad = None
        lq = lm.training_loss_packed(pb); lq.backward()
        torch.testing.assert_close(lq.detach(), lp.detach(), rtol=1e-4, atol=1e-6)
        for k, v in lm.named_parameters():
            if k in g_padded:
                got = v.grad if v.grad is not None else torch.zeros_like(v)
                torch.testing.assert_close(got, g_padded[k], rtol=2e-3, atol=2e-5 * (float(g_padded[k].abs().max()) + 1e-12),
                                           msg=lambda s, k=k: f"gradient of {k}: {s}")


def _bert_data(seed=1):
    from rectools_amd.dataset import Dataset

    rng = np.random.default_rng(seed)
    n_users, n_items, n = 150, 90, 5000
    df = pd.DataFrame({"user_id": rng.integers(0, n_users, n), "item_id": rng.integers(0, n_items, n) + 100, "weight": 1.0,
                       "datetime": pd.to_datetime("2022-01-01") + pd.to_timedelta(rng.integers(0, 500_000, n), unit="m")})
    return Dataset.construct(df), np.arange(n_users)


def test_packed_bert_train_loop_takes_the_steps_of_the_padded_loop(monkeypatch):
    """BERT4RecModel's product loop with packed batches (host-cut row offsets, `rt_collate_packed_bert` on the draws of the padded
    collate, packed Pre-LN blocks) against the padded loop: the same masked positions, equal losses step by step, equal parameters."""
    from rectools_amd.models import BERT4RecModel

    ds, _ = _bert_data()
    kw = dict(n_factors=64, n_blocks=2, n_heads=2, session_max_len=24, lr=0.005, batch_size=32, dropout_rate=0.0, seed=5, epochs=1,
              mask_prob=0.3)
    losses, params = {}, {}
    for packed in ("0", "1"):
        monkeypatch.setenv("RT_PACKED_TRAIN", packed)
        m = BERT4RecModel(**kw)
        m._build_model_from_dataset(ds)
        loop = m.training_loop()
        assert loop.packed == (packed == "1")
        m.lightning_model.train()
        torch.manual_seed(123)                                         # the masking draws come from torch's device generator
        loop.begin_epoch(0)
        losses[packed] = [float(loop.step()) for _ in range(9)]
        params[packed] = {k: v.detach().clone() for k, v in m.torch_model.state_dict().items()}
    np.testing.assert_allclose(losses["1"], losses["0"], rtol=2e-4)
    d = kw["n_factors"]
    for k, v in params["0"].items():
        a, b = params["1"][k], v
        if k.endswith("in_proj_bias"):    # the key bias has no gradient beyond rounding noise, which Adam turns into O(lr) steps: q and v thirds
            a, b = torch.cat([a[:d], a[2 * d:]]), torch.cat([b[:d], b[2 * d:]])
        torch.testing.assert_close(a, b, rtol=5e-3, atol=5e-4, msg=lambda s, k=k: f"{k}: {s}")


def test_bert_recommend_with_packed_encoder_equals_padded(monkeypatch):
    from rectools_amd.models import BERT4RecModel

    ds, users = _bert_data(2)
    model = BERT4RecModel(n_factors=64, n_blocks=2, n_heads=2, session_max_len=20, epochs=1, batch_size=64, seed=3)
    model.fit(ds)
    monkeypatch.setenv("RT_PACKED", "1")
    fast = model.recommend(users=users, dataset=ds, k=7, filter_viewed=True)
    monkeypatch.setenv("RT_PACKED", "0")
    slow = model.recommend(users=users, dataset=ds, k=7, filter_viewed=True)
    assert len(fast) == len(slow) > 0
    same = (fast["item_id"].values == slow["item_id"].values).mean()
    assert same > 0.995                                                # fp32 rounding may swap near-ties
    np.testing.assert_allclose(np.sort(fast["score"].values), np.sort(slow["score"].values), rtol=1e-4, atol=1e-5)


def test_pre_ln_stack_without_key_padding_mask_keeps_the_padded_window(monkeypatch):
    """Without the key-padding mask the pad rows of a Pre-LN stack carry state real queries read: no packed form, the loop says so."""
    from rectools_amd.models import BERT4RecModel

    ds, _ = _bert_data()
    monkeypatch.setenv("RT_PACKED_TRAIN", "1")
    m = BERT4RecModel(n_factors=64, n_blocks=1, n_heads=2, session_max_len=24, batch_size=32, use_key_padding_mask=False, seed=1)
    m._build_model_from_dataset(ds)
    assert m.training_loop().packed is False


@pytest.mark.parametrize("p,causal", [(0.0, False), (0.25, False), (0.25, True)])
def test_native_preln_block_equals_the_python_block(p, causal, monkeypatch):
    """`rt_preln_block_packed_fwd / _bwd` (the block's launch sequence issued by compiled code, csrc/rt_block.hip) against the Python
    path that issues the same kernels one by one with the same dropout streams: output, input gradient, every parameter gradient; with
    the parameters in a FlatAdam buffer the native path also runs on the pre-split weight planes."""
    from rectools_amd import lightning as hl
    from rectools_amd import nn as hnn
    from rectools_amd import ops

    torch.manual_seed(17)
    d, H, window = 64, 2, 40
    lens = [40, 1, 17, 33, 8, 25, 39, 2, 40, 31]
    B, N = len(lens), sum(lens)
    Np = (N + 127) // 128 * 128
    cu = torch.tensor(np.r_[0, np.cumsum(lens)], dtype=torch.int64).cuda()
    stack = hnn.PreLNTransformerLayers(2, d, H, p).cuda().train()
    for prm in stack.parameters():
        if prm.dim() == 1:
            torch.nn.init.normal_(prm, std=0.3)
    opt = hl.FlatAdam(stack, lr=1e-3)                      # parameters become views of one flat buffer: weight planes apply
    x0 = torch.randn(Np, d); x0[N:] = 0
    gout = torch.randn(Np, d); gout[N:] = 0
    res = {}
    for name, native in (("native", "1"), ("python", "0")):
        monkeypatch.setenv("RT_NATIVE_BLOCK", native)
        opt.zero_grad()
        ops.RNG.seed, ops.RNG.step, ops.RNG._stream = 4242, 9, 0
        x = x0.cuda().requires_grad_(True)
        out = stack.forward_packed_train(x, cu, B, window, True, rows_real=N, causal=causal)
        out.backward(gout.cuda())
        ops.join_side_streams()
        torch.cuda.synchronize()
        res[name] = (out.detach()[:N].clone(), x.grad[:N].clone(), {k: v.grad.clone() for k, v in stack.named_parameters()})
    torch.testing.assert_close(res["native"][0], res["python"][0], rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(res["native"][1], res["python"][1], rtol=2e-5, atol=2e-6)
    for k in res["python"][2]:
        torch.testing.assert_close(res["native"][2][k], res["python"][2][k], rtol=2e-4, atol=2e-6 * (float(res["python"][2][k].abs().max()) + 1e-12),
                                   msg=lambda s, k=k: f"gradient of {k}: {s}")
