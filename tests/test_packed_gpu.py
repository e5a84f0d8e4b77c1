"""GPU tests of the packed (padding-free) recommend() encoder: `rt_mha_varlen_fwd` / `rt_mha_varlen_last_fwd`
against a plain torch fp32 restatement of the PADDED window (pad key / value rows = the projection biases), the packed SASRec
stack against the padded `encode_last`, and recommend() with and without it."""
import math

import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu


def _padded_reference(Q, K, V, lens, window, H, bk, bv):
    """Per session: left-padded window, causal mask, pad key / value rows = bk / bv (absent when bk is None: pads masked)."""
    outs, r0 = [], 0
    d = Q.shape[1]
    hd = d // H
    for n in lens:
        q, k, v = Q[r0:r0 + n].double(), K[r0:r0 + n].double(), V[r0:r0 + n].double()
        n_pad = window - n if bk is not None else 0
        if n_pad > 0:
            k = torch.cat([bk.double()[None].expand(n_pad, d), k]); v = torch.cat([bv.double()[None].expand(n_pad, d), v])
        qh, kh, vh = (t.view(-1, H, hd).transpose(0, 1) for t in (q, k, v))
        sc = qh @ kh.transpose(-1, -2) / math.sqrt(hd)                            # [H, n, n_pad + n]
        keep = torch.ones(n, n_pad + n, dtype=torch.bool)
        keep[:, n_pad:] = torch.tril(torch.ones(n, n, dtype=torch.bool))
        sc = sc.masked_fill(~keep, float("-inf"))
        outs.append((torch.softmax(sc, -1) @ vh).transpose(0, 1).reshape(n, d))
        r0 += n
    return torch.cat(outs).float()


@pytest.mark.parametrize("H,hd", [(2, 32), (4, 64), (1, 64)])
@pytest.mark.parametrize("pads", [True, False])
def test_mha_varlen_fwd_and_last_equal_the_padded_window(H, hd, pads):
    from rectools_amd import ops

    torch.manual_seed(H * 100 + hd + pads)
    window, d = 200, H * hd
    lens = [1, 200, 32, 33, 64, 7, 199, 128, 96, 2]
    N = sum(lens)
    Np = (N + 127) // 128 * 128
    Q, K, V = (torch.randn(Np, d) * 0.7 for _ in range(3))
    bk, bv = (torch.randn(d) * 0.5, torch.randn(d) * 0.5) if pads else (None, None)
    cu = torch.tensor(np.r_[0, np.cumsum(lens)], dtype=torch.int64).cuda()
    ref = _padded_reference(Q[:N], K[:N], V[:N], lens, window, H, bk, bv)
    Qd, KVd = Q.cuda(), torch.cat([K, V], 1).cuda().contiguous()                # k | v column blocks of one [Np, 2d] buffer
    bkd, bvd = (bk.cuda(), bv.cuda()) if pads else (None, None)
    out = torch.zeros(Np, d, device="cuda")
    ops._c("rt_mha_varlen_fwd", Qd, d, KVd, 2 * d, KVd[:, d:], 2 * d, cu, bkd, bvd, len(lens), H, hd, window, window, out, d)
    torch.testing.assert_close(out[:N].cpu(), ref, rtol=2e-4, atol=2e-5)
    assert float(out[N:].abs().max()) == 0.0 if Np > N else True                 # tail rows untouched
    last_rows = cu[1:] - 1
    out_last = torch.empty(len(lens), d, device="cuda")
    ops._c("rt_mha_varlen_last_fwd", Qd.index_select(0, last_rows).contiguous(), d, KVd, 2 * d, KVd[:, d:], 2 * d, cu, bkd, bvd,
           len(lens), H, hd, window, window, out_last, d)
    torch.testing.assert_close(out_last.cpu(), ref[(cu[1:] - 1).cpu()], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("keypad", [False, True])
@pytest.mark.parametrize("n_blocks,H,d", [(2, 2, 64), (1, 4, 256), (3, 1, 32)])
def test_encode_last_packed_equals_padded_encode_last(n_blocks, H, d, keypad):
    from rectools_amd import nn as hnn

    torch.manual_seed(n_blocks + H + d)
    V, L = 300, 70
    item_model = hnn.SumOfEmbeddingsConstructor(V, [hnn.IdEmbeddingsItemNet(d, V, 0.0)])
    backbone = hnn.TransformerTorchBackbone(
        H, 0.0, item_model, hnn.LearnableInversePositionalEncoding(True, L, d),
        hnn.SASRecTransformerLayers(n_blocks, d, H, 0.0), hnn.DistanceSimilarityModule(), True, keypad).cuda().eval()
    for prm in backbone.parameters():                                             # biases matter here: make them non-trivial
        if prm.dim() == 1:
            torch.nn.init.normal_(prm, std=0.3)
    rng = np.random.default_rng(0)
    lens = np.r_[rng.integers(1, 2 * L, 150), 1, L, L + 1, 32, 64]               # shorter than, equal to and longer than the window
    offsets = torch.tensor(np.r_[0, np.cumsum(lens)], dtype=torch.int64).cuda()
    items = torch.tensor(rng.integers(1, V, int(lens.sum())), dtype=torch.int64).cuda()
    rows = torch.tensor(rng.permutation(len(lens)), dtype=torch.int64).cuda()
    with torch.no_grad():
        assert backbone.can_encode_packed(d, L)
        packed = backbone.encode_last_packed(offsets, items, rows, L)
        x = torch.zeros(len(rows), L, dtype=torch.int64, device="cuda")          # the padded window of the same sessions
        for b, r in enumerate(rows.tolist()):
            tail = items[offsets[r]:offsets[r + 1]][-L:]
            x[b, L - len(tail):] = tail
        padded = backbone.encode_last({"x": x})
        # a recommend() call's cache: the FIRST block's keys | values gathered from projected tables (rows >= catalog rows here)
        cache = {}
        tabled = backbone.encode_last_packed(offsets, items, rows, L, cache=cache)
        assert ("kv_tables" in cache) == (n_blocks > 1)
        again = backbone.encode_last_packed(offsets, items, rows, L, cache=cache)      # (the second launch of a call reuses them)
    torch.testing.assert_close(packed, padded, rtol=2e-4, atol=2e-5 * float(padded.abs().max()))
    torch.testing.assert_close(tabled, padded, rtol=2e-4, atol=2e-5 * float(padded.abs().max()))
    assert torch.equal(tabled, again)


def test_recommend_with_packed_encoder_equals_padded(monkeypatch):
    from rectools_amd.dataset import Dataset
    from rectools_amd.models import SASRecModel

    rng = np.random.default_rng(1)
    n_users, n_items, n = 200, 90, 5000
    df = pd.DataFrame({"user_id": rng.integers(0, n_users, n), "item_id": rng.integers(0, n_items, n), "weight": 1.0,
                       "datetime": pd.to_datetime("2022-01-01") + pd.to_timedelta(rng.integers(0, 50_000, n), unit="m")})
    ds = Dataset.construct(df)
    model = SASRecModel(n_factors=64, n_blocks=2, n_heads=2, session_max_len=16, lr=0.01, batch_size=64, epochs=2,
                        loss="sampled_softmax", n_negatives=4, seed=1).fit(ds)
    users = ds.user_id_map.external_ids
    fast = model.recommend(users=users, dataset=ds, k=7, filter_viewed=True)
    monkeypatch.setenv("RT_PACKED", "0")
    slow = model.recommend(users=users, dataset=ds, k=7, filter_viewed=True)
    assert fast[["user_id", "item_id", "rank"]].equals(slow[["user_id", "item_id", "rank"]])
    np.testing.assert_allclose(fast["score"].values, slow["score"].values, rtol=1e-4, atol=1e-5)


def _drop_mask(seed, bh, n, n_all, p):
    """The kernels' dropout mask of one (session, head): [n queries, n_all keys] in {0, 1/(1-p)} (rt_attention_varlen.hip:drop_hash)."""
    if p <= 0:
        return np.ones((n, n_all))
    M32 = np.uint64(0xFFFFFFFF)
    q = np.arange(n, dtype=np.uint64)[:, None]
    key = np.arange(n_all, dtype=np.uint64)[None, :]
    x = (np.uint64(seed) & M32) ^ ((q * np.uint64(0x9E3779B1)) & M32) ^ (((key >> np.uint64(1)) * np.uint64(0x85EBCA77)) & M32) \
        ^ ((np.uint64(bh) * np.uint64(0xC2B2AE3D)) & M32) ^ ((np.uint64(seed) >> np.uint64(32)) & M32)
    x = x ^ (x >> np.uint64(16)); x = (x * np.uint64(0x7FEB352D)) & M32; x = x ^ (x >> np.uint64(15))
    bits = (x >> (np.uint64(16) * (key & np.uint64(1)))) & np.uint64(0xFFFF)
    thr = np.uint64(int(np.float32(p) * np.float32(65536.0)))
    return (bits >= thr).astype(np.float64) / (1.0 - p)


@pytest.mark.parametrize("H,hd,p,pads", [(2, 32, 0.0, True), (2, 64, 0.25, True), (1, 32, 0.3, False), (4, 64, 0.2, True)])
def test_mha_varlen_training_pair_equals_autograd_on_the_padded_window(H, hd, p, pads):
    """Forward (dropout, lse) and backward (dq, dk, dv, the pad keys' shares of d_bk and d_bv) of the packed attention against torch
    autograd on the explicit padded window — pad keys as real rows b_k / b_v behind the session's keys, same dropout masks."""
    from rectools_amd import ops

    torch.manual_seed(7 * H + hd)
    window, d = 96, H * hd
    lens = [1, 96, 32, 33, 64, 7, 95, 50]
    B, N = len(lens), sum(lens)
    Np = (N + 127) // 128 * 128
    q0, kv0 = torch.randn(Np, d) * 0.7, torch.randn(Np, 2 * d) * 0.7
    bk0, bv0 = torch.randn(d) * 0.5, torch.randn(d) * 0.5
    gout = torch.randn(Np, d); gout[N:] = 0
    cu = torch.tensor(np.r_[0, np.cumsum(lens)], dtype=torch.int64).cuda()
    ops.RNG.seed, ops.RNG.step = 12345, 3
    ops.RNG._stream = 0
    s0, sid = (ops.RNG.seed + 0x9E3779B97F4A7C15 * ops.RNG.step) & 0xFFFFFFFFFFFFFFFF, 1      # what RNG.next() will hand out
    seed = (s0 + 0xD1B54A32D192ED03 * sid) & 0xFFFFFFFFFFFFFFFF

    qd, kvd = q0.cuda().requires_grad_(True), kv0.cuda().requires_grad_(True)
    bkd, bvd = (bk0.cuda().requires_grad_(True), bv0.cuda().requires_grad_(True)) if pads else (None, None)
    out = ops.mha_varlen(qd, kvd, bkd, bvd, cu, B, H, window, p)
    out.backward(gout.cuda())

    q, kv = q0.double().requires_grad_(True), kv0.double().requires_grad_(True)
    bk, bv = bk0.double().requires_grad_(True), bv0.double().requires_grad_(True)
    outs, r0 = [], 0
    for b, n in enumerate(lens):
        n_pad = window - n if pads else 0
        k_all = torch.cat([kv[r0:r0 + n, :d], bk[None].expand(n_pad, d)]); v_all = torch.cat([kv[r0:r0 + n, d:], bv[None].expand(n_pad, d)])
        qh = q[r0:r0 + n].view(n, H, hd).transpose(0, 1)
        kh, vh = k_all.view(-1, H, hd).transpose(0, 1), v_all.view(-1, H, hd).transpose(0, 1)
        sc = qh @ kh.transpose(-1, -2) / math.sqrt(hd)
        vis = torch.ones(n, n + n_pad, dtype=torch.bool); vis[:, :n] = torch.tril(torch.ones(n, n, dtype=torch.bool))
        pr = torch.softmax(sc.masked_fill(~vis, float("-inf")), -1)
        mask = torch.stack([torch.from_numpy(_drop_mask(seed, b * H + h, n, n + n_pad, p)) for h in range(H)])
        outs.append(((pr * mask) @ vh).transpose(0, 1).reshape(n, d))
        r0 += n
    ref = torch.cat(outs)
    (ref * gout[:N].double()).sum().backward()
    tol = dict(rtol=3e-4, atol=3e-5)
    torch.testing.assert_close(out[:N].detach().cpu().double(), ref.detach(), **tol)
    torch.testing.assert_close(qd.grad[:N].cpu().double(), q.grad[:N], **tol)
    torch.testing.assert_close(kvd.grad[:N].cpu().double(), kv.grad[:N], **tol)
    if pads:
        # bv's autograd gradient on the reference = the pad keys' share only (the real rows' v come from kv here)
        torch.testing.assert_close(bvd.grad.cpu().double(), bv.grad, rtol=3e-4, atol=3e-4)
        torch.testing.assert_close(bkd.grad.cpu().double(), bk.grad, rtol=3e-4, atol=3e-4)     # = -colsum(dk): the window's total is 0


@pytest.mark.parametrize("loss,keypad", [("sampled_softmax", False), ("softmax", False), ("gBCE", True)])
def test_packed_training_loss_and_gradients_equal_the_padded_batch(loss, keypad):
    """`training_loss_packed` against `training_loss` on the padded batch of the same sessions (dropout 0, same negatives): the loss
    and every parameter gradient."""
    from rectools_amd import nn as hnn
    from rectools_amd import lightning as hl

    torch.manual_seed(3)
    V, L, d, H, n_neg = 200, 48, 64, 2, 5
    item_model = hnn.SumOfEmbeddingsConstructor(V, [hnn.IdEmbeddingsItemNet(d, V, 0.0)])
    backbone = hnn.TransformerTorchBackbone(H, 0.0, item_model, hnn.LearnableInversePositionalEncoding(True, L, d),
                                            hnn.SASRecTransformerLayers(2, d, H, 0.0), hnn.DistanceSimilarityModule(), True, keypad)
    lm = hl.TransformerLossModule(backbone, loss, n_neg).cuda().train()
    for prm in lm.parameters():
        if prm.dim() == 1:
            torch.nn.init.normal_(prm, std=0.3)
    rng = np.random.default_rng(0)
    lens = np.r_[rng.integers(2, 2 * L, 40), 2, L + 1, L + 2, 33, 65]
    offsets = torch.tensor(np.r_[0, np.cumsum(lens)], dtype=torch.int64).cuda()
    items = torch.tensor(rng.integers(1, V, int(lens.sum())), dtype=torch.int64).cuda()
    weights = torch.tensor(rng.random(int(lens.sum())).astype(np.float32) + 0.5).cuda()
    rows = torch.arange(len(lens), dtype=torch.int64).cuda()
    cu, x, y, yw, dist = hnn.pack_train_items(offsets, items, weights, rows, L)
    N = int(x.numel()); tail = (N + 127) // 128 * 128 - N
    pad = lambda t: torch.nn.functional.pad(t, (0, tail))   # noqa: E731
    B = len(lens)
    xp = torch.zeros(B, L, dtype=torch.int64, device="cuda"); yp = torch.zeros_like(xp); wp = torch.zeros(B, L, device="cuda")
    for b in range(B):                                              # the padded batch of the same sessions (sasrec.py:86-104)
        n = int(cu[b + 1] - cu[b])
        xp[b, L - n:], yp[b, L - n:], wp[b, L - n:] = x[cu[b]:cu[b + 1]], y[cu[b]:cu[b + 1]], yw[cu[b]:cu[b + 1]]
    neg_p = torch.tensor(rng.integers(1, V, (B, L, n_neg)), dtype=torch.int64).cuda()
    neg = pad(neg_p[xp != 0].t()).t().contiguous() if loss != "softmax" else None
    padded = {"x": xp, "y": yp, "yw": wp}
    packed = {"x": pad(x), "y": pad(y), "yw": pad(yw), "dist": pad(dist), "cu": cu, "window": L}
    if loss != "softmax":
        padded["negatives"], packed["negatives"] = neg_p, neg
    lp = lm.training_loss(padded); lp.backward()
    g_padded = {k: v.grad.clone() for k, v in lm.named_parameters() if v.grad is not None}
    for v in lm.parameters():
        v.grad = None
    lq = lm.training_loss_packed(packed); lq.backward()
    torch.testing.assert_close(lq.detach(), lp.detach(), rtol=1e-4, atol=1e-6)
    for k, v in lm.named_parameters():
        if k in g_padded:
            got = v.grad if v.grad is not None else torch.zeros_like(v)
            torch.testing.assert_close(got, g_padded[k], rtol=2e-3, atol=2e-5 * (float(g_padded[k].abs().max()) + 1e-12), msg=f"gradient of {k}")


@pytest.mark.parametrize("p,pad_keys", [(0.0, True), (0.2, True), (0.2, False)])
def test_fused_packed_block_equals_the_modular_packed_block(p, pad_keys):
    """`ops.sasrec_layer_packed_train` (one autograd node) against the block built from the individual autograd ops, same dropout
    streams: output, input gradient and every parameter gradient."""
    from rectools_amd import nn as hnn
    from rectools_amd import ops

    torch.manual_seed(11)
    d, H, window = 64, 2, 40
    lens = [40, 1, 17, 33, 8, 25, 39, 2]
    B, N = len(lens), sum(lens)
    Np = (N + 127) // 128 * 128
    cu = torch.tensor(np.r_[0, np.cumsum(lens)], dtype=torch.int64).cuda()
    layer = hnn.SASRecTransformerLayer(d, H, p).cuda().train()
    for prm in layer.parameters():
        if prm.dim() == 1:
            torch.nn.init.normal_(prm, std=0.3)
    x0 = torch.randn(Np, d); x0[N:] = 0
    gout = torch.randn(Np, d); gout[N:] = 0
    res = {}
    for name, fwd in (("fused", layer.forward_packed_train), ("modular", layer.forward_packed_modular)):
        for prm in layer.parameters():
            prm.grad = None
        ops.RNG.seed, ops.RNG.step, ops.RNG._stream = 777, 5, 0                 # both variants draw the same dropout streams
        x = x0.cuda().requires_grad_(True)
        out = fwd(x, cu, B, window, pad_keys)
        out.backward(gout.cuda())
        torch.cuda.synchronize()
        res[name] = (out.detach()[:N].clone(), x.grad[:N].clone(), {k: v.grad.clone() for k, v in layer.named_parameters()})
    if p == 0.0:   # with dropout the two variants consume their streams in different orders: compare only the dropout-free case bitwise-ish
        torch.testing.assert_close(res["fused"][0], res["modular"][0], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(res["fused"][1], res["modular"][1], rtol=1e-4, atol=1e-5)
        for k in res["fused"][2]:
            torch.testing.assert_close(res["fused"][2][k], res["modular"][2][k], rtol=2e-3,
                                       atol=2e-5 * (float(res["modular"][2][k].abs().max()) + 1e-12), msg=f"gradient of {k}")
    else:          # statistics only: same scale of outputs and gradients (the masks differ)
        for a, b in ((res["fused"][0], res["modular"][0]), (res["fused"][1], res["modular"][1])):
            assert torch.isfinite(a).all() and 0.5 < float(a.norm() / b.norm()) < 2.0


@pytest.mark.parametrize("train", [True, False])
def test_collate_packed_kernel_equals_the_tensor_op_restatement(train):
    """`rt_collate_packed` (one launch, row offsets cut on the host) against `nn.pack_train_items` / `pack_last_items` (plain torch
    ops pinned to the reference's collate in tests/test_host_path.py): bit-identical x / y / yw / dist, zero tail."""
    from rectools_amd import nn as hnn
    from rectools_amd import ops

    rng = np.random.default_rng(5)
    L, V = 50, 1000
    lens = np.r_[rng.integers(2, 3 * L, 300), 2, L, L + 1, L + 2, 1 if not train else 2]
    offsets_h = np.r_[0, np.cumsum(lens)].astype(np.int64)
    offsets = torch.tensor(offsets_h).cuda()
    items = torch.tensor(rng.integers(1, V, int(lens.sum())), dtype=torch.int64).cuda()
    weights = torch.tensor(rng.random(int(lens.sum())).astype(np.float32) + 0.5).cuda()
    idx_h = rng.permutation(len(lens))[:200].astype(np.int64)
    idx = torch.tensor(idx_h).cuda()
    n_h = np.clip(offsets_h[idx_h + 1] - offsets_h[idx_h] - (1 if train else 0), 0, L)      # what the host side of the loop computes
    cu_h = np.r_[0, np.cumsum(n_h)].astype(np.int64)
    N = int(cu_h[-1]); rows = (N + 127) // 128 * 128
    cu = torch.tensor(cu_h).cuda()
    if train:
        x, y, yw, dist = ops.collate_packed(offsets, items, weights, idx, cu, rows, train=True)
        cu_r, x_r, y_r, yw_r, dist_r = hnn.pack_train_items(offsets, items, weights, idx, L)
        assert torch.equal(y[:N], y_r) and torch.equal(yw[:N], yw_r) and float(yw[N:].abs().sum()) == 0 and int(y[N:].abs().sum()) == 0
    else:
        x, dist = ops.collate_packed(offsets, items, None, idx, cu, rows, train=False)
        cu_r, x_r, dist_r = hnn.pack_last_items(offsets, items, idx, L)
    assert torch.equal(cu, cu_r) and torch.equal(x[:N], x_r) and torch.equal(dist[:N], dist_r)
    assert int(x[N:].abs().sum()) == 0 and int(dist[N:].abs().sum()) == 0


def test_packed_train_loop_takes_the_steps_of_the_padded_loop(monkeypatch):
    """The product loop with RT_PACKED_TRAIN=1 (host-cut row offsets, rt_collate_packed, fused packed embedding, packed blocks) against
    the padded loop on the same model and data, dropout 0 and a deterministic sampler: equal losses step by step, equal parameters."""
    from rectools_amd.data_preparator import TransformerNegativeSamplerBase
    from rectools_amd.dataset import Dataset
    from rectools_amd.models import SASRecModel

    class RowHashSampler(TransformerNegativeSamplerBase):
        """Negatives as a pure function of the row's input item (so the padded and the packed batch draw the same ones)."""

        def get_negatives(self, batch_dict, lowest_id, highest_id, session_len_limit=None, **kwargs):
            x = batch_dict["x"]
            j = torch.arange(self.n_negatives, device=x.device, dtype=torch.int64)
            return lowest_id + (x[..., None] * 7919 + j * 104729 + 13) % (highest_id - lowest_id)

    rng = np.random.default_rng(1)
    n_users, n_items, n = 150, 90, 5000
    df = pd.DataFrame({"user_id": rng.integers(0, n_users, n), "item_id": rng.integers(0, n_items, n) + 100, "weight": 1.0,
                       "datetime": pd.to_datetime("2022-01-01") + pd.to_timedelta(rng.integers(0, 500_000, n), unit="m")})
    ds = Dataset.construct(df)
    kw = dict(n_factors=64, n_blocks=2, n_heads=2, session_max_len=24, lr=0.005, batch_size=32, dropout_rate=0.0, loss="sampled_softmax",
              n_negatives=6, seed=5, epochs=1, negative_sampler_type=RowHashSampler)
    losses, params = {}, {}
    for packed in ("0", "1"):
        monkeypatch.setenv("RT_PACKED_TRAIN", packed)
        m = SASRecModel(**kw)
        m._build_model_from_dataset(ds)
        loop = m.training_loop()
        assert loop.packed == (packed == "1")
        m.lightning_model.train()
        loop.begin_epoch(0)
        losses[packed] = [float(loop.step()) for _ in range(9)]       # runs over the epoch's end (5 batches) into the next one
        params[packed] = {k: v.detach().clone() for k, v in m.torch_model.state_dict().items()}
    np.testing.assert_allclose(losses["1"], losses["0"], rtol=2e-4)
    d = kw["n_factors"]
    for k, v in params["0"].items():
        a, b = params["1"][k], v
        if k.endswith("in_proj_bias"):
            # the key bias has NO gradient (it shifts every logit of a query alike): the packed path writes exact zeros where the padded
            # one accumulates rounding noise that Adam turns into O(lr) steps — compare the q and v thirds
            a, b = torch.cat([a[:d], a[2 * d:]]), torch.cat([b[:d], b[2 * d:]])
        torch.testing.assert_close(a, b, rtol=5e-3, atol=5e-4, msg=lambda s, k=k: f"{k}: {s}")


@pytest.mark.parametrize("p,pad_keys", [(0.0, True), (0.25, True), (0.25, False)])
def test_native_block_equals_the_python_block(p, pad_keys, monkeypatch):
    """`rt_sasrec_block_packed_fwd / _bwd` (the launch sequence issued by compiled code, csrc/rt_block.hip) against the Python autograd
    node that issues the same kernels one by one, same dropout streams: output, input gradient, every parameter gradient — and the
    inference form against the Python inference block, for every row and for the last rows only."""
    from rectools_amd import nn as hnn
    from rectools_amd import ops

    torch.manual_seed(13)
    d, H, window = 64, 2, 40
    lens = [40, 1, 17, 33, 8, 25, 39, 2, 40, 31]
    B, N = len(lens), sum(lens)
    Np = (N + 127) // 128 * 128
    cu = torch.tensor(np.r_[0, np.cumsum(lens)], dtype=torch.int64).cuda()
    layer = hnn.SASRecTransformerLayer(d, H, p).cuda().train()
    for prm in layer.parameters():
        if prm.dim() == 1:
            torch.nn.init.normal_(prm, std=0.3)
    x0 = torch.randn(Np, d); x0[N:] = 0
    gout = torch.randn(Np, d); gout[N:] = 0
    res = {}
    for name, native in (("native", "1"), ("python", "0")):
        monkeypatch.setenv("RT_NATIVE_BLOCK", native)
        for prm in layer.parameters():
            prm.grad = None
        ops.RNG.seed, ops.RNG.step, ops.RNG._stream = 4242, 9, 0
        x = x0.cuda().requires_grad_(True)
        out = layer.forward_packed_train(x, cu, B, window, pad_keys, rows_real=N)
        out.backward(gout.cuda())
        ops.join_side_streams()
        torch.cuda.synchronize()
        res[name] = (out.detach()[:N].clone(), x.grad[:N].clone(), {k: v.grad.clone() for k, v in layer.named_parameters()})
    torch.testing.assert_close(res["native"][0], res["python"][0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(res["native"][1], res["python"][1], rtol=1e-5, atol=1e-6)
    for k in res["python"][2]:
        torch.testing.assert_close(res["native"][2][k], res["python"][2][k], rtol=1e-4, atol=1e-6 * (float(res["python"][2][k].abs().max()) + 1e-12),
                                   msg=lambda s, k=k: f"gradient of {k}: {s}")
    layer.eval()
    with torch.no_grad():
        inf = {}
        for name, native in (("native", "1"), ("python", "0")):
            monkeypatch.setenv("RT_NATIVE_BLOCK", native)
            full = layer.forward_packed(x0.cuda(), cu, B, window, pad_keys, rows_real=N)
            last = layer.forward_packed(x0.cuda(), cu, B, window, pad_keys, last_rows=cu[1:] - 1, rows_real=N)
            inf[name] = (full[:N].clone(), last.clone())
        torch.testing.assert_close(inf["native"][0], inf["python"][0], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(inf["native"][1], inf["python"][1], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(inf["native"][1], inf["native"][0][(cu[1:] - 1)], rtol=2e-4, atol=2e-5)


def test_native_block_with_weight_planes_equals_the_python_block(monkeypatch):
    """The native block on PRE-SPLIT weight planes (parameters in a FlatAdam flat buffer -> `ops.WeightPlanes`, rt_gemm_wp forward and
    data-gradient products) against the Python node on rt_gemm: output, input gradient and parameter gradients to fp32 rounding; stale
    planes are impossible — a parameter poked between two passes changes the next pass."""
    from rectools_amd import lightning as hl
    from rectools_amd import nn as hnn
    from rectools_amd import ops

    torch.manual_seed(17)
    d, H, window = 128, 2, 64
    lens = [64, 3, 40, 33, 17, 64, 50, 9, 28, 61, 12]
    B, N = len(lens), sum(lens)
    Np = (N + 127) // 128 * 128
    cu = torch.tensor(np.r_[0, np.cumsum(lens)], dtype=torch.int64).cuda()
    stack = hnn.SASRecTransformerLayers(2, d, H, 0.0).cuda().train()
    for prm in stack.parameters():
        if prm.dim() == 1:
            torch.nn.init.normal_(prm, std=0.3)
    hl.FlatAdam(stack, lr=1e-3)                                  # parameters become views of one flat buffer
    x0 = torch.randn(Np, d); x0[N:] = 0
    gout = torch.randn(Np, d); gout[N:] = 0
    res = {}
    for name, native in (("planes", "1"), ("python", "0")):
        monkeypatch.setenv("RT_NATIVE_BLOCK", native)
        for prm in stack.parameters():
            prm.grad = None
        x = x0.cuda().requires_grad_(True)
        out = stack.forward_packed_train(x, cu, B, window, False, rows_real=N)
        out.backward(gout.cuda())
        ops.join_side_streams(); torch.cuda.synchronize()
        res[name] = (out.detach()[:N].clone(), x.grad[:N].clone(), {k: v.grad.clone() for k, v in stack.named_parameters()})
    assert stack._planes_cache[1].ok
    torch.testing.assert_close(res["planes"][0], res["python"][0], rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(res["planes"][1], res["python"][1], rtol=2e-4, atol=2e-6 * float(res["python"][1].abs().max()))
    for k in res["python"][2]:
        torch.testing.assert_close(res["planes"][2][k], res["python"][2][k], rtol=2e-4, atol=2e-6 * (float(res["python"][2][k].abs().max()) + 1e-12),
                                   msg=lambda s, k=k: f"gradient of {k}: {s}")
    monkeypatch.setenv("RT_NATIVE_BLOCK", "1")
    with torch.no_grad():
        a = stack.forward_packed_train(x0.cuda(), cu, B, window, False, rows_real=N)[:N].clone()
        stack.transformer_blocks[0].feed_forward.ff_linear_1.weight.mul_(1.5)        # poke a weight: the next pass re-splits
        b = stack.forward_packed_train(x0.cuda(), cu, B, window, False, rows_real=N)[:N].clone()
        monkeypatch.setenv("RT_WEIGHT_PLANES", "0")
        c = stack.forward_packed_train(x0.cuda(), cu, B, window, False, rows_real=N)[:N].clone()
    assert float((a - b).abs().max()) > 1e-3
    torch.testing.assert_close(b, c, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("H,hd,L,lens", [(2, 64, 200, [200, 1, 57, 64, 130, 199, 33]), (2, 128, 96, [96, 5, 40, 64, 95]),
                                        (1, 32, 300, [300, 2, 129, 65, 256])])
def test_attention_behind_a_shared_pad_prefix_equals_the_padded_window(H, hd, L, lens):
    """`rt_mha_varlen_prefix_*` (K4v3p): sessions packed in front of ONE copy of the window's pad rows against the explicit left-padded
    window of every session (causal mask only — torch_backbone.py:245-260 without a key-padding mask), fp64 autograd: outputs of every
    real row and of the prefix's own rows, d q / d k / d v of every row — the prefix rows' gradients are the sum over the sessions that
    read them.  The tail of the row block rides along as one more session, as in the training loop."""
    from rectools_amd import ops

    torch.manual_seed(L + hd)
    d = H * hd
    B = len(lens)
    N = sum(lens)
    Np = N + L + 96                                           # a non-empty tail behind the prefix (the loop's is below 128 rows)
    qkv = (torch.randn(Np, 3 * d) * 0.5)
    cu_h = np.r_[0, np.cumsum(lens), N + L, Np].astype(np.int64)     # B sessions, the prefix, the tail
    gout = torch.randn(Np, d)
    gout[N + L:] = 0                                          # nothing flows into the tail
    x = qkv.double().requires_grad_(True)
    outs = []
    pre = x[N:N + L]
    causal = torch.tril(torch.ones(L, L, dtype=torch.bool))

    def window_attention(rows):
        q, k, v = (rows[:, c * d:(c + 1) * d].view(L, H, hd).transpose(0, 1) for c in range(3))
        s = (q @ k.transpose(-1, -2)) / hd ** 0.5
        s = s.masked_fill(~causal, float("-inf"))
        return (torch.softmax(s, -1) @ v).transpose(0, 1).reshape(L, d)

    r0 = 0
    for n in lens:
        win = torch.cat([pre[:L - n], x[r0:r0 + n]])          # the session's left-padded window: the shared pad rows, then its own
        outs.append(window_attention(win)[L - n:])
        r0 += n
    outs.append(window_attention(pre))                        # the prefix's own rows: pads see pads
    ref = torch.cat(outs)
    (ref * gout[:N + L].double()).sum().backward()

    xd = qkv.cuda().requires_grad_(True)
    cu = torch.tensor(cu_h).cuda()
    out = ops.mha_varlen_qkv(xd, cu, B + 2, H, L, True, 0.0, True, n_prefixed=B)
    out.backward(gout.cuda())
    torch.testing.assert_close(out[:N + L].detach().cpu().double(), ref.detach(), rtol=3e-4, atol=3e-5)
    g, gr = xd.grad[:N + L].cpu().double(), x.grad[:N + L]
    torch.testing.assert_close(g, gr, rtol=3e-4, atol=3e-4 * float(gr.abs().max()))
    assert float(xd.grad[N + L:].abs().max()) == 0.0          # zero gradients flow into the tail


def test_default_esasrec_packs_behind_a_shared_pad_prefix(monkeypatch):
    """SASRecModel on LiGR blocks WITHOUT a key-padding mask (the reference's default eSASRec): the product loop on packed rows behind the
    shared pad prefix against the padded loop — same model, same data, dropout 0, a deterministic sampler: equal losses step by step
    (over an epoch's end, short last batch included), equal parameters; then recommend() from packed sessions against the padded encoder."""
    from rectools_amd import nn as hnn
    from rectools_amd.data_preparator import TransformerNegativeSamplerBase
    from rectools_amd.dataset import Dataset
    from rectools_amd.models import SASRecModel

    class RowHashSampler(TransformerNegativeSamplerBase):
        def get_negatives(self, batch_dict, lowest_id, highest_id, session_len_limit=None, **kwargs):
            x = batch_dict["x"]
            j = torch.arange(self.n_negatives, device=x.device, dtype=torch.int64)
            return lowest_id + (x[..., None] * 7919 + j * 104729 + 13) % (highest_id - lowest_id)

    rng = np.random.default_rng(3)
    n_users, n_items, n = 150, 90, 5000
    df = pd.DataFrame({"user_id": rng.integers(0, n_users, n), "item_id": rng.integers(0, n_items, n) + 100, "weight": 1.0,
                       "datetime": pd.to_datetime("2022-01-01") + pd.to_timedelta(rng.integers(0, 500_000, n), unit="m")})
    ds = Dataset.construct(df)
    kw = dict(n_factors=128, n_blocks=2, n_heads=2, session_max_len=40, lr=0.005, batch_size=32, dropout_rate=0.0, loss="sampled_softmax",
              n_negatives=6, seed=5, epochs=1, negative_sampler_type=RowHashSampler, transformer_layers_type=hnn.LiGRLayers)
    losses, params, models = {}, {}, {}
    for packed in ("0", "1"):
        monkeypatch.setenv("RT_PACKED_TRAIN", packed)
        m = SASRecModel(**kw)
        m._build_model_from_dataset(ds)
        loop = m.training_loop()
        assert loop.packed == (packed == "1") and loop.prefix == (packed == "1")
        m.lightning_model.train()
        loop.begin_epoch(0)
        losses[packed] = [float(loop.step().detach()) for _ in range(9)]       # runs over the epoch's end (5 batches) into the next one
        params[packed] = {k: v.detach().clone() for k, v in m.torch_model.state_dict().items()}
        models[packed] = m
    np.testing.assert_allclose(losses["1"], losses["0"], rtol=2e-4)
    d = kw["n_factors"]
    for k, v in params["0"].items():
        a, b = params["1"][k], v
        if k.endswith("in_proj_bias"):
            # the key bias shifts every logit of a query alike: its true gradient is zero, what either path accumulates is rounding noise
            # that Adam turns into O(lr) steps — compare the q and v thirds (as the SASRec loop test above does)
            a, b = torch.cat([a[:d], a[2 * d:]]), torch.cat([b[:d], b[2 * d:]])
        torch.testing.assert_close(a, b, rtol=5e-3, atol=5e-4, msg=lambda s, k=k: f"{k}: {s}")
    monkeypatch.delenv("RT_PACKED_TRAIN")
    m = models["1"]
    m.is_fitted = True
    users = ds.user_id_map.external_ids
    fast = m.recommend(users=users, dataset=ds, k=7, filter_viewed=True)
    monkeypatch.setenv("RT_PACKED", "0")
    slow = m.recommend(users=users, dataset=ds, k=7, filter_viewed=True)
    assert fast[["user_id", "item_id", "rank"]].equals(slow[["user_id", "item_id", "rank"]])
    np.testing.assert_allclose(fast["score"].values, slow["score"].values, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("mode", ["rows", "prefix"])
def test_final_ligr_block_on_one_query_row_per_session(mode):
    """`LiGRLayers.forward_last_packed`: the final block with LayerNorm_1 as its only pass over all rows (the last query's attention through
    `rt_mha_varlen_last_x_fwd` — in "prefix" mode with the shared pad prefix's rows as keys in front of the session's own) == every block on
    every row, last rows taken (what it was until round 6)."""
    from rectools_amd import nn as hnn
    from rectools_amd import ops

    torch.manual_seed(4)
    d, H, L, B = 128, 2, 40, 19
    layers = hnn.LiGRLayers(2, d, H, 0.0).cuda().eval()
    for prm in layers.parameters():
        if prm.dim() == 1:
            torch.nn.init.normal_(prm, std=0.3)
    rng = np.random.default_rng(1)
    lens = rng.integers(1, L + 1, B); lens[0], lens[1] = L, 1
    n = int(lens.sum())
    prefix = mode == "prefix"
    cu_h = np.r_[0, np.cumsum(lens)]
    if prefix:
        cu_h = np.r_[cu_h, cu_h[-1] + L]
    cu = torch.tensor(cu_h, dtype=torch.int64).cuda()
    rows_real = n + (L if prefix else 0)
    Np = (rows_real + 127) // 128 * 128
    x = torch.randn(Np, d, device="cuda") * 0.5
    kw = {"n_prefixed": B} if prefix else {}
    with torch.no_grad():
        got = layers.forward_last_packed(x, cu, B, L, not prefix, rows_real=rows_real, causal=True, **kw)
        with ops.active_planes(layers._fresh_planes()):       # the all-rows form
            seqs = x
            for blk in layers.transformer_blocks:
                seqs = blk.forward_packed(seqs, B, L, True, None, None, None, cu, B if prefix else None)
        want = seqs.index_select(0, cu[1:B + 1] - 1)
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-5 * float(want.abs().max()))
