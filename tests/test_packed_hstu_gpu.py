"""GPU tests of the packed (padding-free) HSTU path: `rt_hstu_attn_varlen_*` against the padded kernels on the
left-padded batch of the same sessions (outputs and every gradient, the relative time / position tables' included),
`rt_collate_packed_ts` against the padded collate's timestamp rows, the packed STU stack against the padded one (loss and parameter
gradients), and HSTUModel's product loop with and without packed batches."""
import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sessions(lens, L, d, seed):
    """Padded [B*L, d] q / k / v (pad rows zero, as the STU block feeds them) + ids + ts [B, L+1], and the packed twins."""
    g = torch.Generator().manual_seed(seed)
    B = len(lens)
    ids = torch.zeros(B, L, dtype=torch.int64)
    ts = torch.cumsum(torch.randint(1, 5_000_000, (B, L + 1), generator=g), dim=1) + 1_400_000_000
    qkv = torch.randn(3, B, L, d, generator=g) * 0.5
    for b, n in enumerate(lens):
        ids[b, L - n:] = 1 + torch.arange(n)
        ts[b, : L - n] = ts[b, L - n]
        qkv[:, b, : L - n] = 0
    real = ids.reshape(-1) != 0
    cu = torch.tensor(np.r_[0, np.cumsum(lens)], dtype=torch.int64)
    ts_packed = torch.cat([ts[b, L - n:] for b, n in enumerate(lens)])          # n + 1 entries per session
    return ids, ts, qkv.reshape(3, B * L, d), real, cu, ts_packed


@pytest.mark.parametrize("impl", ["ring", "v2", "v3"])
@pytest.mark.parametrize("H,hd,L,time,pos", [(2, 32, 96, True, True), (4, 64, 130, True, False), (1, 64, 512, True, True), (2, 64, 70, False, True)])
def test_hstu_varlen_attention_equals_the_padded_kernels(H, hd, L, time, pos, impl, monkeypatch):
    """impl = "v3" (the default): the streamed bf16-plane kernels (K6v3: 64 owner rows per workgroup, 64-row chunks, accumulators in
    registers); impl = "v2": the whole-session workgroups on the same arithmetic (K6v2, RT_HSTU_ATTN=v2; sessions longer than 192 rows walk the
    partner rows in chunks); the padded kernels that make the reference values do not read the switch."""
    from rectools_amd import ops

    monkeypatch.setenv("RT_HSTU_ATTN", impl)

    d = H * hd
    lens = [L, 1, 33, 64, 7, L - 1, 32, max(1, L // 2)]
    B = len(lens)
    ids, ts, qkv, real, cu, ts_packed = _sessions(lens, L, d, seed=H + L)
    N = int(real.sum()); Np = (N + 127) // 128 * 128
    torch.manual_seed(1)
    tw = (torch.randn(129) * 0.3).cuda().requires_grad_(True) if time else None
    pw = (torch.randn(2 * L - 1) * 0.3).cuda().requires_grad_(True) if pos else None
    thr = ops.hstu_time_thresholds().cuda()
    gout = torch.randn(B * L, d); gout[~real] = 0

    qp, kp, vp = (t.cuda().requires_grad_(True) for t in qkv)
    out_p = ops.hstu_attn(qp, kp, vp, tw, pw, ids.cuda(), ts.cuda() if time else None, thr, B, H, L)
    out_p.backward(gout.cuda())
    ref = dict(out=out_p.detach()[real.cuda()], dq=qp.grad[real.cuda()], dk=kp.grad[real.cuda()], dv=vp.grad[real.cuda()],
               dtw=tw.grad.clone() if time else None, dpw=pw.grad.clone() if pos else None)
    if time: tw.grad = None
    if pos: pw.grad = None

    pad = lambda t: torch.nn.functional.pad(t, (0, 0, 0, Np - N))   # noqa: E731
    # the three operands as column blocks of ONE [Np, 3d] buffer (what the STU block hands over: strided rows)
    packed = torch.cat([pad(t[real]) for t in qkv], dim=1).cuda().requires_grad_(True)
    q, k, v = packed[:, :d], packed[:, d:2 * d], packed[:, 2 * d:]
    out = ops.hstu_attn_varlen(q, k, v, tw, pw, cu.cuda(), ts_packed.cuda() if time else None, thr, B, H, L)
    out.backward(pad(gout[real]).cuda())
    tol = dict(rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(out[:N].detach(), ref["out"], **tol)
    assert float(out[N:].detach().abs().max()) == 0.0 if Np > N else True
    g = packed.grad
    torch.testing.assert_close(g[:N, :d], ref["dq"], **tol)
    torch.testing.assert_close(g[:N, d:2 * d], ref["dk"], **tol)
    torch.testing.assert_close(g[:N, 2 * d:], ref["dv"], **tol)
    assert float(g[N:].abs().max()) == 0.0 if Np > N else True
    if time:
        torch.testing.assert_close(tw.grad, ref["dtw"], rtol=1e-3, atol=1e-4)
    if pos:
        torch.testing.assert_close(pw.grad, ref["dpw"], rtol=1e-3, atol=1e-4)


def test_collate_packed_ts_equals_the_padded_timestamp_rows():
    """`rt_collate_packed_ts`: session b's n + 1 timestamps = the columns behind the left pad of `rt_collate` mode 0's [B, L+1] rows."""
    from rectools_amd import ops

    rng = np.random.default_rng(2)
    L = 40
    lens = np.r_[rng.integers(2, 3 * L, 200), 2, L, L + 1, L + 2]
    offsets_h = np.r_[0, np.cumsum(lens)].astype(np.int64)
    offsets = torch.tensor(offsets_h).cuda()
    total = int(lens.sum())
    items = torch.tensor(rng.integers(1, 500, total), dtype=torch.int64).cuda()
    weights = torch.ones(total).cuda()
    unix_ts = torch.tensor(np.cumsum(rng.integers(1, 1000, total)) + 1_400_000_000, dtype=torch.int64).cuda()
    idx_h = rng.permutation(len(lens))[:150].astype(np.int64)
    idx = torch.tensor(idx_h).cuda()
    B = len(idx_h)
    n_h = np.clip(offsets_h[idx_h + 1] - offsets_h[idx_h] - 1, 0, L)
    cu_h = np.r_[0, np.cumsum(n_h)].astype(np.int64)
    cu = torch.tensor(cu_h).cuda()
    N = int(cu_h[-1])
    got = ops.collate_packed_ts(offsets, unix_ts, idx, cu, N)
    x = torch.empty(B, L, dtype=torch.int64, device="cuda"); y = torch.empty_like(x); yw = torch.empty(B, L, device="cuda")
    ts = torch.empty(B, L + 1, dtype=torch.int64, device="cuda")
    ops._c("rt_collate", offsets, items, weights, unix_ts, idx, B, L, 0, None, None, 0.0, 0, x, y, yw, ts)
    want = torch.cat([ts[b, L - int(n):] for b, n in enumerate(n_h)])
    assert got.shape == want.shape and torch.equal(got, want)


@pytest.mark.parametrize("loss,time", [("sampled_softmax", True), ("softmax", False)])
def test_packed_hstu_loss_and_gradients_equal_the_padded_batch(loss, time):
    from rectools_amd import lightning as hl
    from rectools_amd import nn as hnn

    torch.manual_seed(6)
    V, L, d, H, n_neg = 200, 48, 64, 2, 5
    item_model = hnn.SumOfEmbeddingsConstructor(V, [hnn.IdEmbeddingsItemNet(d, V, 0.0)])
    layers = hnn.STULayers(2, d, H, d // H, d // H, L, time, True, 0.0, 0.0)
    backbone = hnn.TransformerTorchBackbone(H, 0.0, item_model, hnn.LearnableInversePositionalEncoding(True, L, d, use_scale_factor=True),
                                            layers, hnn.DistanceSimilarityModule("cosine"), True, False)
    lm = hl.TransformerLossModule(backbone, loss, n_neg, logits_t=0.1).cuda().train()
    for prm in lm.parameters():
        if prm.dim() == 1:
            torch.nn.init.normal_(prm, std=0.3)
    assert layers.packed_ok(d, L, True, False)
    rng = np.random.default_rng(0)
    lens = np.r_[rng.integers(1, L + 1, 30), 1, L, L - 1, 33, 32]
    B = len(lens)
    g = torch.Generator().manual_seed(3)
    xp = torch.zeros(B, L, dtype=torch.int64); yp = torch.zeros_like(xp); wp = torch.zeros(B, L)
    ts = torch.cumsum(torch.randint(1, 5_000_000, (B, L + 1), generator=g), dim=1) + 1_400_000_000
    for b, n in enumerate(lens):
        xp[b, L - n:] = torch.tensor(rng.integers(1, V, n)); yp[b, L - n:] = torch.tensor(rng.integers(1, V, n))
        wp[b, L - n:] = torch.tensor(rng.random(n).astype(np.float32) + 0.5)
        ts[b, : L - n] = ts[b, L - n]
    xp, yp, wp, ts = xp.cuda(), yp.cuda(), wp.cuda(), ts.cuda()
    real = xp != 0
    N = int(real.sum()); tail = (N + 127) // 128 * 128 - N
    pad = lambda t: torch.nn.functional.pad(t, (0, tail))   # noqa: E731
    cu = torch.tensor(np.r_[0, np.cumsum(lens)], dtype=torch.int64).cuda()
    dist_p = (L - 1 - torch.arange(L, device="cuda"))[None, :].expand(B, L)
    neg_p = torch.tensor(rng.integers(1, V, (B, L, n_neg)), dtype=torch.int64).cuda()
    padded = {"x": xp, "y": yp, "yw": wp, "unix_ts": ts}
    packed = {"x": pad(xp[real]), "y": pad(yp[real]), "yw": pad(wp[real]), "dist": pad(dist_p[real]), "cu": cu, "window": L,
              "ts": torch.cat([ts[b, L - int(n):] for b, n in enumerate(lens)])}
    if loss != "softmax":
        padded["negatives"], packed["negatives"] = neg_p, pad(neg_p[real].t()).t().contiguous()
    lp = lm.training_loss(padded); lp.backward()
    g_padded = {k: v.grad.clone() for k, v in lm.named_parameters() if v.grad is not None}
    for pb in (packed, dict(packed, n_rows=N)):        # the individual ops, and (row count known on the host) the fused packed node
        for v in lm.parameters():
            v.grad = None
        lq = lm.training_loss_packed(pb); lq.backward()
        torch.testing.assert_close(lq.detach(), lp.detach(), rtol=1e-4, atol=1e-6)
        for k, v in lm.named_parameters():
            if k in g_padded:
                got = v.grad if v.grad is not None else torch.zeros_like(v)
                torch.testing.assert_close(got, g_padded[k], rtol=2e-3, atol=2e-5 * (float(g_padded[k].abs().max()) + 1e-12),
                                           msg=lambda s, k=k: f"gradient of {k}: {s}")


def test_packed_hstu_train_loop_takes_the_steps_of_the_padded_loop(monkeypatch):
    from rectools_amd.data_preparator import TransformerNegativeSamplerBase
    from rectools_amd.dataset import Dataset
    from rectools_amd.models import HSTUModel

    class RowHashSampler(TransformerNegativeSamplerBase):
        def get_negatives(self, batch_dict, lowest_id, highest_id, session_len_limit=None, **kwargs):
            x = batch_dict["x"]
            j = torch.arange(self.n_negatives, device=x.device, dtype=torch.int64)
            return lowest_id + (x[..., None] * 7919 + j * 104729 + 13) % (highest_id - lowest_id)

    rng = np.random.default_rng(1)
    n_users, n_items, n = 150, 90, 5000
    df = pd.DataFrame({"user_id": rng.integers(0, n_users, n), "item_id": rng.integers(0, n_items, n) + 100, "weight": 1.0,
                       "datetime": pd.to_datetime("2022-01-01") + pd.to_timedelta(rng.integers(0, 500_000, n), unit="m")})
    ds = Dataset.construct(df)
    kw = dict(n_factors=64, n_blocks=2, n_heads=2, session_max_len=40, lr=0.005, batch_size=32, dropout_rate=0.0, loss="sampled_softmax",
              n_negatives=6, seed=5, epochs=1, negative_sampler_type=RowHashSampler)
    losses, params = {}, {}
    for packed in ("0", "1"):
        monkeypatch.setenv("RT_PACKED_TRAIN", packed)
        m = HSTUModel(**kw)
        m._build_model_from_dataset(ds)
        loop = m.training_loop()
        assert loop.packed == (packed == "1")
        m.lightning_model.train()
        loop.begin_epoch(0)
        losses[packed] = [float(loop.step().detach()) for _ in range(9)]
        params[packed] = {k: v.detach().clone() for k, v in m.torch_model.state_dict().items()}
    np.testing.assert_allclose(losses["1"], losses["0"], rtol=3e-4)
    for k, v in params["0"].items():
        torch.testing.assert_close(params["1"][k], v, rtol=5e-3, atol=5e-4, msg=lambda s, k=k: f"{k}: {s}")


def test_hstu_recommend_with_context_on_the_device_path_equals_the_reference_shaped_path():
    """HSTUModel.recommend(context=...) through the device glue (session index with timestamps, packed STU encoder, the request's time
    as the last timestamp of every packed session) against the pandas / numpy path that mirrors the reference line by line: same users,
    items and ranks, scores to fp32 rounding; a context that lies BEFORE a user's last interaction is not the session's last row, and
    the call takes the reference-shaped path for it."""
    import warnings

    from rectools_amd.dataset import Dataset
    from rectools_amd.models import HSTUModel

    rng = np.random.default_rng(0)
    n_users, n_items, n = 200, 120, 5000
    df = pd.DataFrame({"user_id": rng.integers(0, n_users, n) * 3 + 7, "item_id": rng.integers(0, n_items, n) + 1000, "weight": 1.0,
                       "datetime": pd.to_datetime("2022-01-01") + pd.to_timedelta(rng.integers(0, 50_000, n), unit="m")})
    train = Dataset.construct(df[df["item_id"] < 1000 + 90])
    full = Dataset.construct(df)
    model = HSTUModel(n_factors=64, n_blocks=2, n_heads=2, session_max_len=16, lr=0.01, batch_size=64, epochs=1, loss="sampled_softmax",
                      n_negatives=4, seed=1).fit(train)
    users = rng.permutation(full.user_id_map.external_ids)[:120]
    when = pd.to_datetime("2022-03-01") + pd.to_timedelta(rng.integers(0, 10_000, len(users)), unit="m")
    context = pd.DataFrame({"user_id": users, "datetime": when})
    calls = {"fast": 0}
    orig = model._recommend_device_glue

    def counting(*a, **k):
        out = orig(*a, **k)
        calls["fast"] += out is not None
        return out

    for kw in (dict(k=5, filter_viewed=True), dict(k=3, filter_viewed=False, items_to_recommend=np.arange(1000, 1040))):
        model._recommend_device_glue = counting
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fast = model.recommend(users=users, dataset=full, context=context, **kw)
            model._recommend_device_glue = lambda *a, **k: None
            slow = model.recommend(users=users, dataset=full, context=context, **kw)
        model._recommend_device_glue = orig
        assert fast[["user_id", "item_id", "rank"]].equals(slow[["user_id", "item_id", "rank"]])
        np.testing.assert_allclose(fast["score"].values, slow["score"].values, rtol=2e-5, atol=2e-6)
    assert calls["fast"] == 2
    early = context.copy()
    early.loc[0, "datetime"] = pd.to_datetime("2021-06-01")                 # before that user's history: the glue declines
    model._recommend_device_glue = counting
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = model.recommend(users=users, dataset=full, context=early, k=4, filter_viewed=True)
    model._recommend_device_glue = orig
    assert calls["fast"] == 2 and len(a) > 0
    with pytest.raises(ValueError):
        model.recommend(users=users, dataset=full, context=context.iloc[1:], k=4, filter_viewed=True)      # no context for a target user


@pytest.mark.parametrize("num_buckets", [128, 24])
def test_hstu_varlen_chunked_sessions_against_the_oracle(num_buckets):
    """K6v2 straight against the ORACLE (oracle/transformer_oracle.py: hstu.py:84-153,257-288 restated), not against the padded kernels:
    11 sessions, nine of them longer than the 192 partner rows one chunk holds (193 .. 512 rows: two and three chunks, chunk edges at
    and next to multiples of 32), time + position bias, outputs and every gradient incl. the two bias tables'.  num_buckets = 24: the
    weight vector is shorter than the buckets the timestamps reach — the clamp of hstu.py:84-86 through the threshold table's trailer."""
    import torch.nn.functional as F

    from oracle import transformer_oracle as T
    from rectools_amd import ops

    H, hd, L = 4, 64, 512
    d = H * hd
    lens = [512, 193, 300, 257, 448, 200, 385, 511, 194, 64, 5]
    B = len(lens)
    ids, ts, qkv, real, cu, ts_packed = _sessions(lens, L, d, seed=77)
    N = int(real.sum()); Np = (N + 127) // 128 * 128
    g = torch.Generator().manual_seed(5)
    tw0, pw0 = torch.randn(num_buckets + 1, generator=g) * 0.3, torch.randn(2 * L - 1, generator=g) * 0.3
    gout = torch.randn(B * L, d, generator=g); gout[~real] = 0
    m = (ids != 0).float()

    def ref_fn(q, k, v, tw, pw):
        rab = T.rel_attn_bias({"x.time_weights": tw, "x.pos_weights": pw}, "x.", {"x": ids, "unix_ts": ts}, L)
        qh, kh, vh = (t.view(B, L, H, hd) for t in (q, k, v))
        a = F.silu(torch.einsum("bnhd,bmhd->bhnm", qh, kh) + rab[:, None]) / L
        a = a * torch.tril(torch.ones(L, L))[None, None] * (m[:, None, :, None] * m[:, None, None, :])
        return torch.einsum("bhnm,bmhd->bnhd", a, vh).reshape(B * L, d)

    ins = [t.clone().requires_grad_(True) for t in (qkv[0], qkv[1], qkv[2], tw0, pw0)]
    ref = ref_fn(*ins)
    ref.backward(gout)
    pad = lambda t: F.pad(t, (0, 0, 0, Np - N))   # noqa: E731
    packed = torch.cat([pad(t[real]) for t in qkv], dim=1).cuda().requires_grad_(True)
    tw, pw = tw0.cuda().requires_grad_(True), pw0.cuda().requires_grad_(True)
    thr = ops.hstu_time_thresholds(num_buckets).cuda()
    out = ops.hstu_attn_varlen(packed[:, :d], packed[:, d:2 * d], packed[:, 2 * d:], tw, pw, cu.cuda(), ts_packed.cuda(), thr, B, H, L)
    out.backward(pad(gout[real]).cuda())

    def close(got, want, name, rtol=5e-4, atol_rel=5e-5):
        torch.testing.assert_close(got.cpu(), want, rtol=rtol, atol=atol_rel * float(want.abs().max()) + 1e-9, msg=lambda s: f"{name}: {s}")

    close(out[:N].detach(), ref.detach()[real], "out")
    for i, name in enumerate(("dq", "dk", "dv")):
        close(packed.grad[:N, i * d:(i + 1) * d], ins[i].grad[real], name, rtol=2e-3, atol_rel=2e-4)
    close(tw.grad, ins[3].grad, "d time_weights", rtol=2e-3, atol_rel=2e-4)
    close(pw.grad, ins[4].grad, "d pos_weights", rtol=2e-3, atol_rel=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("time_bias,pos_bias", [(True, True), (False, True), (True, False), (False, False)])
def test_final_stu_block_on_one_query_row_per_session(time_bias, pos_bias):
    """`STULayers.forward_last_packed`: the final block with u / q, the attention (`rt_hstu_attn_varlen_last_fwd`), LayerNorm(attn), the gate
    and the output MLP on ONE row per session == every block on every row, last rows taken (what it was until round 6)."""
    from rectools_amd import nn as hnn

    torch.manual_seed(3)
    d, H, hd, L, B = 64, 2, 32, 50, 23
    layers = hnn.STULayers(2, d, H, hd, hd, L, time_bias, pos_bias, 0.0, 0.0, 1e-6).cuda().eval()
    for prm in layers.parameters():
        if prm.dim() == 1:
            torch.nn.init.normal_(prm, std=0.3)
    rng = np.random.default_rng(0)
    lens = rng.integers(1, L + 1, B); lens[0], lens[1] = L, 1
    cu = torch.tensor(np.r_[0, np.cumsum(lens)], dtype=torch.int64).cuda()
    n = int(cu[-1]); Np = (n + 127) // 128 * 128
    x = torch.randn(Np, d, device="cuda") * 0.5
    ts = None
    if time_bias:      # n_b + 1 ascending stamps per session (the last one: the request's time), packed at cu[b] + b
        parts = [np.sort(rng.integers(0, 10_000_000, int(m) + 1)) for m in lens]
        ts = torch.tensor(np.concatenate(parts), dtype=torch.int64).cuda()
    with torch.no_grad():
        want = layers.forward_packed_train(x, cu, B, L, False, ts=ts).index_select(0, cu[1:B + 1] - 1)
        got = layers.forward_last_packed(x, cu, B, L, False, ts=ts)
    torch.testing.assert_close(got, want, rtol=2e-4, atol=2e-5 * float(want.abs().max()))
