"""GPU parity at the shapes BASELINE.json quotes its numbers on (the bench's own launch geometries), against the oracle.

The per-kernel / golden tests run small shapes; the heavy-row reducers of the sampled loss, the streaming attention
family at L = 512, the d = 512 ring depth of the top-k kernel and the d = 512 LiGR block only engage at the sizes below:
  * C2  SASRec d256 L200, sampled_softmax N=128 over V=26,744 items, Zipf-popular targets (B = 32 sequences)
  * C4  HSTU L=512, hd=64, H=4: relative-bias attention (streaming family) and one full STU training step
  * C5  eSASRec: LiGR (SwiGLU) block at d=512; `rt_topk_score` at d=512 on a 300k-row catalog, filter + whitelist,
        every users-per-pass tile (16 / 32 / 64 / 128)
Tolerances as in the small-shape tests (fp32, different summation orders); top-k ids/order must be IDENTICAL.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from scipy import sparse

from oracle import ranker_oracle
from oracle import transformer_oracle as T
from test_ops_gpu import close, grads_of, rnd
from test_rank_gpu import CASES as RANK_CASES
from test_transformer_gpu import _close, _random_case, build_hip_model

pytestmark = pytest.mark.gpu


def _zipf_ids(shape, V, gen, alpha=1.0):
    """Item ids in [1, V] with Zipf(alpha) popularity over a fixed random permutation (SURVEY.md §8d inputs)."""
    p = 1.0 / torch.arange(1, V + 1, dtype=torch.float64) ** alpha
    perm = torch.randperm(V, generator=gen) + 1
    n = int(np.prod(shape))
    return perm[torch.multinomial(p / p.sum(), n, replacement=True, generator=gen)].reshape(shape)


# ---------------------------------------------------------------------------------------------------------------------
# C5: top-k at d = 512 (ring depth / stage geometry of the 5M x 512 run), every user tile
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("upp", [16, 32, 64, 128])
@pytest.mark.parametrize("distance", ["dot", "cosine"])
def test_topk_d512_300k_filter_whitelist_vs_oracle(distance, upp):
    from rectools_amd.rank import HipRanker

    g = torch.Generator().manual_seed(21)
    n_items, d, n_users = 300_000, 512, 150
    users = torch.randn(n_users, d, generator=g).numpy()
    items = torch.randn(n_items, d, generator=g).numpy()
    rs = np.random.RandomState(5)
    wl = np.sort(rs.choice(n_items, size=n_items - 4321, replace=False))
    filt = sparse.random(n_users, n_items, density=0.0004, format="csr", random_state=rs, data_rvs=lambda n: np.ones(n))
    sids = np.arange(n_users)
    ranker = HipRanker(distance, "cuda", users, items, batch_size=upp)
    gu, gi, gs = ranker.rank(sids, k=10, filter_pairs_csr=filt, sorted_object_whitelist=wl)
    eu, ei, es = ranker_oracle.rank(users, items, sids, k=10, filter_pairs_csr=filt, sorted_object_whitelist=wl, distance=distance)
    np.testing.assert_array_equal(gu, eu)
    np.testing.assert_array_equal(gi, ei)          # identical ids AND order (Gaussian factors: tie-free)
    np.testing.assert_allclose(gs, es, rtol=5e-5, atol=5e-5)


def test_topk_users_per_launch_16_equals_32():
    """The 16-user tile (HBM-bound regime of the 5M x 512 run) against the 32-user tile: identical ids and order; scores
    equal to fp32 rounding (the two MFMA shapes associate the k products differently)."""
    from rectools_amd.rank import HipRanker

    g = torch.Generator(device="cuda").manual_seed(3)
    items = torch.randn(200_000, 512, generator=g, device="cuda")
    users = torch.randn(16, 512, generator=g, device="cuda")
    a = HipRanker("dot", "cuda", users, items, batch_size=16).rank_device(np.arange(16), k=10)
    b = HipRanker("dot", "cuda", users, items, batch_size=32).rank_device(np.arange(16), k=10)
    assert torch.equal(a[0], b[0])
    torch.testing.assert_close(a[1], b[1], rtol=2e-6, atol=2e-5)


def test_reference_kats_need_no_near_tie_swaps():
    """The 52 reference cases must come back in the reference's order without using the near-tie allowance of
    `_assert_same_ranking`, except where the reference's own scores are EXACTLY tied (torch.topk leaves that order open)."""
    from rectools_amd.rank import HipRanker

    swaps = []
    for case in RANK_CASES:
        ranker = HipRanker(case["distance"], "cuda", case["users"], case["items"])
        gu, gi, gs = ranker.rank(case["sids"], k=case["k"], filter_pairs_csr=case["filt"], sorted_object_whitelist=case["wl"])
        ei, es, eu = case["ref_items"], case["ref_scores"], case["ref_subjects"]
        for j in np.nonzero(gi != ei)[0]:
            cand = np.nonzero((eu == eu[j]) & (ei == gi[j]))[0]
            exact_tie = cand.size == 1 and es[cand[0]] == es[j]
            if not exact_tie:
                swaps.append((case["tag"], int(j)))
    assert not swaps, f"order differs from the reference outside exact ties: {swaps[:10]}"


# ---------------------------------------------------------------------------------------------------------------------
# C4: HSTU at L = 512, hd = 64, H = 4 (streaming attention family)
# ---------------------------------------------------------------------------------------------------------------------
def test_hstu_attention_L512_hd64():
    from rectools_amd import ops

    B, L, d, H = 2, 512, 256, 4
    hd = d // H
    g = torch.Generator().manual_seed(L)
    ids = torch.randint(1, 50, (B, L), generator=g); ids[0, : L // 4] = 0
    ts = torch.cumsum(torch.randint(0, 3_000_000, (B, L + 1), generator=g), 1) + 1_300_000_000
    q, k, v = rnd(B * L, d, seed=1) * 0.5, rnd(B * L, d, seed=2) * 0.5, rnd(B * L, d, seed=3)
    tw, pw = rnd(129, seed=4) * 0.5, rnd(2 * L - 1, seed=5) * 0.5
    m = (ids != 0).float()

    def ref_fn(q, k, v, tw, pw):
        rab = T.rel_attn_bias({"x.time_weights": tw, "x.pos_weights": pw}, "x.", {"x": ids, "unix_ts": ts}, L)
        qh, kh, vh = (t.view(B, L, H, hd) for t in (q, k, v))
        a = F.silu(torch.einsum("bnhd,bmhd->bhnm", qh, kh) + rab[:, None]) / L
        a = a * torch.tril(torch.ones(L, L))[None, None] * (m[:, None, :, None] * m[:, None, None, :])
        return torch.einsum("bhnm,bmhd->bnhd", a, vh).reshape(B * L, d)

    ref, gref = grads_of(ref_fn, [q, k, v, tw, pw])
    thr = ops.hstu_time_thresholds().cuda()
    got, ggot = grads_of(lambda q, k, v, tw, pw: ops.hstu_attn(q, k, v, tw, pw, ids.cuda(), ts.cuda(), thr, B, H, L),
                         [t.cuda() for t in (q, k, v, tw, pw)])
    close(got, ref, rtol=5e-4, atol_rel=5e-5, msg="hstu fwd")
    for i, n in enumerate(("dq", "dk", "dv", "dtw", "dpw")):
        close(ggot[i], gref[i], rtol=2e-3, atol_rel=2e-4, msg=f"hstu {n}")


def test_softmax_attention_L512_hd64():
    """The softmax family on the same streaming geometry (SASRec / BERT4Rec with session_max_len = 512)."""
    from rectools_amd import ops

    B, L, d, H = 2, 512, 256, 4
    hd = d // H
    ids = torch.randint(1, 50, (B, L), generator=torch.Generator().manual_seed(7)); ids[1, : L // 3] = 0
    q, k, v = rnd(B * L, d, seed=1), rnd(B * L, d, seed=2), rnd(B * L, d, seed=3)
    mask = T.attention_mask(ids, True, True)

    def ref_fn(q, k, v):
        qh, kh, vh = (t.view(B, L, H, hd).transpose(1, 2) for t in (q, k, v))
        s = qh @ kh.transpose(-1, -2) / math.sqrt(hd) + mask[:, None]
        return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B * L, d)

    ref, gref = grads_of(ref_fn, [q, k, v])
    got, ggot = grads_of(lambda q, k, v: ops.mha(q, k, v, ids.cuda(), B, H, L, True, True, 0.0), [q.cuda(), k.cuda(), v.cuda()])
    close(got, ref, rtol=5e-4, atol_rel=5e-5, msg="mha fwd")
    for i, n in enumerate("qkv"):
        close(ggot[i], gref[i], rtol=2e-3, atol_rel=2e-4, msg=f"mha d{n}")


MEASURED = {}     # test name -> {tensor: max |got - oracle| / max |oracle|}: printed (pytest -s / on failure) and kept for profiles/


def _pack(batch, prefix=False):
    """The padded [B, L] batch as the packed batch `training_loss_packed` takes (real positions only, 128-row tile tail of zeros; the
    row count known on the host selects the native block executors / the fused packed STU node).  prefix: the window's pad rows ride
    along once, as session B behind the real ones (id 0, no target, positions L - 1 .. 0) — `nn.LiGRLayers.packed_mode` "prefix"."""
    x = batch["x"].cuda()
    B, L = x.shape
    real = x != 0
    N = int(real.sum())
    rows_used = N + (L if prefix else 0)
    tail = (rows_used + 127) // 128 * 128 - N
    pad = lambda t: torch.nn.functional.pad(t, (0, tail))   # noqa: E731
    lens = real.sum(1)
    cu = torch.zeros(B + 1, dtype=torch.int64, device="cuda"); cu[1:] = torch.cumsum(lens, 0)
    dist = (L - 1 - torch.arange(L, device="cuda"))[None, :].expand(B, L)
    out = {"x": pad(x[real]), "y": pad(batch["y"].cuda()[real]), "yw": pad(batch["yw"].cuda()[real]), "dist": pad(dist[real]), "cu": cu,
           "window": L, "n_rows": N}
    if prefix:
        out["dist"][N:N + L] = L - 1 - torch.arange(L, device="cuda")
        full = torch.cat([cu, torch.tensor([N + L, N + tail], device="cuda")])
        out.update(cu=full[:B + 2], n_rows=N + L, n_prefixed=B)
        if tail > L:
            out["cu_attn"] = full
    if "negatives" in batch:
        out["negatives"] = pad(batch["negatives"].cuda()[real].t()).t().contiguous()
    if "unix_ts" in batch:
        ts = batch["unix_ts"].cuda()
        out["ts"] = torch.cat([ts[b, L - int(n):] for b, n in enumerate(lens.tolist())])
    return out


def _step_vs_oracle(cfg, batch, grad_rtol=1e-3, name="step", packed=False):
    """One whole training step against the oracle: loss to 5e-5, every parameter gradient to rtol 1e-3 of its own value (entries below
    2e-5 of the tensor's maximum: absolute) — what is measured is ~1e-5 of the tensor scale; the measured maxima are recorded."""
    from rectools_amd import lightning as hl

    torch.manual_seed(100)
    lm = build_hip_model(cfg)
    hl.xavier_normal_init(lm.torch_model)
    with torch.no_grad():
        for _, p in lm.torch_model.named_parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    params = {k: v.detach().cpu().clone() for k, v in lm.torch_model.state_dict().items()}
    loss_ref, g_ref = T.loss_and_grads(cfg, params, batch)
    dbatch = {k: v.cuda() for k, v in batch.items()}
    lm.train()
    lm.zero_grad()
    if packed == "compiled":      # the step as fit() issues it for the stock SASRec configuration: rt_sasrec_step_run (csrc/rt_step.hip)
        opt = hl.FlatAdam(lm.torch_model, lr=cfg["lr"])
        native = hl.NativeSasrecStep.plan(lm, opt)
        assert native is not None, "the stock configuration must take the compiled step"
        pb = _pack(batch)
        assert native.ready(pb)
        loss = native.forward_backward(pb)
        grads = dict(zip((id(p) for p in opt.params), native.gradients()))
        assert abs(float(loss) - float(loss_ref)) <= 5e-5 * abs(float(loss_ref)) + 5e-6, (float(loss), float(loss_ref))
        rec = MEASURED.setdefault(name, {"loss_rel": abs(float(loss) - float(loss_ref)) / abs(float(loss_ref))})
        for n, p in lm.torch_model.named_parameters():
            got = grads[id(p)]
            assert got is not None, n
            rec[n] = float((got - g_ref[n].to(got.device)).abs().max()) / (float(g_ref[n].abs().max()) + 1e-30)
            _close(got, g_ref[n], grad_rtol, 2e-5 if g_ref[n].abs().max() > 1e-6 else 1.0, f"grad {n}")
        print(f"[{name}] loss rel err {rec['loss_rel']:.2e}; max gradient error / tensor scale {max(v for k, v in rec.items() if k != 'loss_rel'):.1e}")
        # ... and its Adam step against torch.optim.Adam's formula on the oracle's gradients (first step: m = (1 - b1) g, v = (1 - b2) g^2)
        before = {n: p.detach().clone() for n, p in lm.torch_model.named_parameters()}
        native.adam()
        torch.cuda.synchronize()
        b1, b2 = opt.betas
        for n, p in lm.torch_model.named_parameters():
            g = g_ref[n].to(p.device).double()
            m_hat, v_hat = g, g * g                                 # (1 - b) g / (1 - b^1)
            want = before[n].double() - cfg["lr"] * m_hat / (v_hat.sqrt() + opt.eps)
            live = g.abs() > 1e-4 * g.abs().max()                   # (where |g| ~ eps the step is ill-conditioned in the gradient's last bits)
            assert float((p.detach().double() - want)[live].abs().max()) <= 2e-2 * cfg["lr"], n
        return
    if packed:      # the padding-free path of the same step, straight against the oracle
        tm = lm.torch_model
        assert tm.transformer_layers.packed_ok(cfg["d"], cfg["L"], tm.use_causal_attn, tm.use_key_padding_mask)
        mode = getattr(tm.transformer_layers, "packed_mode", None)
        prefix = mode is not None and mode(cfg["d"], cfg["L"], tm.use_causal_attn, tm.use_key_padding_mask) == "prefix"
        loss = lm.training_loss_packed(_pack(batch, prefix))
    else:
        loss = lm.training_loss(dbatch)
    loss.backward()
    from rectools_amd import ops

    ops.join_side_streams()
    assert abs(float(loss.detach()) - float(loss_ref)) <= 5e-5 * abs(float(loss_ref)) + 5e-6, (float(loss.detach()), float(loss_ref))
    rec = MEASURED.setdefault(name, {"loss_rel": abs(float(loss.detach()) - float(loss_ref)) / abs(float(loss_ref))})
    for n, p in lm.torch_model.named_parameters():
        ref = g_ref[n].to(p.grad.device)
        rec[n] = float((p.grad - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
    print(f"[{name}] loss rel err {rec['loss_rel']:.2e}; max gradient error / tensor scale: " +
          ", ".join(f"{k.split('.')[-2] if '.' in k else k}.{k.split('.')[-1]}={v:.1e}" for k, v in rec.items() if k != "loss_rel"))
    for n, p in lm.torch_model.named_parameters():
        _close(p.grad, g_ref[n], grad_rtol, 2e-5 if g_ref[n].abs().max() > 1e-6 else 1.0, f"grad {n}")


@pytest.mark.parametrize("packed,B", [(False, 2), (True, 2), (True, 8)])
def test_stu_training_step_L512_d256_H4(packed, B):
    """C4 model shape: HSTU, relative time + position bias, cosine, sampled_softmax, logits_t 0.05 — through the padded window and
    through the packed rows (K6v2 bf16-plane attention, sessions longer than 192 rows in chunks; fused packed STU node), B = 2 and
    B = 8 sequences of random lengths up to 512."""
    cfg, batch = _random_case("stu", "sampled_softmax", "cosine", 512, 256, 4, B, 600, 16, 31, logits_t=0.05)
    _step_vs_oracle(cfg, batch, name=f"C4 STU L512 B{B}" + (" packed" if packed else ""), packed=packed)


@pytest.mark.parametrize("packed", [False, True], ids=["padded", "packed_behind_the_pad_prefix"])
@pytest.mark.parametrize("L,B", [(64, 3), (200, 4)])
def test_ligr_training_step_d512(L, B, packed):
    """C5 model shape: SASRec data path on LiGR blocks (SwiGLU, no FFN bias, multiplier 4) at d = 512, H = 4 (hd = 128) — at a short
    window and at the configuration's own L = 200 — through the padded window and, the reference's DEFAULT (no key-padding mask: the
    pad rows carry state), through packed rows behind ONE copy of the window's pad rows (`nn.LiGRLayers.packed_mode` "prefix"): loss and
    every parameter gradient straight against the oracle's padded window."""
    cfg, batch = _random_case("ligr", "sampled_softmax", "dot", L, 512, 4, B, 700, 16, 32,
                              layer_kwargs=dict(ff_factors_multiplier=4, ff_activation="swiglu", bias_in_ff=False))
    _step_vs_oracle(cfg, batch, name=f"C5 LiGR d512 L{L}" + (" packed (pad prefix)" if packed else ""), packed=packed)


@pytest.mark.parametrize("packed", [False, True])
def test_ligr_training_step_d512_key_padding_mask(packed):
    """The same stack with `use_key_padding_mask=True` — the configuration under which LiGR packs exactly (no real query sees a pad key;
    ligr.py:66-106 is row-wise otherwise) — through the padded window and through the packed rows (every GEMM / LayerNorm / gate on the
    real rows only, the attention on the window they are scattered into), both straight against the oracle."""
    cfg, batch = _random_case("ligr", "sampled_softmax", "dot", 64, 512, 4, 3, 700, 16, 33, keypad=True,
                              layer_kwargs=dict(ff_factors_multiplier=4, ff_activation="swiglu", bias_in_ff=False))
    _step_vs_oracle(cfg, batch, name="C5 LiGR d512 kpm" + (" packed" if packed else ""), packed=packed)


def test_ligr_packing_modes(monkeypatch):
    cfg, _ = _random_case("ligr", "sampled_softmax", "dot", 64, 512, 4, 3, 700, 16, 32,
                          layer_kwargs=dict(ff_factors_multiplier=4, ff_activation="swiglu", bias_in_ff=False))
    layers = build_hip_model(cfg).torch_model.transformer_layers
    assert layers.packed_mode(cfg["d"], cfg["L"], True, True) == "rows" and layers.packed_mode(cfg["d"], cfg["L"], True, False) == "prefix"
    assert layers.packed_mode(cfg["d"], cfg["L"], False, False) is None          # (bidirectional without masks: every row sees every pad)
    monkeypatch.setenv("RT_PACKED_PREFIX", "0")
    assert layers.packed_mode(cfg["d"], cfg["L"], True, False) is None and not layers.packed_ok(cfg["d"], cfg["L"], True, False)


@pytest.mark.parametrize("packed", [False, True])
def test_bert4rec_training_step_C3_shape(packed):
    """BASELINE config 3: BERT4Rec d256, 2 Pre-LN blocks, 4 heads, L200, key-padding masks, mask_prob 0.15, FULL softmax over the
    26,744-item catalog (+ PAD + MASK = 26,746 classes: the padded-to-128 exact-tile GEMMs, the -inf-masked pad columns and the
    split-K dS product of `ops._SoftmaxLoss`), B = 16 sequences: loss and every parameter gradient against the oracle."""
    L, d, H, B, V = 200, 256, 4, 16, 26_744
    cfg, batch = _random_case("preln", "softmax", "dot", L, d, H, B, V, 1, 51, causal=False, keypad=True,
                              layer_kwargs=dict(ff_factors_multiplier=4))
    g = torch.Generator().manual_seed(52)
    x, y = batch["x"].clone(), batch["x"].clone()          # bert4rec.py:109-153: targets are the masked inputs themselves
    real = x != 0
    masked = real & (torch.rand(B, L, generator=g) < 0.15)
    masked[0, -1] = True
    roll = torch.rand(B, L, generator=g)
    x[masked & (roll < 0.8)] = 1                                                      # -> MASK
    rnd_ids = torch.randint(2, V + 2, (B, L), generator=g)
    swap = masked & (roll >= 0.8) & (roll < 0.9)
    x[swap] = rnd_ids[swap]                                                           # -> a random item; the rest stays
    y[~masked] = 0
    batch["x"], batch["y"] = x, y
    batch["yw"] = (y != 0).float() * (0.5 + torch.rand(B, L, generator=g))
    assert 0.10 < float(masked.sum()) / float(real.sum()) < 0.20
    _step_vs_oracle(cfg, batch, name="C3 BERT4Rec" + (" packed" if packed else ""), packed=packed)


@pytest.mark.parametrize("causal,keypad", [(True, False), (False, True), (True, True)])
def test_softmax_attention_hd128_L200(causal, keypad):
    """The C5 model's head size (d 512 / 4 heads = 128) at the C2 / C3 window: forward and the three gradients of the padded attention
    family against plain torch (the hd = 128 kernels were only exercised at L = 64 before)."""
    from rectools_amd import ops

    B, L, d, H = 3, 200, 512, 4
    hd = d // H
    ids = torch.randint(1, 50, (B, L), generator=torch.Generator().manual_seed(17)); ids[1, : L // 3] = 0; ids[2, : L - 5] = 0
    q, k, v = rnd(B * L, d, seed=11), rnd(B * L, d, seed=12), rnd(B * L, d, seed=13)
    mask = T.attention_mask(ids, causal, keypad)

    def ref_fn(q, k, v):
        qh, kh, vh = (t.view(B, L, H, hd).transpose(1, 2) for t in (q, k, v))
        s = qh @ kh.transpose(-1, -2) / math.sqrt(hd) + mask[:, None]
        return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B * L, d)

    ref, gref = grads_of(ref_fn, [q, k, v])
    got, ggot = grads_of(lambda q, k, v: ops.mha(q, k, v, ids.cuda(), B, H, L, causal, keypad, 0.0), [q.cuda(), k.cuda(), v.cuda()])
    close(got, ref, rtol=5e-4, atol_rel=5e-5, msg="mha fwd hd128")
    for i, n in enumerate("qkv"):
        close(ggot[i], gref[i], rtol=2e-3, atol_rel=2e-4, msg=f"mha hd128 d{n}")


# ---------------------------------------------------------------------------------------------------------------------
# C2: the bench's own loss / embedding geometry — V = 26,744, N = 128, Zipf-popular ids (heavy-row reducers engage)
# ---------------------------------------------------------------------------------------------------------------------
def test_sampled_softmax_V26744_N128_zipf_targets():
    from rectools_amd import ops

    M, d, V, N = 32 * 200, 256, 26_745, 128
    g = torch.Generator().manual_seed(1)
    sess, table = rnd(M, d, seed=2) * 0.3, rnd(V, d, seed=3) * 0.3
    y = _zipf_ids((M,), V - 1, g)
    y[torch.rand(M, generator=g) < 0.28] = 0          # left padding share of ML-20M-shaped batches
    neg = torch.randint(1, V, (M, N), generator=g)
    w = (y != 0).float()
    assert int(torch.bincount(y[y > 0]).max()) > 300   # the most popular target owns several reducer chunks

    def ref_fn(sess, table):
        cand = torch.cat([y[:, None], neg], 1)
        lg = torch.einsum("mcd,md->mc", table[cand], sess)[None]
        return T.sampled_softmax_loss(lg, y[None], w[None]).reshape(1)

    ref, gref = grads_of(ref_fn, [sess, table])
    got, ggot = grads_of(lambda s, e: ops.sampled_loss(s, e, y.cuda(), neg.cuda(), w.cuda(), ops.LOSS_SAMPLED_SOFTMAX, False, 1.0, 0.0)[0].reshape(1),
                         [sess.cuda(), table.cuda()])
    close(got, ref, rtol=2e-5, atol_rel=1e-6, msg="loss")
    gref[1][0] = 0
    close(ggot[0], gref[0], rtol=2e-3, atol_rel=2e-4, msg="d_sess")
    close(ggot[1], gref[1], rtol=2e-3, atol_rel=2e-4, msg="d_table")


@pytest.mark.parametrize("loss,dist,kw", [("BCE", "dot", {}), ("gBCE", "dot", {"keypad": True}), ("sampled_softmax", "cosine", {"logits_t": 0.05})],
                         ids=["bce", "gbce_keypad", "cosine_t005"])
def test_compiled_sasrec_step_vs_oracle(loss, dist, kw):
    """`lightning.NativeSasrecStep` (one compiled call per step) against the oracle at a small shape: the three sampled losses, with and
    without key-padding masks, cosine similarity with the C4 temperature — loss, every gradient it leaves in its arena, its Adam step."""
    cfg, batch = _random_case("sasrec", loss, dist, 64, 128, 2, 6, 900, 24, 77, **kw)
    _step_vs_oracle(cfg, batch, name=f"compiled SASRec {loss} {dist}", packed="compiled")


@pytest.mark.parametrize("packed", [False, True, "compiled"])
def test_sasrec_training_step_C2_shape_zipf(packed):
    """One whole C2 training step (d256, 2 blocks, 4 heads, L200, sampled_softmax N=128, V=26,744) on 32 Zipf-popular
    sequences: loss and EVERY parameter gradient (incl. the embedding-table gradient through both heavy-row reducers)."""
    L, d, H, B, V, N = 200, 256, 4, 32, 26_744, 128
    cfg, batch = _random_case("sasrec", "sampled_softmax", "dot", L, d, H, B, V, N, 41)
    g = torch.Generator().manual_seed(42)
    seq = _zipf_ids((B, L + 1), V, g)
    lens = torch.randint(20, L + 1, (B,), generator=g); lens[0] = L
    pad = torch.arange(L)[None, :] < (L - lens)[:, None]
    batch["x"] = seq[:, :-1].masked_fill(pad, 0)
    batch["y"] = seq[:, 1:].masked_fill(pad, 0)
    batch["yw"] = (batch["y"] != 0).float()
    _step_vs_oracle(cfg, batch, name="C2 SASRec" + (f" {packed}" if isinstance(packed, str) else " packed" if packed else ""), packed=packed)
