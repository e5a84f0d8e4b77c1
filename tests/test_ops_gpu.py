"""Per-kernel GPU parity: every `rt_*` training kernel (through rectools_amd.ops / the C ABI) against the plain
torch fp32 restatement of the same op on CPU (functions of oracle/transformer_oracle.py where they exist).
fp32 tolerances stated per test."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import transformer_oracle as T

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def grads_of(fn, inputs):
    ins = [t.detach().clone().requires_grad_(True) for t in inputs]
    out = fn(*ins)
    g = rnd(*out.shape, seed=99).to(out.device)
    out.backward(g)
    return out.detach(), [t.grad for t in ins]


def close(a, b, rtol=2e-4, atol_rel=2e-5, msg=""):
    b = b.to(a.device)
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol_rel * (float(b.abs().max()) + 1e-12), msg=lambda m: f"{msg}: {m}")


# the last three shapes give exact 128-tile grids: LDS-DMA GEMM path, direct and split-K, bias gradient fused in wgrad
# ... and the C2 M-products at their real size (M = 25,600, N = 256 / 512)
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (300, 77, 40), (1000, 256, 256), (33, 51, 16), (130, 520, 132),
                                   (256, 128, 128), (2048, 256, 128), (4096, 128, 384), (25600, 256, 256), (25600, 512, 256)])
def test_linear_fwd_bwd(M, N, K):
    from rectools_amd import ops

    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.2), rnd(N, seed=3), rnd(M, N, seed=4)
    for relu, res in ((False, False), (True, False), (False, True)):
        if relu and N % 4:  # element-wise kernels stream float4 groups: feature dims are multiples of 4 (ABI)
            continue
        if relu and M * N > 2_000_000:
            # 6.5 M pre-activations always hold a few within fp32 rounding of zero, where relu'(z) legitimately differs between
            # two summation orders (253 of 6,553,600 dx entries at M = 25,600): the relu epilogue is covered by the smaller shapes
            continue
        ref, gref = grads_of(lambda x, w, b, r: (F.relu(x @ w.T + b) if relu else x @ w.T + b + (r if res else 0)), [x, w, b, r])
        got, ggot = grads_of(lambda x, w, b, r: ops.linear(x, w, b, r if res else None, relu), [t.cuda() for t in (x, w, b, r)])
        close(got, ref, msg=f"linear fwd relu={relu} res={res}")
        for i, name in enumerate(("dx", "dw", "db")):
            close(ggot[i], gref[i], rtol=1e-3, atol_rel=1e-4, msg=f"linear {name} relu={relu} res={res}")
        if res:
            close(ggot[3], gref[3], msg="linear dres")


def test_linear_strided_views_and_matmul_nn():
    from rectools_amd import ops

    buf = rnd(200, 96, seed=5)
    w = rnd(48, 32, seed=6, scale=0.3)
    ref, gref = grads_of(lambda bf, w: bf[:, 32:64] @ w.T, [buf, w])
    got, ggot = grads_of(lambda bf, w: ops.linear(bf[:, 32:64], w), [buf.cuda(), w.cuda()])
    close(got, ref, msg="strided fwd"); close(ggot[0], gref[0], rtol=1e-3, msg="strided dx"); close(ggot[1], gref[1], rtol=1e-3, msg="strided dw")
    x, p = rnd(150, 24, seed=7), rnd(24, 100, seed=8, scale=0.3)
    ref, gref = grads_of(lambda x, p: x @ p, [x, p])
    got, ggot = grads_of(lambda x, p: ops.matmul_nn(x, p), [x.cuda(), p.cuda()])
    close(got, ref, msg="nn fwd"); close(ggot[0], gref[0], rtol=1e-3, msg="nn dx"); close(ggot[1], gref[1], rtol=1e-3, msg="nn dp")


@pytest.mark.parametrize("M,d,eps", [(37, 16, 1e-5), (500, 256, 1e-8), (129, 512, 1e-6)])
def test_layernorm(M, d, eps):
    from rectools_amd import ops

    x, w, b = rnd(M, d, seed=1) * 2 + 0.3, rnd(d, seed=2) * 0.1 + 1, rnd(d, seed=3) * 0.1
    x[3] = 0  # a fully masked (all-zero) row, as SASRec feeds after `seqs *= timeline_mask`
    ref, gref = grads_of(lambda x, w, b: T.layer_norm(x, w, b, eps), [x, w, b])
    got, ggot = grads_of(lambda x, w, b: ops.layer_norm(x, w, b, eps), [x.cuda(), w.cuda(), b.cuda()])
    close(got, ref, msg="ln fwd")
    sane = torch.ones(M, dtype=torch.bool); sane[3] = eps >= 1e-6  # rstd of a zero row is 1e4 at eps=1e-8: dx there is noise-scaled
    close(ggot[0][sane.cuda()], gref[0][sane], rtol=2e-3, atol_rel=2e-4, msg="ln dx")
    close(ggot[1], gref[1], rtol=2e-3, atol_rel=2e-4, msg="ln dw"); close(ggot[2], gref[2], rtol=2e-3, msg="ln db")


def test_embed_and_masks():
    from rectools_amd import ops

    V, L, B, d = 60, 10, 5, 32
    table, pos = rnd(V, d, seed=1), rnd(L, d, seed=2)
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, V, (B, L), generator=g); ids[:, :3] = 0
    for scale in (1.0, math.sqrt(d)):
        ref, gref = grads_of(lambda t, p: T.embed_sessions({T.ITEM_EMB: t, T.POS_EMB: p}, ids, scale != 1.0).reshape(B * L, d), [table, pos])
        got, ggot = grads_of(lambda t, p: ops.embed(t, p, ids.cuda(), L, scale, 0.0), [table.cuda(), pos.cuda()])
        close(got, ref, msg="embed fwd")
        gref[0][0] = 0  # nn.Embedding(padding_idx=0): PAD row gets no gradient (item_net.py:260-264)
        close(ggot[0], gref[0], rtol=1e-3, msg="embed dtable"); close(ggot[1], gref[1], rtol=1e-3, msg="embed dpos")
    a, b = rnd(B * L, d, seed=4), rnd(B * L, d, seed=5)
    m = (ids.reshape(-1, 1) != 0).float()
    ref, gref = grads_of(lambda a, b: a * b * m, [a, b])
    got, ggot = grads_of(lambda a, b: ops.mul_mask(a, b, ids.cuda()), [a.cuda(), b.cuda()])
    close(got, ref, msg="mulmask"); close(ggot[0], gref[0], msg="mulmask da"); close(ggot[1], gref[1], msg="mulmask db")


@pytest.mark.gpu
@pytest.mark.parametrize("V,L,B,d,hot", [(300, 50, 40, 512, True), (5000, 20, 8, 64, False), (7, 3, 1, 4, False)])
def test_embed_backward_sorted_reduction(V, L, B, d, hot):
    """Counting-sort embedding backward: popularity skew (workgroup-per-row path), sparse catalog, tiny shapes."""
    from rectools_amd import ops

    table, pos = rnd(V, d, seed=1), rnd(L, d, seed=2)
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, V, (B, L), generator=g)
    if hot:
        ids[torch.rand(B, L, generator=g) < 0.6] = 5
    ref, gref = grads_of(lambda t, p: T.embed_sessions({T.ITEM_EMB: t, T.POS_EMB: p}, ids, False).reshape(B * L, d), [table, pos])
    got, ggot = grads_of(lambda t, p: ops.embed(t, p, ids.cuda(), L, 1.0, 0.0), [table.cuda(), pos.cuda()])
    gref[0][0] = 0
    close(got, ref, msg="embed fwd")
    close(ggot[0], gref[0], rtol=1e-3, msg="embed dtable"); close(ggot[1], gref[1], rtol=1e-3, msg="embed dpos")


def _bag_case(V, F_, max_per_item, seed, popular):
    """Item -> category-value CSR: empty rows, ascending ids per row; `popular` values tag most items (several backward
    chunks each); the last value is carried by no item (zero gradient row)."""
    g = torch.Generator().manual_seed(seed)
    rows = []
    for i in range(V):
        n = int(torch.randint(0, max_per_item + 1, (1,), generator=g))
        vals = set(torch.randint(0, max(F_ - 1, 1), (n,), generator=g).tolist()) if F_ > 1 else set()
        for pv in popular:
            if float(torch.rand(1, generator=g)) < 0.8:
                vals.add(pv)
        rows.append(sorted(vals) if i > 0 else [])      # row 0: PAD, no features
    lens = torch.tensor([len(r) for r in rows], dtype=torch.int64)
    return torch.tensor([v for r in rows for v in r], dtype=torch.int64), torch.cumsum(lens, 0) - lens, lens


@pytest.mark.gpu
@pytest.mark.parametrize("V,F_,d,mx,popular,with_ids", [(700, 23, 256, 5, (0, 3), True), (50, 6, 16, 2, (), True),
                                                        (1200, 9, 512, 3, (1,), False), (5, 1, 4, 0, (), True)])
def test_item_table_bag_sum(V, F_, d, mx, popular, with_ids):
    """K1b: ids_emb + EmbeddingBag-sum of the category embeddings (item_net.py:101-132,463-482) and its backward over the
    chunked transposed structure, against the oracle's plain restatement."""
    from rectools_amd import ops

    inputs, offsets, lens = _bag_case(V, F_, mx, 7, popular)
    ids_w, cat_w = rnd(V, d, seed=1), rnd(F_, d, seed=2)
    P = T.CAT_BLOCK

    def ref_fn(iw, cw):
        p = {T.ITEM_EMB: iw if with_ids else torch.zeros(V, d), T.CAT_EMB: cw, P + "emb_bag_inputs": inputs,
             P + "offsets": offsets, P + "input_lengths": lens}
        return T.item_table(p)

    ref, gref = grads_of(ref_fn, [ids_w, cat_w])
    bag = ops.BagStructure(inputs.cuda(), offsets.cuda(), lens.cuda(), F_)
    if popular:
        assert bag.n_chunks > F_       # popular values are cut into several chunks
    got, ggot = grads_of(lambda iw, cw: ops.item_table(iw if with_ids else None, cw, bag, 0.0), [ids_w.cuda(), cat_w.cuda()])
    close(got, ref, msg="item table")
    close(ggot[1], gref[1], rtol=1e-3, msg="d cat_emb")
    if with_ids:
        close(ggot[0], gref[0], msg="d ids_emb")
    else:
        assert ggot[0] is None
    if F_ > 1:
        assert float(ggot[1][F_ - 1].abs().max()) == 0.0     # value carried by no item


@pytest.mark.gpu
def test_item_table_dropout_mask_is_regenerated_in_backward():
    """Dropout acts on the bag sums only (item_net.py:117-119); backward must re-create the forward's mask."""
    from rectools_amd import ops

    V, F_, d, p = 300, 5, 64, 0.4
    inputs, offsets, lens = _bag_case(V, F_, 3, 11, (0,))
    bag = ops.BagStructure(inputs.cuda(), offsets.cuda(), lens.cuda(), F_)
    ids_w = rnd(V, d, seed=1).cuda().requires_grad_(True)
    cat_w = (rnd(F_, d, seed=2).abs() + 0.5).cuda().requires_grad_(True)     # positive: a zero bag sum means "dropped"
    ops.RNG.step, ops.RNG._stream = 3, 0
    out = ops.item_table(ids_w, cat_w, bag, p)
    with torch.no_grad():
        clean = ops.item_table(ids_w, cat_w, bag, 0.0) - ids_w
        bagpart = out - ids_w
        has = (lens > 0).cuda()[:, None].expand(V, d)
        kept = (bagpart.abs() > 1e-6) & has
        frac = float(kept[has].float().mean())
        assert abs(frac - (1 - p)) < 0.03, frac
        close(bagpart[kept], (clean / (1 - p))[kept], msg="kept entries are scaled by 1/(1-p)")
    g = rnd(V, d, seed=5).cuda()
    out.backward(g)
    close(ids_w.grad, g, msg="id embeddings see the whole gradient")
    # expected d cat: scatter of the masked, scaled gradient rows over the structure
    gm = (g * kept.float() / (1 - p)).cpu()
    exp = torch.zeros(F_, d)
    item_of = torch.repeat_interleave(torch.arange(V), lens)
    exp.index_add_(0, inputs, gm[item_of])
    close(cat_w.grad, exp, rtol=1e-3, msg="d cat_emb under dropout")


@pytest.mark.parametrize("kind,fn", [(1, F.relu), (2, F.gelu), (3, F.silu), (4, torch.sigmoid)])
def test_activations_swiglu_gate(kind, fn):
    from rectools_amd import ops

    z = rnd(64, 48, seed=kind) * 2
    ref, gref = grads_of(lambda z: fn(z), [z])
    got, ggot = grads_of(lambda z: ops.act_dropout(z, kind, 0.0), [z.cuda()])
    close(got, ref, msg="act fwd"); close(ggot[0], gref[0], rtol=1e-3, msg="act bwd")
    a, b, x = rnd(64, 48, seed=7), rnd(64, 48, seed=8), rnd(64, 48, seed=9)
    ref, gref = grads_of(lambda a, b: F.silu(a) * b, [a, b])
    got, ggot = grads_of(lambda a, b: ops.swiglu(a, b, 0.0), [a.cuda(), b.cuda()])
    close(got, ref, msg="swiglu"); close(ggot[0], gref[0], rtol=1e-3, msg="swiglu da"); close(ggot[1], gref[1], rtol=1e-3, msg="swiglu db")
    ref, gref = grads_of(lambda x, g, a: x + torch.sigmoid(g) * a, [x, a, b])
    got, ggot = grads_of(lambda x, g, a: ops.gate(x, g, a, 0.0), [x.cuda(), a.cuda(), b.cuda()])
    close(got, ref, msg="gate")
    for i in range(3):
        close(ggot[i], gref[i], rtol=1e-3, msg=f"gate grad {i}")


# L_pad * head_dim decides the kernel family: LDS-resident K,V (first 8 cases; 288/300 need two schedule rounds) or the
# streaming kernels (last 3 cases); RT_ATTN_IMPL=stream forces streaming for every case (second pytest pass on the GPU box)
@pytest.mark.parametrize("L,d,H,causal,keypad", [(8, 16, 2, True, False), (40, 64, 2, True, True), (70, 128, 4, False, True),
                                                   (200, 256, 4, True, False), (33, 64, 1, True, False), (288, 64, 1, True, True),
                                                   (300, 64, 2, False, True), (100, 128, 1, True, False),
                                                   (320, 128, 2, True, False), (600, 64, 2, False, True), (200, 256, 2, True, True)])
def test_mha(L, d, H, causal, keypad):
    from rectools_amd import ops

    B = 3
    g = torch.Generator().manual_seed(L)
    ids = torch.randint(1, 50, (B, L), generator=g)
    ids[0, : L // 3] = 0; ids[1, : L - 1] = 0
    q, k, v = rnd(B * L, d, seed=1), rnd(B * L, d, seed=2), rnd(B * L, d, seed=3)
    mask = T.attention_mask(ids, causal, keypad)
    hd = d // H

    def ref_fn(q, k, v):
        qh, kh, vh = (t.view(B, L, H, hd).transpose(1, 2) for t in (q, k, v))
        s = qh @ kh.transpose(-1, -2) / math.sqrt(hd)
        if mask is not None:
            s = s + mask[:, None]
        return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B * L, d)

    ref, gref = grads_of(ref_fn, [q, k, v])
    got, ggot = grads_of(lambda q, k, v: ops.mha(q, k, v, ids.cuda(), B, H, L, causal, keypad, 0.0), [q.cuda(), k.cuda(), v.cuda()])
    close(got, ref, rtol=5e-4, atol_rel=5e-5, msg="mha fwd")
    for i, n in enumerate("qkv"):
        close(ggot[i], gref[i], rtol=2e-3, atol_rel=2e-4, msg=f"mha d{n}")


@pytest.mark.parametrize("L,d,H,rt,rp", [(8, 16, 2, True, True), (50, 64, 2, True, False), (96, 128, 4, False, True), (130, 64, 2, True, True),
                                         (200, 128, 2, True, True), (330, 128, 2, True, True)])
def test_hstu_attention(L, d, H, rt, rp):
    from rectools_amd import ops

    B, hd = 2, d // H
    g = torch.Generator().manual_seed(L)
    ids = torch.randint(1, 50, (B, L), generator=g); ids[0, : L // 4] = 0
    ts = torch.cumsum(torch.randint(0, 3_000_000, (B, L + 1), generator=g), 1) + 1_300_000_000
    q, k, v = rnd(B * L, d, seed=1) * 0.5, rnd(B * L, d, seed=2) * 0.5, rnd(B * L, d, seed=3)
    tw, pw = rnd(129, seed=4) * 0.5, rnd(2 * L - 1, seed=5) * 0.5
    m = (ids != 0).float()

    def ref_fn(q, k, v, tw, pw):
        p = {}
        if rt: p["x.time_weights"] = tw
        if rp: p["x.pos_weights"] = pw
        rab = T.rel_attn_bias(p, "x.", {"x": ids, "unix_ts": ts}, L)
        qh, kh, vh = (t.view(B, L, H, hd) for t in (q, k, v))
        a = F.silu(torch.einsum("bnhd,bmhd->bhnm", qh, kh) + rab[:, None]) / L
        a = a * torch.tril(torch.ones(L, L))[None, None] * (m[:, None, :, None] * m[:, None, None, :])
        return torch.einsum("bhnm,bmhd->bnhd", a, vh).reshape(B * L, d)

    ref, gref = grads_of(ref_fn, [q, k, v, tw, pw])
    thr = ops.hstu_time_thresholds().cuda()
    got, ggot = grads_of(lambda q, k, v, tw, pw: ops.hstu_attn(q, k, v, tw if rt else None, pw if rp else None, ids.cuda(), ts.cuda(), thr, B, H, L),
                         [t.cuda() for t in (q, k, v, tw, pw)])
    close(got, ref, rtol=5e-4, atol_rel=5e-5, msg="hstu fwd")
    for i, n in enumerate(("dq", "dk", "dv")):
        close(ggot[i], gref[i], rtol=2e-3, atol_rel=2e-4, msg=f"hstu {n}")
    if rt: close(ggot[3], gref[3], rtol=2e-3, atol_rel=2e-4, msg="hstu dtw")
    if rp: close(ggot[4], gref[4], rtol=2e-3, atol_rel=2e-4, msg="hstu dpw")


@pytest.mark.parametrize("loss", ["BCE", "gBCE", "sampled_softmax"])
@pytest.mark.parametrize("cosine", [False, True])
def test_sampled_losses(loss, cosine):
    _sampled_case(loss, cosine, M=150, d=64, V=90, N=37, hot=False)


@pytest.mark.gpu
@pytest.mark.parametrize("cosine", [False, True])
def test_sampled_loss_popularity_skew(cosine):
    """One item is the target of most positions: its table row takes the workgroup-per-row reduction path."""
    _sampled_case("sampled_softmax", cosine, M=1500, d=128, V=400, N=5, hot=True)


@pytest.mark.parametrize("cosine", [False, True])
def test_sampled_loss_with_pairs_sorted_ahead_of_the_forward_pass(cosine):
    """`rt_sampled_loss_prepare` (ops.prepare_sampled_pairs: the counting sort of the (position, candidate) pairs on the side stream, before
    the forward pass; `rt_sampled_loss_fwd_train / _bwd(prepared = 1)`) against the sort inside the passes: same loss, same gradients."""
    from rectools_amd import ops

    M, d, V, N, t = 2048, 256, 4100, 128, 0.7
    g = torch.Generator().manual_seed(5)
    sess, emb = rnd(M, d, seed=1), rnd(V, d, seed=2, scale=0.3)
    y = torch.randint(1, V, (M,), generator=g); y[::7] = 0
    neg = torch.randint(1, V, (M, N), generator=g)
    w = torch.rand(M, generator=g) + 0.5
    out = {}
    for ahead in (False, True):
        yd, nd = y.cuda(), neg.cuda()

        def run(s, e):
            ops.RNG.next_step()
            if ahead:
                ops.prepare_sampled_pairs(yd, nd, V, d)
                assert len(ops._PREPARED_PAIRS) == 1
            loss = ops.sampled_loss(s, e, yd, nd, w.cuda(), 2, cosine, t, 0.0)[0].reshape(1)
            assert not ops._PREPARED_PAIRS           # consumed by the forward pass
            return loss

        out[ahead] = grads_of(run, [sess.cuda(), emb.cuda()])
        ops.join_side_streams()
    (l0, g0), (l1, g1) = out[False], out[True]
    close(l1, l0, rtol=1e-6, atol_rel=1e-6, msg="loss")
    close(g1[0], g0[0], rtol=1e-5, atol_rel=1e-6, msg="d_sess")
    close(g1[1], g0[1], rtol=1e-5, atol_rel=1e-6, msg="d_table")


def _sampled_case(loss, cosine, M, d, V, N, hot):
    from rectools_amd import lightning as hl
    from rectools_amd import ops

    g = torch.Generator().manual_seed(1)
    sess, table = rnd(M, d, seed=2), rnd(V, d, seed=3)
    y = torch.randint(1, V, (M,), generator=g); y[::5] = 0
    if hot:
        y[torch.rand(M, generator=g) < 0.7] = 7
    neg = torch.randint(1, V, (M, N), generator=g)
    w = (0.5 + torch.rand(M, generator=g)) * (y != 0)
    t = 0.7
    beta = hl.gbce_beta(N, V - 1, 0.2)

    def ref_fn(sess, table):
        s, e = (T._l2norm(sess), T._l2norm(table)) if cosine else (sess, table)
        cand = torch.cat([y[:, None], neg], 1)
        lg = ((e[cand] @ s.unsqueeze(-1)).squeeze(-1) / t)[None]
        yy, ww = y[None], w[None]
        if loss == "BCE": return T.bce_loss(lg, yy, ww).reshape(1)
        if loss == "gBCE": return T.bce_loss(T.gbce_logits(lg, V - 1, N, 0.2), yy, ww).float().reshape(1)
        return T.sampled_softmax_loss(lg, yy, ww).reshape(1)

    kind = {"BCE": 0, "gBCE": 1, "sampled_softmax": 2}[loss]
    ref, gref = grads_of(ref_fn, [sess, table])
    got, ggot = grads_of(lambda s, e: ops.sampled_loss(s, e, y.cuda(), neg.cuda(), w.cuda(), kind, cosine, t, beta)[0].reshape(1),
                         [sess.cuda(), table.cuda()])
    close(got, ref, rtol=2e-5, atol_rel=1e-6, msg="loss")
    gref[1][0] = 0
    close(ggot[0], gref[0], rtol=2e-3, atol_rel=2e-4, msg="d_sess"); close(ggot[1], gref[1], rtol=2e-3, atol_rel=2e-4, msg="d_table")


@pytest.mark.parametrize("cosine", [False, True])
def test_full_softmax_loss(cosine):
    from rectools_amd import ops

    M, d, V = 120, 32, 301
    g = torch.Generator().manual_seed(1)
    sess, table = rnd(M, d, seed=2), rnd(V, d, seed=3)
    y = torch.randint(1, V, (M,), generator=g); y[::3] = 0
    w = (0.5 + torch.rand(M, generator=g)) * (y != 0)

    def ref_fn(sess, table):
        s, e = (T._l2norm(sess), T._l2norm(table)) if cosine else (sess, table)
        return T.softmax_loss((s @ e.T / 0.5)[None], y[None], w[None]).reshape(1)

    def hip_fn(s, e):
        if cosine: s, e = ops.l2norm(s), ops.l2norm(e)
        act = torch.nonzero(y).reshape(-1).cuda()
        return ops.softmax_loss(s, e, act, y.cuda()[act].contiguous(), w.cuda()[act].contiguous(), 0.5).reshape(1)

    ref, gref = grads_of(ref_fn, [sess, table])
    got, ggot = grads_of(hip_fn, [sess.cuda(), table.cuda()])
    close(got, ref, rtol=2e-5, atol_rel=1e-6, msg="softmax loss")
    if not cosine: gref[1][0] = 0
    close(ggot[0], gref[0], rtol=2e-3, atol_rel=2e-4, msg="d_sess")
    if not cosine: close(ggot[1], gref[1], rtol=2e-3, atol_rel=2e-4, msg="d_table")


@pytest.mark.gpu
@pytest.mark.parametrize("p", [0.0, 0.25])
@pytest.mark.parametrize("B,L,d,H", [(3, 20, 64, 2), (2, 256, 128, 2)])
def test_fused_sasrec_layer_matches_modular_ops(B, L, d, H, p):
    """The single-node SASRec block (ops.sasrec_layer) against the same block built from the individual autograd ops:
    identical dropout streams, so outputs and every gradient must agree to rounding."""
    from rectools_amd import nn as hnn
    from rectools_amd import ops

    torch.manual_seed(0)
    layer = hnn.SASRecTransformerLayer(d, H, p).cuda().train()
    for prm in layer.parameters():  # biases / LN params away from their trivial init
        if prm.ndim == 1:
            prm.data.add_(0.1 * torch.randn_like(prm))
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1, 50, (B, L), generator=g); ids[0, : L // 3] = 0
    ids = ids.cuda()
    x = rnd(B * L, d, seed=2).cuda()
    gout = rnd(B * L, d, seed=3).cuda()

    def run(fused):
        ops.RNG.step, ops.RNG._stream = 7, 0
        for prm in layer.parameters():
            prm.grad = None
        xi = x.clone().requires_grad_(True)
        out = layer(xi, ids, B, L, True, False) if fused else layer.forward_modular(ops.mul_mask(xi, None, ids), ids, B, L, True, False)
        out.backward(gout)
        return out.detach(), xi.grad, {k: v.grad.clone() for k, v in layer.named_parameters()}

    o1, gx1, gp1 = run(True)
    o2, gx2, gp2 = run(False)
    close(o1, o2, rtol=1e-5, atol_rel=1e-6, msg="fused fwd")
    close(gx1, gx2, rtol=1e-4, atol_rel=1e-5, msg="fused dx")
    for k in gp2:
        close(gp1[k], gp2[k], rtol=1e-4, atol_rel=1e-5, msg=f"fused d{k}")


@pytest.mark.gpu
@pytest.mark.parametrize("p,L,rt,rp", [(0.0, 70, True, True), (0.3, 130, True, False), (0.0, 40, False, True)])
def test_fused_stu_layer_matches_modular_path(p, L, rt, rp):
    """ops.stu_layer (one autograd node: packed u/v/q/k gradient, strided u, LayerNorm backward with the mask and skip passes
    fused) against the same block assembled from the individual ops — same dropout streams, so p > 0 compares exactly too."""
    from rectools_amd import nn as hnn
    from rectools_amd import ops

    torch.manual_seed(0)
    B, d, H, hd = 3, 64, 2, 32
    layer = hnn.STULayer(d, H, hd, hd, L, rt, rp, p, p, 1e-6).cuda().train()
    with torch.no_grad():
        for prm in layer.parameters():
            prm.data.add_(0.1 * torch.randn_like(prm))
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(1, 50, (B, L), generator=g); ids[0, : L // 3] = 0
    ts = torch.cumsum(torch.randint(0, 3_000_000, (B, L + 1), generator=g), 1) + 1_300_000_000
    ids, batch = ids.cuda(), {"unix_ts": ts.cuda()}
    thr = ops.hstu_time_thresholds().cuda()
    x = rnd(B * L, d, seed=2).cuda()
    gout = rnd(B * L, d, seed=3).cuda()

    def run(fused):
        ops.RNG.step, ops.RNG._stream = 7, 0
        for prm in layer.parameters():
            prm.grad = None
        xi = x.clone().requires_grad_(True)
        out = layer(xi, ids, B, L, batch, thr) if fused else layer.forward_modular(ops.mul_mask(xi, None, ids), ids, B, L, batch, thr)
        out.backward(gout)
        return out.detach(), xi.grad, {k: v.grad.clone() for k, v in layer.named_parameters()}

    o1, gx1, gp1 = run(True)
    o2, gx2, gp2 = run(False)
    close(o1, o2, rtol=1e-5, atol_rel=1e-6, msg="fused stu fwd")
    close(gx1, gx2, rtol=1e-4, atol_rel=1e-5, msg="fused stu dx")
    for k in gp2:
        close(gp1[k], gp2[k], rtol=1e-4, atol_rel=1e-5, msg=f"fused stu d{k}")


@pytest.mark.gpu
def test_flat_adam_segments_match_torch_adam():
    """Segmented Adam: ragged parameter sizes, an unaligned gradient view, a parameter without gradient, 3 steps."""
    from rectools_amd import lightning as hl

    torch.manual_seed(0)
    shapes = [(7, 5), (33,), (64, 64), (1,), (130, 3), (9,)]
    ref_params = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    mine = torch.nn.ParameterList([torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_params])
    ref_opt = torch.optim.Adam(ref_params, lr=1e-2, betas=(0.9, 0.98), eps=1e-8)
    opt = hl.FlatAdam(mine, lr=1e-2, betas=(0.9, 0.98), eps=1e-8)
    for step in range(3):
        opt.zero_grad(); ref_opt.zero_grad(set_to_none=True)
        for i, (p, q) in enumerate(zip(ref_params, mine)):
            if i == 5:
                continue                                   # never receives a gradient: must stay untouched
            gr = torch.randn(p.numel() + 1, generator=torch.Generator().manual_seed(10 * step + i))
            p.grad = gr[1:].reshape(p.shape).clone()
            q.grad = gr.cuda()[1:].reshape(p.shape)        # 4-byte-aligned view: the kernel's scalar gradient path
        opt.step(); ref_opt.step()
    for p, q in zip(ref_params, mine):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_flat_adam_more_segments_than_one_launch_with_missing_gradients():
    """rt_adam_step_segments batches 48 segments per launch and skips parameters without a gradient: a skipped segment must
    not make the next launch revisit segments of the previous one (a second Adam update in one step)."""
    from rectools_amd import lightning as hl

    torch.manual_seed(1)
    shapes = [(3 + i % 5, 4) if i % 3 else (17 + i,) for i in range(110)]            # 110 segments = 3 launches
    no_grad = {0, 7, 46, 47, 48, 49, 95, 96, 109}                                     # around both batch boundaries
    ref_params = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    mine = torch.nn.ParameterList([torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_params])
    ref_opt = torch.optim.Adam(ref_params, lr=1e-2, betas=(0.9, 0.98), eps=1e-8)
    opt = hl.FlatAdam(mine, lr=1e-2, betas=(0.9, 0.98), eps=1e-8)
    for step in range(2):
        opt.zero_grad(); ref_opt.zero_grad(set_to_none=True)
        for i, (p, q) in enumerate(zip(ref_params, mine)):
            if i in no_grad:
                continue
            gr = torch.randn(p.shape, generator=torch.Generator().manual_seed(1000 * step + i))
            p.grad = gr.clone(); q.grad = gr.cuda()
        opt.step(); ref_opt.step()
    for i, (p, q) in enumerate(zip(ref_params, mine)):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=1e-5, atol=1e-7, msg=f"parameter {i}")


@pytest.mark.gpu
def test_side_stream_weight_gradients_survive_accumulation_and_parameter_slices():
    """Weight-gradient GEMMs run on a second HIP stream.  They may stay in flight until the end of the backward pass only
    when autograd merely adopts the tensors; accumulating into an existing `.grad` (second backward without zeroing) and
    parameter slices (the packed in_proj of the modular attention path) make autograd kernels read them right away."""
    from rectools_amd import nn as hnn
    from rectools_amd import ops

    torch.manual_seed(0)
    B, L, d, H = 4, 128, 128, 2
    layer = hnn.SASRecTransformerLayer(d, H, 0.0).cuda().train()
    ids = torch.randint(1, 50, (B, L)).cuda()
    x = rnd(B * L, d, seed=2).cuda()
    gout = rnd(B * L, d, seed=3).cuda()
    for fused in (True, False):
        for prm in layer.parameters():
            prm.grad = None
        fwd = (lambda: layer(x, ids, B, L, True, False)) if fused else \
              (lambda: layer.forward_modular(ops.mul_mask(x, None, ids), ids, B, L, True, False))
        fwd().backward(gout)
        once = {k: v.grad.clone() for k, v in layer.named_parameters()}
        fwd().backward(gout)                                     # accumulates into the existing .grad tensors
        for k, v in layer.named_parameters():
            close(v.grad, 2 * once[k], rtol=1e-5, atol_rel=1e-6, msg=f"accumulated d{k} (fused={fused})")



# ---- round-2 entry points, each against a plain torch fp32 reference ------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shapes,a_kc,b_kc", [
    ([(256, 128, 64), (512, 256, 64)], 1, 1),            # exact tiles: ONE grouped launch
    ([(256, 128, 128), (256, 128, 256), (128, 384, 32)], 1, 0),
    ([(300, 77, 40), (256, 128, 64)], 1, 1),             # one ragged problem: executed as consecutive rt_gemm calls
    ([(25600, 256, 256), (25600, 512, 256)], 1, 1),      # the q + k/v projections of a C2 block
])
def test_gemm_grouped_equals_reference(shapes, a_kc, b_kc):
    from rectools_amd import ops

    probs, refs, outs = [], [], []
    for i, (M, N, K) in enumerate(shapes):
        A = rnd(M, K, seed=10 + i).cuda()
        Bm = rnd(N, K, seed=20 + i, scale=0.2).cuda()            # logical B [N, K]
        bias = rnd(N, seed=30 + i).cuda() if i % 2 == 0 else None
        R = rnd(M, N, seed=40 + i).cuda() if i % 2 == 1 else None
        C = torch.empty(M, N, device="cuda")
        if b_kc:
            Bst, ldb = Bm, K                                       # k-contiguous: [N, K] row-major
        else:
            Bst, ldb = Bm.t().contiguous(), N                      # row-contiguous: element (n, k) at n + k * ld
        probs.append((A, K, Bst, ldb, C, N, bias, R, N if R is not None else 0, M, N, K, 1 if i == 0 else 0))
        ref = A.double() @ Bm.double().t()
        if bias is not None:
            ref = ref + bias.double()
        if R is not None:
            ref = ref + R.double()
        if i == 0:
            ref = ref.clamp_min(0)
        refs.append(ref.float()); outs.append(C)
    ops._gemm_group(probs, a_kc, b_kc)
    for i, (got, ref) in enumerate(zip(outs, refs)):
        close(got, ref, rtol=2e-4, atol_rel=2e-5, msg=f"grouped gemm problem {i}")


@pytest.mark.gpu
@pytest.mark.parametrize("M,d", [(37, 64), (1000, 256), (513, 512)])
def test_layernorm_masked_and_fused_backward(M, d):
    """rt_layernorm_fwd_masked / rt_layernorm_bwd_fused (row masks on dy and dx, skip-connection add) vs autograd."""
    from rectools_amd import _lib, ops

    x, w, b = rnd(M, d, seed=1), 1 + 0.1 * rnd(d, seed=2), 0.1 * rnd(d, seed=3)
    ids = torch.randint(0, 3, (M,), generator=torch.Generator().manual_seed(4))        # ~1/3 padded rows
    dy, res = rnd(M, d, seed=5), rnd(M, d, seed=6)
    mask = (ids != 0).float()[:, None]
    # forward: y = LN(x * mask), x0 = x * mask
    x0, y = torch.empty(M, d, device="cuda"), torch.empty(M, d, device="cuda")
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops._c("rt_layernorm_fwd_masked", x.cuda(), ids.cuda(), w.cuda(), b.cuda(), 1e-5, M, d, x0, y, mean, rstd)
    close(x0, x * mask, rtol=0, atol_rel=0, msg="masked input")
    close(y, F.layer_norm(x * mask, (d,), w, b, 1e-5), msg="masked layernorm")
    # backward with every fusion on: dx = mask * (LN'(mask * dy) + res)
    xr = (x * mask).clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.layer_norm(xr, (d,), wr, br, 1e-5).backward(dy * mask)
    dx, dw, db = torch.empty(M, d, device="cuda"), torch.empty(d, device="cuda"), torch.empty(d, device="cuda")
    ws_bytes = _lib.load().rt_layernorm_bwd_workspace_bytes(M, d)
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device="cuda")
    ops._c("rt_layernorm_bwd_fused", dy.cuda(), x0, w.cuda(), mean, rstd, res.cuda(), ids.cuda(), 1, 1, M, d, dx, dw, db, ws, ws_bytes)
    close(dx, (xr.grad + res) * mask, rtol=1e-3, msg="fused dx")
    close(dw, wr.grad, rtol=1e-3, msg="fused dw")
    close(db, br.grad, rtol=1e-3, msg="fused db")


@pytest.mark.gpu
@pytest.mark.parametrize("M,d", [(37, 64), (1000, 256), (513, 512)])
def test_layernorm_backward_in_two_calls_equals_the_fused_call(M, d):
    """rt_layernorm_bwd_rows + rt_layernorm_bwd_combine (the combine on ANOTHER stream behind an event, as the block executors issue it)
    == rt_layernorm_bwd_fused, bit for bit; and through the autograd op: dw / db adopted by a leaf parameter come from the side stream."""
    from rectools_amd import _lib, ops

    x, w = rnd(M, d, seed=1).cuda(), (1 + 0.1 * rnd(d, seed=2)).cuda()
    ids = torch.randint(0, 3, (M,), generator=torch.Generator().manual_seed(4)).cuda()
    dy, res = rnd(M, d, seed=5).cuda(), rnd(M, d, seed=6).cuda()
    mean, rstd = x.mean(1), 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)
    ws_bytes = _lib.load().rt_layernorm_bwd_workspace_bytes(M, d)
    new = lambda *s_: torch.empty(*s_, device="cuda")      # noqa: E731
    ws_a, ws_b = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda"), torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    dx_a, dw_a, db_a = new(M, d), new(d), new(d)
    ops._c("rt_layernorm_bwd_fused", dy, x, w, mean, rstd, res, ids, 1, 1, M, d, dx_a, dw_a, db_a, ws_a, ws_bytes)
    dx_b, dw_b, db_b = new(M, d), new(d), new(d)
    ops._c("rt_layernorm_bwd_rows", dy, x, w, mean, rstd, res, ids, 1, 1, M, d, dx_b, ws_b, ws_bytes)
    other = torch.cuda.Stream()
    other.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(other):
        ops._c("rt_layernorm_bwd_combine", ws_b, ws_bytes, M, d, dw_b, db_b)
    torch.cuda.current_stream().wait_stream(other)
    assert torch.equal(dx_a, dx_b) and torch.equal(dw_a, dw_b) and torch.equal(db_a, db_b)
    # the autograd op: leaf parameters without a .grad -> the combine runs on the library's side stream, joined before anybody reads
    wp, bp = torch.nn.Parameter(w.clone()), torch.nn.Parameter(torch.zeros(d, device="cuda"))
    xr = x.clone().requires_grad_(True)
    ops.layer_norm(xr, wp, bp, 1e-5).backward(dy)
    ops.join_side_streams()
    xt, wt, bt = x.clone().requires_grad_(True), w.clone().requires_grad_(True), torch.zeros(d, device="cuda", requires_grad=True)
    F.layer_norm(xt, (d,), wt, bt, 1e-5).backward(dy)
    close(xr.grad, xt.grad, rtol=1e-3, msg="dx"); close(wp.grad, wt.grad, rtol=1e-3, msg="dw"); close(bp.grad, bt.grad, rtol=1e-3, msg="db")


@pytest.mark.gpu
def test_loss_backward_reads_the_upstream_gradient_on_the_device():
    """`loss.backward(g)` with g != 1 (a tensor on the device): the sampled loss and the full softmax scale their gradients by g inside
    the kernels (rt_sampled_loss_bwd / rt_softmax_ce_rows `upstream`) — no host read, no divide launch; gradients = g x the unit ones."""
    from rectools_amd import ops

    M, d, V, N = 96, 64, 300, 7
    g0 = torch.Generator().manual_seed(0)
    table0 = (rnd(V, d, seed=1) * 0.3)
    sess0 = rnd(M, d, seed=2) * 0.3
    y = torch.randint(1, V, (M,), generator=g0); y[::5] = 0
    neg = torch.randint(1, V, (M, N), generator=g0)
    w = torch.ones(M)
    grads = {}
    for scale in (1.0, 2.5):
        table, sess = table0.clone().cuda().requires_grad_(True), sess0.clone().cuda().requires_grad_(True)
        loss, _ = ops.sampled_loss(sess, table, y.cuda(), neg.cuda(), w.cuda(), ops.LOSS_SAMPLED_SOFTMAX, False, 1.0)
        loss.backward(torch.full_like(loss, scale))
        ops.join_side_streams()
        grads[scale] = (sess.grad.clone(), table.grad.clone())
    close(grads[2.5][0], 2.5 * grads[1.0][0], rtol=1e-6, msg="d_sess scales with the upstream gradient")
    close(grads[2.5][1], 2.5 * grads[1.0][1], rtol=1e-6, msg="d_table scales with the upstream gradient")


@pytest.mark.gpu
@pytest.mark.parametrize("d,H,window,pad_keys", [(256, 4, 200, True), (256, 4, 200, False), (64, 2, 50, True), (128, 1, 30, True), (512, 4, 200, True),
                                                 (512, 8, 512, False), (192 + 64, 2, 20, True)])
def test_last_query_without_key_value_rows(d, H, window, pad_keys):
    """rt_mha_varlen_last_x_fwd (qk = W_k,h^T q_h in, xbar = sum_j p_j x_j out; the caller applies W_v,h) == rt_mha_varlen_last_fwd on the
    projected keys / values of every row, incl. the window's pad keys (logit q.b_k, value b_v) — the two forms of the final block's
    attention in recommend() (sasrec.py:221-224 at the last position)."""
    from rectools_amd import ops

    B, hd = 37, d // H
    g = torch.Generator().manual_seed(d + H)
    lens = torch.randint(1, window + 1, (B,), generator=g)
    lens[0], lens[1] = window, 1
    cu = torch.zeros(B + 1, dtype=torch.int64); cu[1:] = torch.cumsum(lens, 0)
    n = int(cu[-1])
    x = (rnd(n, d, seed=1) * 0.7).cuda()
    Wk, Wv = (rnd(d, d, seed=2) * 0.1).cuda(), (rnd(d, d, seed=3) * 0.1).cuda()
    bk, bv = (rnd(d, seed=4) * 0.3).cuda(), (rnd(d, seed=5) * 0.3).cuda()
    q = rnd(B, d, seed=6).cuda()
    cud = cu.cuda()
    # the projection form: K | V of every row
    K, V = x @ Wk.T + bk, x @ Wv.T + bv
    want = torch.empty(B, d, device="cuda")
    ops._c("rt_mha_varlen_last_fwd", q, d, K, d, V, d, cud, bk if pad_keys else None, bv if pad_keys else None, B, H, hd, window, window, want, d)
    # the projection-free form
    qk = torch.einsum("bhe,hed->bhd", q.view(B, H, hd), Wk.view(H, hd, d)).contiguous()
    xbar = torch.empty(B, H, d, device="cuda")
    ops._c("rt_mha_varlen_last_x_fwd", qk, x, d, cud, B, H, d, window, window, int(pad_keys), -1, xbar)
    got = torch.einsum("bhd,hed->bhe", xbar, Wv.view(H, hd, d)).reshape(B, d) + bv
    close(got, want, rtol=2e-4, atol_rel=2e-5, msg="last query, projection-free")
    # fp64 restatement of the reference's own formula for one session (nn.MultiheadAttention on the left-padded window)
    b = 0 if not pad_keys else 2
    nb, npad = int(lens[b]), (window - int(lens[b])) if pad_keys else 0
    xs = x[int(cu[b]):int(cu[b + 1])].double().cpu()
    xs = torch.cat([torch.zeros(npad, d, dtype=torch.float64), xs])
    Kr, Vr = xs @ Wk.double().cpu().T + bk.double().cpu(), xs @ Wv.double().cpu().T + bv.double().cpu()
    outs = []
    for h in range(H):
        sc = (Kr[:, h * hd:(h + 1) * hd] @ q[b, h * hd:(h + 1) * hd].double().cpu()) / hd ** 0.5
        outs.append(torch.softmax(sc, 0) @ Vr[:, h * hd:(h + 1) * hd])
    close(got[b].cpu().double(), torch.cat(outs), rtol=2e-4, atol_rel=2e-5, msg="last query vs fp64")


@pytest.mark.gpu
@pytest.mark.parametrize("d,V,L", [(256, 500, 200), (64, 90, 16), (512, 300, 50)])
def test_first_block_inputs_from_projected_tables(d, V, L):
    """rt_embed_block1_fwd: LN1(x), Q = W_q LN1(x) + b_q and K | V = W_kv x + b_kv of x = scale E[id] + P[dist] gathered from tables projected
    once (Q = rstd (scale QE[id] + QP[dist] - mean W_q g) + W_q beta + b_q) == the embedding, LayerNorm and Linear ops on the rows."""
    from rectools_amd import ops

    M, scale, eps = 777, 1.7, 1e-8
    g0 = torch.Generator().manual_seed(d)
    E, P = (rnd(V, d, seed=1) * 0.5).cuda(), (rnd(L, d, seed=2) * 0.5).cuda()
    E[0] = 0
    ids = torch.randint(0, V, (M,), generator=g0).cuda()
    dist = torch.randint(0, L, (M,), generator=g0).cuda()
    g, beta = (1 + 0.2 * rnd(d, seed=3)).cuda(), (0.1 * rnd(d, seed=4)).cuda()
    in_w, in_b = (rnd(3 * d, d, seed=5) * 0.1).cuda(), (rnd(3 * d, seed=6) * 0.2).cuda()
    # the rows' own ops
    x = E[ids] * scale + P[dist]
    q_ref = F.layer_norm(x, (d,), g, beta, eps)
    Q_ref = q_ref @ in_w[:d].T + in_b[:d]
    KV_ref = x @ in_w[d:].T + in_b[d:]
    # the projected tables (what nn.TransformerTorchBackbone.encode_last_packed builds once per recommend() call)
    QE, QP = (E * g) @ in_w[:d].T, (P * g) @ in_w[:d].T
    wg, wb = in_w[:d] @ g, in_w[:d] @ beta + in_b[:d]
    KVE, KVP = E @ in_w[d:].T, P @ in_w[d:].T + in_b[d:]
    q, Q, KV = torch.empty(M, d, device="cuda"), torch.empty(M, d, device="cuda"), torch.empty(M, 2 * d, device="cuda")
    ops._c("rt_embed_block1_fwd", ids, dist, E, P, scale, g, beta, eps, QE.contiguous(), QP.contiguous(), wg.contiguous(), wb.contiguous(),
           KVE.contiguous(), KVP.contiguous(), M, d, q, Q, KV)
    close(q, q_ref, rtol=2e-4, atol_rel=2e-5, msg="LN1(x)")
    close(Q, Q_ref, rtol=2e-4, atol_rel=2e-5, msg="queries")
    close(KV, KV_ref, rtol=2e-4, atol_rel=2e-5, msg="keys | values")


@pytest.mark.gpu
def test_mul_mask_ld_strided_slices():
    from rectools_amd import ops

    M, hh = 300, 64
    packed = rnd(M, 4 * hh, seed=1).cuda()
    other = rnd(M, hh, seed=2).cuda()
    ids = torch.randint(0, 2, (M,), generator=torch.Generator().manual_seed(3)).cuda()
    out = torch.zeros(M, 4 * hh, device="cuda")
    # read the second column block of `packed`, write into the third column block of `out`
    ops._c("rt_mul_mask_ld", packed[:, hh:], 4 * hh, other, hh, ids, M, hh, out[:, 2 * hh:], 4 * hh)
    ref = torch.zeros_like(out)
    ref[:, 2 * hh:3 * hh] = packed[:, hh:2 * hh] * other * (ids != 0).float()[:, None]
    close(out, ref, rtol=0, atol_rel=0, msg="strided mul_mask")


@pytest.mark.gpu
@pytest.mark.parametrize("L,d,H,causal,keypad", [(8, 16, 2, True, False), (70, 128, 4, False, True), (200, 256, 4, True, True), (300, 64, 1, True, False)])
def test_mha_last_query_equals_full_attention(L, d, H, causal, keypad):
    """rt_mha_last_fwd (one query row per session) vs the last row of rt_mha_fwd."""
    from rectools_amd import ops

    B = 5
    g = torch.Generator().manual_seed(L)
    ids = torch.randint(1, 50, (B, L), generator=g); ids[0, : L // 3] = 0; ids[1, : L - 1] = 0
    q, k, v = rnd(B * L, d, seed=1).cuda(), rnd(B * L, d, seed=2).cuda(), rnd(B * L, d, seed=3).cuda()
    ids = ids.cuda()
    full = ops.mha(q, k, v, ids, B, H, L, causal, keypad, 0.0).view(B, L, d)[:, -1, :]
    q_last = q.view(B, L, d)[:, -1, :].contiguous()
    out = torch.empty(B, d, device="cuda")
    ops._c("rt_mha_last_fwd", q_last, d, k, d, v, d, ids.reshape(-1), B, H, L, d // H, int(causal), int(keypad), out, d)
    close(out, full, rtol=2e-5, atol_rel=2e-6, msg="last-query attention")


@pytest.mark.gpu
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (1, 0), (0, 1), (0, 0)])
@pytest.mark.parametrize("M,N,K,split_k", [(256, 128, 64, 1), (25600, 256, 256, 1), (256, 256, 25600, 16)])
def test_gemm_bf16x6_split_is_fp32_accurate(M, N, K, split_k, a_kc, b_kc, monkeypatch):
    """The default inner loop of the exact-tile GEMM: every product of an exact 3-way bf16 split down to 2^-16, on the bf16
    matrix pipe (RT_GEMM_SPLIT=exact selects the f32-input MFMA instead).
    The error against an fp64 product must be of the size of the exact fp32 kernel's own (both bounded relative to
    sum_k |a||b|), for all four operand layouts, the split-K combine and the grouped launch."""
    from rectools_amd import ops

    A = rnd(M, K, seed=1).cuda()                                   # logical A [M, K], B [N, K]
    Bm = rnd(N, K, seed=2, scale=0.3).cuda()
    A[0, :8] = torch.tensor([1e-20, -3e19, 1.0, -1.0, 0.0, 2.0 ** -120, 65504.0, 1.0 + 2.0 ** -23]).cuda()   # exponent range
    Ast, lda = (A, K) if a_kc else (A.t().contiguous(), M)
    Bst, ldb = (Bm, K) if b_kc else (Bm.t().contiguous(), N)
    ref = A.double() @ Bm.double().t()
    scale = A.double().abs() @ Bm.double().abs().t() + 1e-300

    def run():
        C = torch.empty(M, N, device="cuda")
        ops._gemm(Ast, lda, a_kc, Bst, ldb, b_kc, C, N, None, None, 0, M, N, K, 0, split_k)
        return C

    monkeypatch.setenv("RT_GEMM_SPLIT", "exact")
    exact = run()
    monkeypatch.delenv("RT_GEMM_SPLIT")                            # the default
    split = run()
    if split_k == 1 and a_kc:
        C2 = torch.empty(M, N, device="cuda")                      # grouped launch, same switch
        ops._gemm_group([(Ast, lda, Bst, ldb, split.new_empty(M, N), N, None, None, 0, M, N, K, 0),
                         (Ast, lda, Bst, ldb, C2, N, None, None, 0, M, N, K, 0)], a_kc, b_kc)
        assert torch.equal(C2, split), "grouped bf16x6 launch differs from the single launch"
    monkeypatch.setenv("RT_GEMM_SPLIT", "exact")
    assert torch.equal(run(), exact), "the switch must not leak into the exact kernel"
    e_exact = float(((exact.double() - ref).abs() / scale).max())
    e_split = float(((split.double() - ref).abs() / scale).max())
    assert not torch.equal(split, exact) or M * N <= 256 * 128, "bf16x6 path did not run (bitwise equal to the exact kernel)"
    bound = 2.0 ** -24 * (3 + 2 * (K // max(split_k, 1)) ** 0.5)   # 3 dropped-term units + random-walk accumulation rounding
    assert e_split < bound, f"bf16x6 error {e_split:.2e} (exact kernel {e_exact:.2e}, bound {bound:.2e})"


@pytest.mark.parametrize("w_tr", [0, 1])
def test_gemm_with_presplit_weight_planes_is_fp32_accurate(w_tr):
    """`rt_split_planes` + `rt_gemm_wp` (csrc/rt_gemm_wp.hip: weights split once into three bf16 planes, activations split in registers,
    six bf16 MFMA products per fp32 product; w_tr = 1 reads the planes through the transpose LDS read) against an fp64 product: error at
    the level of the f32-input MFMA kernel, for a single product with bias + residual + relu and for a grouped launch of two products
    that share one plane buffer (the q / kv projections of a block); the planes themselves reassemble the weights exactly."""
    import ctypes

    from rectools_amd import _lib, ops

    torch.manual_seed(3 + w_tr)
    M, d = 384, 256
    W = (torch.randn(3 * d, d) * torch.logspace(-2, 1, 3 * d)[:, None]).cuda()       # in_proj-like [3d, d], rows spanning the exponent range
    n = W.numel()
    stride = (n + 7) // 8 * 8
    planes = torch.empty(3 * stride, dtype=torch.int16, device="cuda")
    ops._c("rt_split_planes", W, n, planes, stride)
    pl = planes.view(3, stride)[:, :n].view(torch.bfloat16).float().view(3, 3 * d, d)
    assert torch.equal(pl[0] + pl[1] + pl[2], W)                                      # exact three-way split

    def run(problems):
        arr = (_lib.GemmWpProblem * len(problems))()
        for q, (A, Wp, ldw, C, bias, R, Mq, Nq, Kq, relu) in zip(arr, problems):
            q.A, q.lda, q.W, q.plane_stride, q.ldw, q.C, q.ldc = A.data_ptr(), A.stride(0), Wp, stride, ldw, C.data_ptr(), C.stride(0)
            q.bias, q.R, q.ldr = (None if bias is None else bias.data_ptr()), (None if R is None else R.data_ptr()), (0 if R is None else R.stride(0))
            q.M, q.N, q.K, q.relu = Mq, Nq, Kq, relu
        ops._c("rt_gemm_wp", ctypes.cast(arr, ctypes.c_void_p), len(problems), w_tr)

    def err(got, ref):
        return float((got.double() - ref).abs().max() / ref.abs().max())

    bias, R = torch.randn(d).cuda(), torch.randn(M, d).cuda()
    if w_tr == 0:      # y = x W^T: Wq = rows [0, d), Wkv = rows [d, 3d)
        x = torch.randn(M, d).cuda()
        yq, ykv = torch.empty(M, d, device="cuda"), torch.empty(M, 2 * d, device="cuda")
        run([(x, planes.data_ptr(), d, yq, bias, R, M, d, d, 1), (x, planes.data_ptr() + 2 * d * d, d, ykv, None, None, M, 2 * d, d, 0)])
        ref_q = torch.relu(x.double() @ W[:d].double().t() + bias.double() + R.double())
        ref_kv = x.double() @ W[d:].double().t()
        assert err(yq, ref_q) < 2e-6 and err(ykv, ref_kv) < 2e-6, (err(yq, ref_q), err(ykv, ref_kv))
        y1 = torch.empty(M, d, device="cuda")
        run([(x, planes.data_ptr(), d, y1, None, None, M, d, d, 0)])
        y2 = torch.empty(M, d, device="cuda")
        ops._gemm(x, d, 1, W, d, 1, y2, d, None, None, 0, M, d, d)                    # rt_gemm (both operands split in registers)
        torch.testing.assert_close(y1, y2, rtol=1e-5, atol=1e-6 * float(y2.abs().max()))
    else:              # dx = dy W: g_q = gQ Wq (+ residual), g_kv = gKV Wkv
        gQ, gKV = torch.randn(M, d).cuda(), torch.randn(M, 2 * d).cuda()
        g_q, g_kv = torch.empty(M, d, device="cuda"), torch.empty(M, d, device="cuda")
        run([(gQ, planes.data_ptr(), d, g_q, None, R, M, d, d, 0), (gKV, planes.data_ptr() + 2 * d * d, d, g_kv, None, None, M, d, 2 * d, 0)])
        ref_q = gQ.double() @ W[:d].double() + R.double()
        ref_kv = gKV.double() @ W[d:].double()
        assert err(g_q, ref_q) < 2e-6 and err(g_kv, ref_kv) < 2e-6, (err(g_q, ref_q), err(g_kv, ref_kv))
    bad = torch.empty(100, d, device="cuda")      # not an exact tile grid: refused (the caller falls back to rt_gemm), never computed otherwise
    with pytest.raises(NotImplementedError):
        run([(torch.randn(100, d).cuda(), planes.data_ptr(), d, bad, None, None, 100, d, d, 0)])


@pytest.mark.parametrize("rows", [384, 389, 100])
def test_linear_and_matmul_nn_read_the_armed_weight_planes(rows):
    """`ops.active_planes` (K7w outside the native executors: the LiGR / STU stacks): a product whose weight lies in the armed planes runs
    its full 128-row tiles on `rt_gemm_wp` and the rows behind them on `rt_gemm`; same six bf16 products per fp32 product, so the results
    agree with the unarmed product to fp32 rounding — forward, data gradient (which reads the planes found in the forward pass) and
    weight gradient.  389 rows: a tail behind three tiles; 100 rows: no full tile, `rt_gemm` only."""
    from rectools_amd import lightning as hl
    from rectools_amd import nn as hnn
    from rectools_amd import ops

    torch.manual_seed(5)
    stack = hnn.LiGRLayers(1, 256, 2, 0.0).cuda()
    hl.FlatAdam(stack, lr=1e-3)                        # parameters become views of one flat buffer
    blk = stack.transformer_blocks[0]
    w, b = blk.feed_forward.ff_linear_1.weight, blk.multi_head_attn.in_proj_bias[:256]
    p_nn = blk.multi_head_attn.in_proj_weight[:256]   # a [K, N] operand for matmul_nn (a row slice: contiguous, inside the planes)
    x = torch.randn(rows, 256, device="cuda", requires_grad=True)
    res = torch.randn(rows, w.shape[0], device="cuda")

    def run(armed):
        for t in (x, w):
            t.grad = None
        planes = stack._fresh_planes() if armed else None
        assert (planes is not None) == armed
        with ops.active_planes(planes):
            assert (ops._planes_of(w) is not None) == armed and (ops._planes_of(p_nn) is not None) == armed
            y = ops.linear(x, w, None, res, relu=True)
            z = ops.matmul_nn(x, p_nn)
        (y.sum() + (z * z).sum()).backward()           # backward OUTSIDE the block: the nodes kept what they found
        torch.cuda.synchronize()
        return y.detach().clone(), z.detach().clone(), x.grad.clone(), w.grad.clone()

    got, ref = run(True), run(False)
    for a, r, what in zip(got, ref, ("linear", "matmul_nn", "dx", "dw")):
        torch.testing.assert_close(a, r, rtol=2e-6, atol=2e-6 * float(r.abs().max()), msg=lambda m, what=what: f"{what}: {m}")
    ref64 = torch.relu(x.detach().double() @ w.detach().double().t() + res.double())
    assert float((got[0].double() - ref64).abs().max() / ref64.abs().max()) < 2e-6
