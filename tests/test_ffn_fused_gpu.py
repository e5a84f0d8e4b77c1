"""K7f (csrc/rt_ffn.hip): the feed-forward half of a SASRec block as one launch per direction against the five-launch sequence it
replaces (rt_layernorm_fwd, rt_gemm_wp, rt_act_dropout_fwd, rt_gemm_wp, rt_act_dropout_fwd / their backward twins) on the same weight
planes and dropout streams, and against an fp64 restatement of sasrec.py:225-229 + net_blocks.py:63-64 with the kernel's own masks."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

SEEDS = dict(seed_h=0x1234_5678_9ABC, sid_h=7, seed_o=0x0FED_CBA9_8765, sid_o=11)


def _planes(w):
    from rectools_amd import ops

    n = w.numel()
    stride = (n + 7) // 8 * 8
    planes = torch.empty(3 * stride, dtype=torch.int16, device=w.device)
    ops._c("rt_split_planes", w, n, planes, stride)
    return planes, stride


def _wp(A, planes, stride, ldw, C, M, N, K, tr, bias=None, R=None, relu=0):
    from rectools_amd import _lib, ops

    arr = (_lib.GemmWpProblem * 1)()
    q = arr[0]
    q.A, q.lda, q.W, q.plane_stride, q.ldw, q.C, q.ldc = A.data_ptr(), A.stride(0), planes.data_ptr(), stride, ldw, C.data_ptr(), C.stride(0)
    q.bias = None if bias is None else bias.data_ptr()
    q.R, q.ldr = (None, 0) if R is None else (R.data_ptr(), R.stride(0))
    q.M, q.N, q.K, q.relu = M, N, K, relu
    ops._c("rt_gemm_wp", ctypes.cast(arr, ctypes.c_void_p), 1, tr)


def _inputs(M, d, dff, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)   # noqa: E731
    y = r(M, d) * 1.5 + 0.3
    w1, w2 = r(dff, d) / d ** 0.5, r(d, dff) / dff ** 0.5
    return dict(y=y, ln_w=1.0 + 0.1 * r(d), ln_b=0.1 * r(d), w1=w1, w2=w2, b1=0.1 * r(dff), b2=0.1 * r(d), g_out=r(M, d))


def _fused_fwd(t, M, d, dff, p):
    from rectools_amd import ops

    # the two weights' planes share ONE plane stride: laid out as the block does — [W1 | W2] split as one range
    both = torch.cat([t["w1"].reshape(-1), t["w2"].reshape(-1)])
    planes, stride = _planes(both)
    w1p, w2p = planes, planes[t["w1"].numel():]
    f, hd, out = torch.empty(M, d, device="cuda"), torch.empty(M, dff, device="cuda"), torch.empty(M, d, device="cuda")
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops._c("rt_ffn_fused_fwd", t["y"], t["ln_w"], t["ln_b"], 1e-5, f, mean, rstd, w1p, w2p, stride, t["b1"], t["b2"], hd, out, M, d, dff, p,
           SEEDS["seed_h"], SEEDS["sid_h"], SEEDS["seed_o"], SEEDS["sid_o"])
    return dict(f=f, mean=mean, rstd=rstd, hdrop=hd, out=out, w1p=w1p, w2p=w2p, stride=stride)


def _unfused_fwd(t, fz, M, d, dff, p):
    from rectools_amd import ops

    f, mean, rstd = torch.empty(M, d, device="cuda"), torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops._c("rt_layernorm_fwd", t["y"], t["ln_w"], t["ln_b"], 1e-5, M, d, f, mean, rstd)
    h, hd, o, out = (torch.empty(M, n, device="cuda") for n in (dff, dff, d, d))
    _wp(f, fz["w1p"], fz["stride"], d, h, M, dff, d, 0, bias=t["b1"], relu=1)
    ops._c("rt_act_dropout_fwd", h, 0, p, SEEDS["seed_h"], SEEDS["sid_h"], M * dff, None, hd)
    _wp(hd, fz["w2p"], fz["stride"], dff, o, M, d, dff, 0, bias=t["b2"])
    ops._c("rt_act_dropout_fwd", o, 0, p, SEEDS["seed_o"], SEEDS["sid_o"], M * d, f, out)
    return dict(f=f, mean=mean, rstd=rstd, h=h, hdrop=hd, out=out)


@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("M,d,dff", [(256, 256, 256), (128, 128, 128), (384, 128, 256), (128, 256, 128), (13312, 256, 256)])
def test_fused_forward_equals_the_five_launch_sequence(M, d, dff, p):
    t = _inputs(M, d, dff)
    fz = _fused_fwd(t, M, d, dff, p)
    un = _unfused_fwd(t, fz, M, d, dff, p)
    torch.cuda.synchronize()
    # the prologue restates layernorm_fwd_kernel's sums (the compiler may contract them differently: a rounding unit); the products run
    # on the same planes in the same k and term order, the masks come from the same hash
    for k in ("f", "mean", "rstd"):
        torch.testing.assert_close(fz[k], un[k], rtol=2e-6, atol=2e-6)
    assert float(((fz["hdrop"] != 0) != (un["hdrop"] != 0)).float().mean()) < 1e-4        # (a relu input within a rounding unit of zero)
    torch.testing.assert_close(fz["hdrop"], un["hdrop"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(fz["out"], un["out"], rtol=1e-5, atol=2e-5)
    print(f"[ffn fwd {M}x{d}x{dff} p={p}] bit-equal hdrop {torch.equal(fz['hdrop'], un['hdrop'])} out {torch.equal(fz['out'], un['out'])}")


@pytest.mark.parametrize("p", [0.0, 0.25])
def test_fused_forward_against_fp64_with_its_own_masks(p):
    M, d, dff = 192, 256, 256
    t = _inputs(M, d, dff, seed=3)
    fz = _fused_fwd(t, M, d, dff, p)
    torch.cuda.synchronize()
    D = {k: v.double() for k, v in t.items()}
    f = torch.nn.functional.layer_norm(D["y"], (d,), D["ln_w"], D["ln_b"], 1e-5)
    h = torch.relu(f @ D["w1"].T + D["b1"])
    keep_h = (fz["hdrop"] != 0) | (h <= 0)            # where h > 0 the kept elements are exactly the non-zero ones
    hd = h * keep_h / (1 - p)
    o = hd @ D["w2"].T + D["b2"]
    # the output mask is not observable from `out` alone: regenerate it through the stand-alone dropout kernel on ones
    from rectools_amd import ops

    ones, mask_o = torch.ones(M, d, device="cuda"), torch.empty(M, d, device="cuda")
    ops._c("rt_act_dropout_fwd", ones, 0, p, SEEDS["seed_o"], SEEDS["sid_o"], M * d, None, mask_o)
    want = f + o * mask_o.double()
    torch.testing.assert_close(fz["f"].double(), f, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(fz["hdrop"].double(), hd, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(fz["out"].double(), want, rtol=1e-5, atol=2e-5)
    if p > 0:
        rate = float((mask_o == 0).float().mean())
        assert abs(rate - p) < 0.01, rate              # 49k draws: 5 sigma = 0.0098


@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("M,d,dff", [(256, 256, 256), (128, 128, 128), (384, 128, 256), (128, 256, 128), (13312, 256, 256)])
def test_fused_backward_equals_the_unfused_sequence(M, d, dff, p):
    from rectools_amd import ops

    t = _inputs(M, d, dff, seed=5)
    fz = _fused_fwd(t, M, d, dff, p)
    un = _unfused_fwd(t, fz, M, d, dff, p)
    g_o, g_h, g_f = torch.full((M, d), float("nan"), device="cuda"), torch.empty(M, dff, device="cuda"), torch.empty(M, d, device="cuda")
    ops._c("rt_ffn_fused_bwd", t["g_out"], fz["hdrop"], fz["w1p"], fz["w2p"], fz["stride"], g_o if p > 0 else None, g_h, g_f, M, d, dff, p,
           SEEDS["seed_o"], SEEDS["sid_o"])
    # the sequence of rt_block.hip's unfused backward
    r_o = t["g_out"]
    if p > 0:
        r_o = torch.empty(M, d, device="cuda")
        ops._c("rt_act_dropout_bwd", t["g_out"], t["g_out"], 0, p, SEEDS["seed_o"], SEEDS["sid_o"], M * d, r_o)
    g_hd, r_h, r_f = torch.empty(M, dff, device="cuda"), torch.empty(M, dff, device="cuda"), torch.empty(M, d, device="cuda")
    _wp(r_o, fz["w2p"], fz["stride"], dff, g_hd, M, dff, d, 1)
    ops._c("rt_act_dropout_bwd", g_hd, un["h"], 1, p, SEEDS["seed_h"], SEEDS["sid_h"], M * dff, r_h)
    _wp(r_h, fz["w1p"], fz["stride"], d, r_f, M, d, dff, 1, R=t["g_out"])
    torch.cuda.synchronize()
    if p > 0:
        assert torch.equal(g_o, r_o)
    torch.testing.assert_close(g_h, r_h, rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(g_f, r_f, rtol=2e-6, atol=4e-6)
    # ... and against fp64 autograd of the same function with the kernel's masks
    D = {k: v.double().requires_grad_(k == "y") for k, v in t.items()}
    f = torch.nn.functional.layer_norm(D["y"], (d,), D["ln_w"], D["ln_b"], 1e-5).detach().requires_grad_(True)
    h = torch.relu(f @ D["w1"].T + D["b1"])
    keep_h = ((fz["hdrop"] != 0) | (h <= 0)).double() / (1 - p)
    mask_o = (g_o != 0).double() / (1 - p) if p > 0 else torch.ones(M, d, device="cuda", dtype=torch.float64)
    out = f + ((h * keep_h) @ D["w2"].T + D["b2"]) * mask_o
    (gf64,) = torch.autograd.grad(out, f, D["g_out"])
    torch.testing.assert_close(g_f.double(), gf64, rtol=1e-5, atol=2e-5)


def test_unsupported_shapes_are_refused():
    from rectools_amd import _lib

    lib = _lib.load()
    assert lib.rt_ffn_fused_supported(128, 256, 256) == 1
    assert lib.rt_ffn_fused_supported(96, 256, 256) == 0 and lib.rt_ffn_fused_supported(128, 64, 256) == 0
    assert lib.rt_ffn_fused_supported(128, 256, 200) == 0 and lib.rt_ffn_fused_supported(0, 256, 256) == 0
    assert lib.rt_ffn_fused_supported(128, 512, 512) == 0 and lib.rt_ffn_fused_supported(128, 256, 384) == 0     # the operand rows must fit the LDS


# ---- the whole tail of a block behind its attention: out-projection + skip, LN2, feed-forward (three products) -----------------------
def _tail_inputs(M, d, dff, seed=0):
    t = _inputs(M, d, dff, seed)
    g = torch.Generator(device="cuda").manual_seed(seed + 100)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)   # noqa: E731
    t.update(attn=r(M, d), q=r(M, d) * 1.5 + 0.3, wo=r(d, d) / d ** 0.5, bo=0.1 * r(d))
    return t


def _tail_planes(t):
    both = torch.cat([t["wo"].reshape(-1), t["w1"].reshape(-1), t["w2"].reshape(-1)])      # one split over the stack's range, as ops.WeightPlanes
    planes, stride = _planes(both)
    n0, n1 = t["wo"].numel(), t["w1"].numel()
    return dict(wop=planes, w1p=planes[n0:], w2p=planes[n0 + n1:], stride=stride, _keep=planes)


def _tail_fwd(t, pl, M, d, dff, p, training):
    from rectools_amd import ops

    y, f, hd, out = (torch.full((M, n), float("nan"), device="cuda") for n in (d, d, dff, d))
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    ops._c("rt_block_tail_fwd", t["attn"], t["q"], pl["wop"], t["bo"], t["ln_w"], t["ln_b"], 1e-5, y if training else None, f,
           mean if training else None, rstd if training else None, pl["w1p"], pl["w2p"], pl["stride"], t["b1"], t["b2"], hd if training else None, out,
           M, d, dff, p, SEEDS["seed_h"], SEEDS["sid_h"], SEEDS["seed_o"], SEEDS["sid_o"], 1 if training else 0)
    return dict(y=y, f=f, mean=mean, rstd=rstd, hdrop=hd, out=out)


def _tail_unfused_fwd(t, pl, M, d, dff, p):
    y = torch.empty(M, d, device="cuda")
    _wp(t["attn"], pl["wop"], pl["stride"], d, y, M, d, d, 0, bias=t["bo"], R=t["q"])
    r = _unfused_fwd(dict(t, y=y), pl, M, d, dff, p)
    r["y"] = y
    return r


@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("M,d,dff", [(256, 256, 256), (128, 128, 128), (13312, 256, 256)])
def test_block_tail_forward_equals_the_separate_launches(M, d, dff, p):
    t = _tail_inputs(M, d, dff)
    pl = _tail_planes(t)
    fz = _tail_fwd(t, pl, M, d, dff, p, True)
    un = _tail_unfused_fwd(t, pl, M, d, dff, p)
    torch.cuda.synchronize()
    assert torch.equal(fz["y"], un["y"])             # same planes, same k order and term order: bit-identical product
    for k in ("f", "mean", "rstd", "hdrop", "out"):  # behind the LayerNorm: its sums may be contracted differently (a rounding unit)
        torch.testing.assert_close(fz[k], un[k], rtol=1e-5, atol=2e-5)
    if p == 0.0:                                     # the inference form: nothing but `out` (and the scratch rows f)
        inf = _tail_fwd(t, pl, M, d, dff, 0.0, False)
        torch.cuda.synchronize()
        assert torch.equal(inf["out"], fz["out"]) and torch.equal(inf["f"], fz["f"])
        assert bool(torch.isnan(inf["y"]).all()) and bool(torch.isnan(inf["hdrop"]).all())      # untouched


@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("M,d,dff", [(256, 256, 256), (128, 128, 128), (13312, 256, 256)])
def test_block_tail_backward_equals_the_separate_launches(M, d, dff, p):
    from rectools_amd import _lib, ops

    lib = _lib.load()
    t = _tail_inputs(M, d, dff, seed=7)
    pl = _tail_planes(t)
    fw = _tail_fwd(t, pl, M, d, dff, p, True)
    g_o, g_h, g_y, g_A = (torch.full((M, n), float("nan"), device="cuda") for n in (d, dff, d, d))
    part = torch.empty(lib.rt_block_tail_partial_floats(M, d), device="cuda")
    dw, db = torch.empty(d, device="cuda"), torch.empty(d, device="cuda")
    ops._c("rt_block_tail_bwd", t["g_out"], fw["hdrop"], fw["y"], fw["mean"], fw["rstd"], t["ln_w"], pl["wop"], pl["w1p"], pl["w2p"], pl["stride"],
           g_o if p > 0 else None, g_h, g_y, g_A, part, M, d, dff, p, SEEDS["seed_o"], SEEDS["sid_o"])
    ops._c("rt_layernorm_bwd_reduce", part, M // 64, d, dw, db)
    # the separate launches: fused feed-forward backward (tested above against its own unfused form), LayerNorm backward, data gradient
    r_o, r_h, r_f, r_y, r_A = (torch.empty(M, n, device="cuda") for n in (d, dff, d, d, d))
    ops._c("rt_ffn_fused_bwd", t["g_out"], fw["hdrop"], pl["w1p"], pl["w2p"], pl["stride"], r_o if p > 0 else None, r_h, r_f, M, d, dff, p,
           SEEDS["seed_o"], SEEDS["sid_o"])
    ws = torch.empty(lib.rt_layernorm_bwd_workspace_bytes(M, d), dtype=torch.uint8, device="cuda")
    rw, rb = torch.empty(d, device="cuda"), torch.empty(d, device="cuda")
    ops._c("rt_layernorm_bwd_fused", r_f, fw["y"], t["ln_w"], fw["mean"], fw["rstd"], None, None, 0, 0, M, d, r_y, rw, rb, ws, ws.numel())
    _wp(r_y, pl["wop"], pl["stride"], d, r_A, M, d, d, 1)
    torch.cuda.synchronize()
    if p > 0:
        assert torch.equal(g_o, r_o)
    assert torch.equal(g_h, r_h)
    torch.testing.assert_close(g_y, r_y, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(g_A, r_A, rtol=2e-6, atol=4e-6)
    scale = float(rw.abs().max())
    torch.testing.assert_close(dw, rw, rtol=1e-5, atol=1e-5 * scale)          # (another partition of the rows: another summation order)
    torch.testing.assert_close(db, rb, rtol=1e-5, atol=1e-5 * float(rb.abs().max()))
    # fp64 autograd of the same function with the kernel's masks, down to the attention's output
    D = {k: v.double() for k, v in t.items()}
    attn = D["attn"].clone().requires_grad_(True)
    lnw = D["ln_w"].clone().requires_grad_(True)
    y = D["q"] + attn @ D["wo"].T + D["bo"]
    f = torch.nn.functional.layer_norm(y, (d,), lnw, D["ln_b"], 1e-5)
    h = torch.relu(f @ D["w1"].T + D["b1"])
    keep_h = ((fw["hdrop"] != 0) | (h <= 0)).double() / (1 - p)
    mask_o = (g_o != 0).double() / (1 - p) if p > 0 else torch.ones(M, d, device="cuda", dtype=torch.float64)
    out = f + ((h * keep_h) @ D["w2"].T + D["b2"]) * mask_o
    gA64, gw64 = torch.autograd.grad(out, (attn, lnw), D["g_out"])
    torch.testing.assert_close(g_A.double(), gA64, rtol=1e-5, atol=3e-5)
    torch.testing.assert_close(dw.double(), gw64, rtol=1e-4, atol=1e-4 * float(gw64.abs().max()))


# ---- straight against the oracle (oracle/transformer_oracle.py restates sasrec.py:221-229 + net_blocks.py:21-64), no dropout ------------
@pytest.mark.parametrize("M,d,dff", [(256, 256, 256), (13312, 256, 256), (384, 128, 256)])
def test_chain_kernels_against_the_oracle(M, d, dff):
    """`rt_ffn_fused_*` and `rt_block_tail_*` against the ORACLE's `layer_norm` / `ffn` (fp32 on the host, autograd for the gradients) —
    not against the older kernels: forward outputs, the data gradients down to the attention output, d ln_w / d ln_b."""
    from oracle import transformer_oracle as T
    from rectools_amd import _lib, ops

    lib = _lib.load()
    t = _tail_inputs(M, d, dff, seed=11)
    pl = _tail_planes(t)
    c = {k: v.detach().cpu() for k, v in t.items()}
    params = {"ff.ff_linear_1.weight": c["w1"], "ff.ff_linear_1.bias": c["b1"], "ff.ff_linear_2.weight": c["w2"], "ff.ff_linear_2.bias": c["b2"]}
    attn, lnw, lnb = (c[k].clone().requires_grad_(True) for k in ("attn", "ln_w", "ln_b"))
    y = c["q"] + attn @ c["wo"].T + c["bo"]                       # sasrec.py:224 (q + mha's out-projection)
    y.retain_grad()
    f = T.layer_norm(y, lnw, lnb, 1e-5)                             # :226
    out = f + T.ffn(f, params, "ff.", "relu")                       # :227-229 without dropout
    out.backward(c["g_out"])
    # a hidden unit whose pre-activation is within rounding of zero may sit on the other side of the ReLU kink in the kernel's arithmetic:
    # its whole gradient contribution flips (a discontinuity of the FUNCTION, not an error).  Such rows are compared in the forward only
    z = f.detach() @ c["w1"].T + c["b1"]
    smooth = ~(z.abs() < 2e-5 * float(z.abs().max())).any(1)
    assert float(smooth.float().mean()) > 0.98
    # the whole tail
    fw = _tail_fwd(t, pl, M, d, dff, 0.0, True)
    g_h, g_y, g_A = (torch.empty(M, n, device="cuda") for n in (dff, d, d))
    part = torch.empty(lib.rt_block_tail_partial_floats(M, d), device="cuda")
    dw, db = torch.empty(d, device="cuda"), torch.empty(d, device="cuda")
    ops._c("rt_block_tail_bwd", t["g_out"], fw["hdrop"], fw["y"], fw["mean"], fw["rstd"], t["ln_w"], pl["wop"], pl["w1p"], pl["w2p"], pl["stride"],
           None, g_h, g_y, g_A, part, M, d, dff, 0.0, SEEDS["seed_o"], SEEDS["sid_o"])
    ops._c("rt_layernorm_bwd_reduce", part, M // 64, d, dw, db)
    # the feed-forward half alone, on the oracle's LayerNorm input
    fz = _fused_fwd(dict(t, y=fw["y"]), M, d, dff, 0.0)
    g_h2, g_f2 = torch.empty(M, dff, device="cuda"), torch.empty(M, d, device="cuda")
    ops._c("rt_ffn_fused_bwd", t["g_out"], fz["hdrop"], fz["w1p"], fz["w2p"], fz["stride"], None, g_h2, g_f2, M, d, dff, 0.0,
           SEEDS["seed_o"], SEEDS["sid_o"])
    torch.cuda.synchronize()

    def close(got, want, name, rtol=2e-4, rel=2e-5):
        torch.testing.assert_close(got.cpu(), want, rtol=rtol, atol=rel * float(want.abs().max()), msg=lambda s: f"{name}: {s}")

    close(fw["y"], y.detach(), "y"); close(fw["f"], f.detach(), "LN(y)"); close(fw["out"], out.detach(), "block output")
    close(fz["out"], out.detach(), "feed-forward half: output")
    close(g_y[smooth.cuda()], y.grad[smooth], "d y", rtol=2e-3, rel=2e-4)
    close(g_A[smooth.cuda()], attn.grad[smooth], "d attention output", rtol=2e-3, rel=2e-4)
    close(dw, lnw.grad, "d ln_w", rtol=5e-3, rel=1e-3); close(db, lnb.grad, "d ln_b", rtol=5e-3, rel=1e-3)     # (column sums over ALL rows)
