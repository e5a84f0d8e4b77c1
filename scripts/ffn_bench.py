"""K7f (rt_ffn_fused_fwd / _bwd) against the five-launch sequences it replaces, at the packed C2 row counts.
   python scripts/ffn_bench.py [M,M,...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

import test_ffn_fused_gpu as T
from rectools_amd import ops


def time_it(fn, n=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


Ms = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [13312, 16384, 18432]
FUSED_ONLY = len(sys.argv) > 2 and sys.argv[2] == "fused"
d = dff = 256
for p in ((0.2,) if FUSED_ONLY else (0.2, 0.0)):
    for M in Ms:
        t = T._inputs(M, d, dff)
        fz = T._fused_fwd(t, M, d, dff, p)
        un = T._unfused_fwd(t, fz, M, d, dff, p)
        g_o, g_h, g_f = (torch.empty(M, n, device="cuda") for n in (d, dff, d))
        g_hd = torch.empty(M, dff, device="cuda")

        def fused_bwd():
            ops._c("rt_ffn_fused_bwd", t["g_out"], fz["hdrop"], fz["w1p"], fz["w2p"], fz["stride"], g_o, g_h, g_f, M, d, dff, p,
                   T.SEEDS["seed_o"], T.SEEDS["sid_o"])

        def unfused_bwd():
            r_o = t["g_out"]
            if p > 0:
                ops._c("rt_act_dropout_bwd", t["g_out"], t["g_out"], 0, p, T.SEEDS["seed_o"], T.SEEDS["sid_o"], M * d, g_o)
                r_o = g_o
            T._wp(r_o, fz["w2p"], fz["stride"], dff, g_hd, M, dff, d, 1)
            ops._c("rt_act_dropout_bwd", g_hd, un["h"], 1, p, T.SEEDS["seed_h"], T.SEEDS["sid_h"], M * dff, g_h)
            T._wp(g_h, fz["w1p"], fz["stride"], d, g_f, M, d, dff, 1, R=t["g_out"])

        fl = 4.0 * M * d * dff
        S = T.SEEDS
        h_, o_ = torch.empty(M, dff, device="cuda"), torch.empty(M, d, device="cuda")

        def fused_fwd():     # buffers allocated once: the loop times the launch alone
            ops._c("rt_ffn_fused_fwd", t["y"], t["ln_w"], t["ln_b"], 1e-5, fz["f"], fz["mean"], fz["rstd"], fz["w1p"], fz["w2p"], fz["stride"],
                   t["b1"], t["b2"], fz["hdrop"], fz["out"], M, d, dff, p, S["seed_h"], S["sid_h"], S["seed_o"], S["sid_o"])

        def unfused_fwd():
            ops._c("rt_layernorm_fwd", t["y"], t["ln_w"], t["ln_b"], 1e-5, M, d, un["f"], un["mean"], un["rstd"])
            T._wp(un["f"], fz["w1p"], fz["stride"], d, h_, M, dff, d, 0, bias=t["b1"], relu=1)
            ops._c("rt_act_dropout_fwd", h_, 0, p, S["seed_h"], S["sid_h"], M * dff, None, un["hdrop"])
            T._wp(un["hdrop"], fz["w2p"], fz["stride"], dff, o_, M, d, dff, 0, bias=t["b2"])
            ops._c("rt_act_dropout_fwd", o_, 0, p, S["seed_o"], S["sid_o"], M * d, un["f"], un["out"])

        r = {"fwd fused": time_it(fused_fwd), "bwd fused": time_it(fused_bwd)}
        if not FUSED_ONLY:
            r.update({"fwd 5 launches": time_it(unfused_fwd), "bwd 4 launches": time_it(unfused_bwd)})
        print(f"M={M} p={p}: " + "  ".join(f"{k} {v:.1f} us ({fl / v / 1e6:.0f} TF)" for k, v in r.items()))
