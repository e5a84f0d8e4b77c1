"""Attention fwd/bwd timing (HIP events) on a training shape; default = C2 (B=128, H=4, L=200, hd=64, causal, dropout 0.2).

    python scripts/attn_bench.py [--B 128 --H 4 --L 200 --d 256] [--hstu]        # RT_ATTN_IMPL=ring|res|stream selects the family
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rectools_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--H", type=int, default=4)
ap.add_argument("--L", type=int, default=200)
ap.add_argument("--d", type=int, default=256)
ap.add_argument("--hstu", action="store_true")
ap.add_argument("--n", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
B, H, L, d = args.B, args.H, args.L, args.d
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(B * L, d, generator=g).to(dev).requires_grad_(True) for _ in range(3))
ids = torch.randint(1, 1000, (B, L), generator=g).to(dev)
go = torch.randn(B * L, d, generator=g).to(dev)
tag = os.environ.get("TAG", os.environ.get("RT_ATTN_IMPL", "auto"))


def t(fn, n=args.n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


pairs = (L // 32) * (L // 32 + 1) / 2 + (L % 32 > 0) * (L // 32 + 1)          # causal 32x32 tile pairs per (b, h)
fl_pair = 4.0 * 32 * 32 * (d // H)                                            # S + PV flops of one pair
if args.hstu:
    ts = torch.cumsum(torch.randint(1, 100000, (B, L + 1), generator=g), 1).to(dev)
    thr = ops.hstu_time_thresholds().to(dev)
    tw = (torch.randn(129, generator=g) * 0.1).to(dev).requires_grad_(True)
    pw = (torch.randn(2 * L - 1, generator=g) * 0.1).to(dev).requires_grad_(True)
    fwd_fn = lambda: ops.hstu_attn(q, k, v, tw, pw, ids, ts, thr, B, H, L)   # noqa: E731
    variants = (("hstu", fwd_fn),)
else:
    variants = tuple((f"p={p}", (lambda p=p: ops.mha(q, k, v, ids, B, H, L, True, False, p))) for p in (0.2, 0.0))
for name, fwd_fn in variants:
    fwd = t(fwd_fn)

    def fb():
        fwd_fn().backward(go)

    both = t(fb)
    tf = lambda us, mult: B * H * pairs * fl_pair * mult / (us * 1e-6) / 1e12   # noqa: E731
    print(f"[{tag}] B{B} H{H} L{L} hd{d // H} {name}: fwd {fwd:.1f} us ({tf(fwd, 1):.0f} TF executed)  fwd+bwd {both:.1f} us  "
          f"(bwd ~{both - fwd:.1f} us, {tf(both - fwd, 3.5):.0f} TF executed)")
