"""Would two launches by session length pay for the packed attention (rt_attention_v2.hip)?  One C2-like batch (128 sessions, ML-20M-shaped
lengths, longest first, 4 heads of 64): the three kernels in ONE launch each (154 KB images: one workgroup per CU) against a long class
(> T rows, window-sized images) + a short class (<= T rows, images of T rows: two workgroups per CU).
   python scripts/attn_split_probe.py [T=96]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rectools_amd import _lib, ops, synth

T = int(sys.argv[1]) if len(sys.argv) > 1 else 96
B, H, hd, L = 128, 4, 64, 200
d = H * hd
rng = np.random.default_rng(0)
lens = np.clip(synth.gen_lengths(B, 144.0, 20, 9254, rng) - 1, 1, L)
lens = np.sort(lens)[::-1].copy()
cu_h = np.zeros(B + 1, np.int64); cu_h[1:] = np.cumsum(lens)
n = int(cu_h[-1]); rows = (n + 127) // 128 * 128
n_long = int((lens > T).sum())
print(f"rows {n} (padded {rows}), sessions > {T}: {n_long} of {B}; sum n^2 long {int((lens[:n_long] ** 2).sum())} short {int((lens[n_long:] ** 2).sum())}")
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)   # noqa: E731
q, kv, do = r(rows, d), r(rows, 2 * d), r(rows, d)
bkv = r(2 * d) * 0.1
cu = torch.from_numpy(cu_h).cuda()
o, lse = torch.empty(rows, d, device="cuda"), torch.empty(rows, H, device="cuda")
dq, dkv, delta, part = torch.empty(rows, d, device="cuda"), torch.empty(rows, 2 * d, device="cuda"), torch.empty(rows, H, device="cuda"), torch.empty(B, d, device="cuda")
p, seed = 0.2, 1234


def fwd(c, nb, max_len, sd):
    ops._c("rt_mha_varlen_train_fwd", q, d, kv, 2 * d, kv[:, d:], 2 * d, c, bkv, bkv[d:], nb, H, hd, max_len, L, p, sd, o, d, lse)


def bwd(c, nb, max_len, sd, prt):
    ops._c("rt_mha_varlen_bwd", q, d, kv, 2 * d, kv[:, d:], 2 * d, o, d, do, d, lse, c, bkv, bkv[d:], nb, H, hd, max_len, L, p, sd, dq, d, dkv, 2 * d,
           dkv[:, d:], 2 * d, delta, prt)


def time_it(fn, reps=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


cu2 = cu[n_long:]
res = {
    "fwd one launch": time_it(lambda: fwd(cu, B, L, seed)),
    "fwd long + short": time_it(lambda: (fwd(cu, n_long, L, seed), fwd(cu2, B - n_long, T, seed + 1))),
    "bwd one launch": time_it(lambda: bwd(cu, B, L, seed, part)),
    "bwd long + short": time_it(lambda: (bwd(cu, n_long, L, seed, part), bwd(cu2, B - n_long, T, seed + 1, part[n_long:]))),
}
print("  ".join(f"{k} {v:.1f} us" for k, v in res.items()))
