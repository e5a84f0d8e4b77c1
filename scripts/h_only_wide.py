"""One-plane (h-only) coarse pass beyond the HBM-bound regime: python scripts/h_only_wide.py [c5|c2] ...
Times the (h, m) two-stage path against the one-plane image at the bench's shapes and checks ids / score bits against it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from rectools_amd.rank import DeviceCSR, HipRanker


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best * 1e3


def main():
    which = sys.argv[1:] or ["c2", "c5"]
    g = torch.Generator(device="cuda").manual_seed(0)
    for w in which:
        if w == "c5":
            V, d, U, filt = 5_000_000, 512, 4096, None
        else:
            V, d, U = 26_744, 256, 16_384
            rng = np.random.default_rng(0)
            indptr = np.r_[0, np.cumsum(rng.integers(20, 270, U))].astype(np.int64)
            indices = np.concatenate([np.sort(rng.choice(V, int(n), replace=False)) for n in np.diff(indptr)]).astype(np.int32)
            filt = DeviceCSR(torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda(), (U, V))
        items = torch.empty((V, d), device="cuda")
        for r0 in range(0, V, 500_000):
            items[r0:r0 + 500_000] = torch.randn((min(500_000, V - r0), d), device="cuda", generator=g)
        users = torch.randn((U, d), device="cuda", generator=g)
        ids = np.arange(U)
        res = {}
        for name, env, kw in (("hm  default", {}, dict(two_stage=True)),
                              ("h   upp64", {"RT_TOPK_H_ONLY_MAX_USERS": "1000000", "RT_TOPK_H_ONLY_MIN_BYTES": "0"}, dict(two_stage=True, batch_size=64)),
                              ("h   upp128", {"RT_TOPK_H_ONLY_MAX_USERS": "1000000", "RT_TOPK_H_ONLY_MIN_BYTES": "0"}, dict(two_stage=True, batch_size=128))):
            for k_, v_ in env.items():
                os.environ[k_] = v_
            r = HipRanker("dot", "cuda", users, items, **kw)
            ms = timed(lambda: r.rank_device(ids, 10, filt))   # noqa: B023
            res[name] = r.rank_device(ids, 10, filt)
            print(f"{w} {name:12s} {ms:9.3f} ms  {U / ms * 1e3:10.0f} users/s  {2.0 * U * V * d / ms / 1e9:7.1f} TF  stats {r.two_stage_stats}", flush=True)
            for k_ in env:
                del os.environ[k_]
            del r
            torch.cuda.empty_cache()
        a = res["hm  default"]
        for name in ("h   upp64", "h   upp128"):
            b = res[name]
            print(f"{w} {name}: ids equal {bool(torch.equal(a[0], b[0]))} score bits equal {bool(torch.equal(a[1].view(torch.int32), b[1].view(torch.int32)))}")
        del items, users, res
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
