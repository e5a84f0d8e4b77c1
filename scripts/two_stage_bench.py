"""Two-stage vs single-stage top-k at the shapes the bench quotes: python scripts/two_stage_bench.py [c5|c2] ...
c5: 5,000,000 x 512 catalog, 4096 users; c2: 26,744 x 256 catalog, 16,384 users, viewed filter (~144 items per user)."""
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
        for name, kw in (("single upp64", dict(two_stage=False, batch_size=64)), ("single upp128", dict(two_stage=False, batch_size=128)),
                         ("two-stage upp64", dict(two_stage=True, batch_size=64)), ("two-stage upp128", dict(two_stage=True, batch_size=128))):
            r = HipRanker("dot", "cuda", users, items, **kw)
            ms = timed(lambda: r.rank_device(ids, 10, filt))   # noqa: B023
            out = r.rank_device(ids, 10, filt)
            res[name] = out
            print(f"{w} {name:18s} {ms:9.3f} ms  {U / ms * 1e3:10.0f} users/s  {2.0 * U * V * d / ms / 1e9:7.1f} TF  stats {r.two_stage_stats}", flush=True)
        if w == "c2":     # what the selection slow path costs: no viewed filter, and k = 1 (few inserts)
            for name, kw in (("single upp64", dict(two_stage=False, batch_size=64)), ("two-stage upp64", dict(two_stage=True, batch_size=64))):
                r = HipRanker("dot", "cuda", users, items, **kw)
                print(f"c2 {name:18s} no filter {timed(lambda: r.rank_device(ids, 10, None)):7.3f} ms   k=1 with filter "   # noqa: B023
                      f"{timed(lambda: r.rank_device(ids, 1, filt)):7.3f} ms", flush=True)   # noqa: B023
        a, b = res["single upp64"], res["two-stage upp128"]
        print(f"{w} ids equal {bool(torch.equal(a[0], b[0]))} score bits equal {bool(torch.equal(a[1].view(torch.int32), b[1].view(torch.int32)))}")
        del items, users, res
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
