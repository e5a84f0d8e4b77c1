import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rectools_amd.rank import HipRanker
g = torch.Generator(device="cuda").manual_seed(0)
V, d = 5_000_000, 512
items = torch.empty((V, d), device="cuda")
for r0 in range(0, V, 500_000):
    items[r0:r0 + 500_000] = torch.randn((500_000, d), device="cuda", generator=g)
def timed(fn, n=3):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(n):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best * 1e3
for U in (32, 64, 96, 128, 256):
    users = torch.randn((U, d), device="cuda", generator=g)
    ids = np.arange(U)
    out = []
    for name, kw in (("single", dict(two_stage=False)), ("two-stage", dict(two_stage=True))):
        r = HipRanker("dot", "cuda", users, items, **kw)
        out.append(f"{name} {timed(lambda: r.rank_device(ids, 10)):8.3f} ms")
    print(U, "users:", "  ".join(out), flush=True)
