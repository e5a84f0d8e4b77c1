"""Two-stage vs exact top-k on the 5M x 512 catalog (HIP events around HipRanker.rank_device; k = 10, dot)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectools_amd.rank import HipRanker  # noqa: E402

V, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5_000_000, 512)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
items = torch.randn(V, d, device=dev, generator=g)
for n_users, batch in ((32, 32), (1024, 128)):
    users = torch.randn(n_users, d, device=dev, generator=g)
    res = {}
    for mode in (False, True):
        r = HipRanker("dot", dev, users, items, batch_size=batch, two_stage=mode)
        ids = np.arange(n_users)
        out = r.rank_device(ids, 10)          # warm-up (builds the bf16 image once)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            out = r.rank_device(ids, 10)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res[mode] = (ms, out[0].clone(), out[1].clone(), dict(r.two_stage_stats))
        del r
    same = torch.equal(res[False][1], res[True][1])
    err = float((res[False][2] - res[True][2]).abs().max())
    print(f"V={V} d={d} users={n_users}: exact {res[False][0]:.2f} ms, two-stage {res[True][0]:.2f} ms ({res[False][0]/res[True][0]:.1f}x), "
          f"ids equal {same}, max |score diff| {err:.2e}, stats {res[True][3]}", flush=True)
