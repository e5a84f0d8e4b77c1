import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rectools_amd import ops
dev = torch.device("cuda:0")
B, H, L, d = 128, 4, 200, 256
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(B * L, d, generator=g).to(dev) for _ in range(3))
ids = torch.randint(1, 1000, (B, L), generator=g).to(dev)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    print(os.environ.get("TAG", ""), " ".join(f"p={p}: {t(lambda: ops.mha(q, k, v, ids, B, H, L, True, False, p)):.1f} us" for p in (0.2, 0.0)))
