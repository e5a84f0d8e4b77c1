"""Attention fwd/bwd timing on the C2 training shape (B=128, H=4, L=200, hd=64, causal, dropout 0.2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rectools_amd import ops
dev = torch.device("cuda:0")
B, H, L, d = 128, 4, 200, 256
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(B * L, d, generator=g).to(dev).requires_grad_(True) for _ in range(3))
ids = torch.randint(1, 1000, (B, L), generator=g).to(dev)
go = torch.randn(B * L, d, generator=g).to(dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for p in (0.2, 0.0):
    fwd = t(lambda: ops.mha(q, k, v, ids, B, H, L, True, False, p))
    def fb():
        o = ops.mha(q, k, v, ids, B, H, L, True, False, p); o.backward(go)
    both = t(fb)
    print(f"{os.environ.get('TAG','')} p={p}: fwd {fwd:.1f} us  fwd+bwd {both:.1f} us  (bwd ~{both - fwd:.1f})")
