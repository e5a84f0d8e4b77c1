"""wgrad (dW = dy^T x, split-K) timing vs number of splits; RT_GEMM_IMPL selects the DMA ring depth."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rectools_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
M, N, K = 25600, 256, 256
x, dy = torch.randn(M, K, generator=g).to(dev), torch.randn(M, N, generator=g).to(dev)
dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("impl", os.environ.get("RT_GEMM_IMPL", "default"))
for sp in (16, 32, 48, 64, 100, 128, 200):
    a = t(lambda: ops._gemm(dy, N, 0, x, K, 0, dw, K, None, None, 0, N, K, M, 0, sp))
    b = t(lambda: ops._gemm(dy, N, 0, x, K, 0, dw, K, None, None, 0, N, K, M, 0, sp, db))
    print(f"  splits={sp:4d}  wgrad {a:6.1f} us   with rowsum {b:6.1f} us")
