import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rectools_amd import ops
torch.manual_seed(0)
for M, N, K in ((25600, 256, 256), (12800, 256, 256), (6400, 256, 256)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.2; b = torch.randn(N, device="cuda")
    y = ops.linear(x, w, b)
    ref = (x.double() @ w.double().T + b.double()).float()
    err = (y - ref).abs()
    bad = (err > 1e-3).nonzero()
    print(M, N, K, "max err", float(err.max()), "bad", bad.shape[0], "first bad", bad[:5].tolist(), "rows with bad", torch.unique(bad[:, 0])[:10].tolist() if bad.numel() else [])
    # backward
    x.requires_grad_(True); wp = w.clone().requires_grad_(True)
    g = torch.randn(M, N, device="cuda")
    y = ops.linear(x, wp, b); y.backward(g); torch.cuda.synchronize()
    dx_ref = (g.double() @ w.double()).float(); dw_ref = (g.double().T @ x.detach().double()).float(); db_ref = g.double().sum(0).float()
    for name, got, ref2 in (("dx", x.grad, dx_ref), ("dw", wp.grad, dw_ref)):
        e = (got - ref2).abs(); print("   ", name, "max err", float(e.max()), "scale", float(ref2.abs().max()), "bad", int((e > 1e-3 * ref2.abs().max()).sum()))
