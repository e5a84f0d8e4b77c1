"""Is the sampled-loss gather bound by L2 misses?  Same positions / negatives per position, catalogs that do and do not fit one
XCD's 4 MB L2 (V x 256 x 4 B): 2,000 (2 MB), 3,500 (3.6 MB), 7,000, 26,744 (27 MB), 200,000 (205 MB)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rectools_amd import ops

M, d, N = int(os.environ.get('PROBE_M', 25600)), 256, 128
g = torch.Generator().manual_seed(0)
sess = (torch.randn(M, d, generator=g) * 0.3).cuda().requires_grad_(True)
for V in tuple(int(v) for v in os.environ.get('PROBE_V', '2000,3500,7000,26744,200000').split(',')):
    table = (torch.randn(V, d, generator=g) * 0.3).cuda().requires_grad_(True)
    y = torch.randint(1, V, (M,), generator=g).cuda()
    y[torch.rand(M, generator=g).cuda() < 0.28] = 0
    neg = torch.randint(1, V, (M, N), generator=g).cuda()
    w = (y != 0).float()

    def step():
        sess.grad = None; table.grad = None
        loss, _ = ops.sampled_loss(sess, table, y, neg, w, ops.LOSS_SAMPLED_SOFTMAX, False, 1.0, 0.0)
        loss.backward()

    for _ in range(3):
        step()
    ops.start_timing(single_stream=True)
    for _ in range(10):
        step()
    rec = ops.stop_timing()
    print(f"V={V:7d} ({V * d * 4 / 1e6:6.1f} MB):", {k: round(sum(t for t, _ in v) / 10 * 1e3, 1) for k, v in rec.items()}, "us per call")
