"""Debug: the sampled loss's table half on the side stream, two backward passes in a row (tests/test_ops_gpu.py::test_sampled_softmax_xcd_sliced_forward)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from rectools_amd import ops

def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))

def grads_of(fn, inputs):
    ins = [t.detach().clone().requires_grad_(True) for t in inputs]
    out = fn(*ins)
    g = rnd(*out.shape, seed=99).to(out.device)
    print("  before backward: keepalive", len(ops._NATIVE_KEEPALIVE), "prep", len(ops._PREP_KEEPALIVE))
    out.backward(g)
    print("  after backward: keepalive", len(ops._NATIVE_KEEPALIVE), "prep", len(ops._PREP_KEEPALIVE))
    return out.detach(), [t.grad for t in ins]

M, d, V, N, t = 2048, 256, 4200, 128, 0.7
g = torch.Generator().manual_seed(4)
sess, table = rnd(M, d, seed=5), rnd(V, d, seed=6)
y = torch.randint(1, V, (M,), generator=g); y[::7] = 0
neg = torch.randint(1, V, (M, N), generator=g)
w = (0.5 + torch.rand(M, generator=g)) * (y != 0)
res = {}
for tag, env in (("sliced", "1"), ("plain", "0"), ("plain2", "0"), ("sliced2", "1")):
    os.environ["RT_LOSS_SLICED"] = env
    print(tag)
    res[tag] = grads_of(lambda s, e: ops.sampled_loss(s, e, y.cuda(), neg.cuda(), w.cuda(), 2, False, t, 0.0)[0].reshape(1), [sess.cuda(), table.cuda()])
    torch.cuda.synchronize()
ref = res["plain2"][1][1]
for tag in res:
    dt = res[tag][1][1]
    print(tag, "max |d_table - plain2|", float((dt - ref).abs().max()), "zero rows", int((dt.abs().sum(1) == 0).sum()), "d_sess diff", float((res[tag][1][0] - res["plain2"][1][0]).abs().max()))
