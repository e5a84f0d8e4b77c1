import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rectools_amd import ops
mode = sys.argv[1]
M, d, V, N = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
g = torch.Generator().manual_seed(0)
sess = (torch.randn(M, d, generator=g) * 0.3).cuda(); table = (torch.randn(V, d, generator=g) * 0.3).cuda()
y = torch.randint(1, V, (M,), generator=g).cuda(); y[::7] = 0
neg = torch.randint(1, V, (M, N), generator=g).cuda(); w = (y != 0).float()
if mode == "eval":
    with torch.no_grad():
        loss, logits = ops.sampled_loss(sess, table, y, neg, w, ops.LOSS_SAMPLED_SOFTMAX, False, 1.0, 0.0)
else:
    sess.requires_grad_(True); table.requires_grad_(True)
    loss, logits = ops.sampled_loss(sess, table, y, neg, w, ops.LOSS_SAMPLED_SOFTMAX, False, 1.0, 0.0)
    torch.cuda.synchronize(); print("forward done", float(loss))
    if mode == "train":
        loss.backward()
torch.cuda.synchronize()
cand = torch.cat([y[:, None], neg], 1)
ref = torch.einsum("mcd,md->mc", table.detach()[cand], sess.detach())
act = y != 0
print(mode, M, d, V, N, "loss", float(loss), "max logit err", float((logits[act] - ref[act]).abs().max()))
