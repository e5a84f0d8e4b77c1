"""Debug aid: which gradients of a packed SASRec step differ between two backward passes over the SAME batch, parameters and dropout
streams (run-to-run reproducibility of the kernels' reductions).   python scripts/debug/grad_repro.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pandas as pd
import torch

os.environ["RT_NATIVE_STEP"] = "0"
from rectools_amd import ops
from rectools_amd.dataset import Dataset
from rectools_amd.models import SASRecModel

rng = np.random.default_rng(3)
n_users, n_items, n = 600, 500, 40000
df = pd.DataFrame({"user_id": rng.integers(0, n_users, n), "item_id": rng.integers(0, n_items, n) + 100, "weight": 1.0,
                   "datetime": pd.to_datetime("2022-01-01") + pd.to_timedelta(rng.integers(0, 500_000, n), unit="m")})
ds = Dataset.construct(df)
for p_drop in (0.0, 0.2):
    m = SASRecModel(n_factors=128, n_blocks=2, n_heads=2, session_max_len=64, lr=0.004, batch_size=128, dropout_rate=p_drop, loss="sampled_softmax",
                    n_negatives=32, seed=11, epochs=1)
    m._build_model_from_dataset(ds)
    loop = m.training_loop()
    m.lightning_model.train()
    loop.begin_epoch(0)
    batch = loop._cut_batch()
    runs = []
    for _ in range(3):
        ops.RNG.__init__(0)
        ops.RNG.next_step()
        loop.opt.zero_grad()
        loss = loop.lm.training_loss_packed(batch)
        loss.backward()
        ops.join_side_streams()
        torch.cuda.synchronize()
        runs.append((float(loss), {k: p.grad.detach().clone() for k, p in m.torch_model.named_parameters()}))
    print(f"dropout {p_drop}: losses {[r[0] for r in runs]}")
    for k in runs[0][1]:
        a, b, c = (r[1][k] for r in runs)
        nd = int((a != b).sum()) + int((a != c).sum())
        if nd:
            print(f"  {k}: {int((a != b).sum())} / {int((a != c).sum())} of {a.numel()} elements differ between passes, max rel {float(((a - b).abs() / (a.abs() + 1e-30)).max()):.1e}")
    print("  (every other gradient: identical bits)")
