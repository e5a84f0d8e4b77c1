"""Debug aid: does a packed training step read memory it never wrote?  The same batch, parameters and dropout streams, run after the
allocator's cached blocks were filled with zeros and after they were filled with a poison value: every gradient must agree (up to the
item table's atomics-order noise).   python scripts/debug/poison_repro.py [L] [p] [B]"""
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

L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
p_drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rng = np.random.default_rng(3)
n_users, n_items, n = 260, 180, 9000
df = pd.DataFrame({"user_id": rng.integers(0, n_users, n), "item_id": rng.integers(0, n_items, n) + 100, "weight": 1.0,
                   "datetime": pd.to_datetime("2022-01-01") + pd.to_timedelta(rng.integers(0, 500_000, n), unit="m")})
ds = Dataset.construct(df)
m = SASRecModel(n_factors=256, n_heads=4, session_max_len=L, n_blocks=4, n_negatives=16, batch_size=B, loss="sampled_softmax", dropout_rate=p_drop,
                use_pos_emb=False, use_key_padding_mask=True, lr=0.004, seed=11, epochs=1)
m._build_model_from_dataset(ds)
loop = m.training_loop()
m.lightning_model.train()
loop.begin_epoch(0)
batch = loop._cut_batch()
print("rows", int(batch["x"].shape[0]), "session rows", batch["n_rows"], "cu_attn" in batch)


def fill(value):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    big = [torch.full((64 << 20,), value, dtype=torch.float32, device="cuda") for _ in range(6)]      # 1.5 GB of cached blocks
    del big
    torch.cuda.synchronize()


runs = []
for value in (0.0, 1.0e3, float("nan")):
    fill(value)
    ops.RNG.__init__(0)
    ops.RNG.next_step()
    loop.opt.zero_grad()
    loss = loop.lm.training_loss_packed(batch)
    loss.backward()
    ops.join_side_streams()
    torch.cuda.synchronize()
    runs.append((value, float(loss), {k: p.grad.detach().clone() for k, p in m.torch_model.named_parameters()}))
base = runs[0]
for value, loss, grads in runs[1:]:
    print(f"poison {value}: loss {loss!r} (zeros: {base[1]!r})")
    for k, g in grads.items():
        a = base[2][k]
        bad = int((a != g).sum()) if not torch.isnan(g).any() else -1
        if bad and "ids_emb" not in k:
            print(f"   {k}: {bad} of {a.numel()} elements differ (max {float((a - g).abs().max()):.2e}; -1 = NaN present)")
print("done")
