"""rocprofv3 target: `model.recommend()` of the C2 model for 16,384 users, 5 calls after a warm-up (kernel trace of the recommend path alone).
    (cd /tmp && rocprofv3 --kernel-trace -d <out> -o p -- python scripts/recommend_trace.py)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from rectools_amd.models import SASRecModel

ds = bench.make_ml20m_dataset()
model = SASRecModel(n_factors=256, n_blocks=2, n_heads=4, session_max_len=200, dropout_rate=0.2, loss="sampled_softmax",
                    n_negatives=128, batch_size=128, lr=1e-3, epochs=1, seed=32)
model._build_model_from_dataset(ds)
model.is_fitted = True
users = np.asarray(ds.user_id_map.external_ids)[:16384]
model.recommend(users=users[:2048], dataset=ds, k=10, filter_viewed=True)
model.recommend(users=users, dataset=ds, k=10, filter_viewed=True)
torch.cuda.synchronize()
import time

t0 = time.perf_counter()
for _ in range(5):
    model.recommend(users=users, dataset=ds, k=10, filter_viewed=True)
torch.cuda.synchronize()
print(f"5 x recommend(16384 users): {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call")
