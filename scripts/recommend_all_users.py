"""SASRecModel.recommend() for ALL users of the ML-20M-shaped synthetic dataset (138,493 users, 26,744 items) at several
encoder launch sizes (RT_ENCODE_SESSIONS): median of 5 whole calls each, phases of one instrumented call.
   python scripts/recommend_all_users.py [sessions per encoder launch ... | 0 = the library's default]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd
import torch

from rectools_amd import synth
from rectools_amd.dataset import Columns, Dataset
from rectools_amd.models import SASRecModel

sizes = [int(a) for a in sys.argv[1:]] or [4096]
u, it, ts = synth.gen_interactions(synth.ML_20M["n_users"], synth.ML_20M["n_items"], mean_len=144.0, min_len=20, max_len=9254, seed=0)
ds = Dataset.construct(pd.DataFrame({Columns.User: u, Columns.Item: it, Columns.Weight: 1.0, Columns.Datetime: pd.to_datetime(ts, unit="s")}))
model = SASRecModel(n_factors=256, n_blocks=2, n_heads=4, session_max_len=200, dropout_rate=0.2, loss="sampled_softmax", n_negatives=128,
                    batch_size=128, epochs=1, seed=32)
model._build_model_from_dataset(ds)
model.is_fitted = True
users = np.asarray(ds.user_id_map.external_ids)
model.recommend(users[:2048], ds, k=10, filter_viewed=True)
for size in sizes:
    if size > 0:
        os.environ["RT_ENCODE_SESSIONS"] = str(size)
    else:
        os.environ.pop("RT_ENCODE_SESSIONS", None)      # 0: the library's default
    model.recommend(users, ds, k=10, filter_viewed=True)
    times = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.recommend(users, ds, k=10, filter_viewed=True)
        times.append(time.perf_counter() - t0)
    model.phase_log = {}
    model.recommend(users, ds, k=10, filter_viewed=True)
    phases = {k: round(v * 1e3, 1) for k, v in model.phase_log.items()}
    model.phase_log = None
    print(f"sessions per launch {size}: median {np.median(times) * 1e3:.1f} ms = {len(users) / np.median(times):.0f} users/s "
          f"(best {len(users) / min(times):.0f}); phases ms {phases}; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
