"""Where does `model.recommend()` spend its time?  ML-20M-shaped SASRec (C2), 16,384 users, k = 10, filter_viewed."""
import os
import sys
import time

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
torch.cuda.synchronize()

import cProfile
import pstats

pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
reco = model.recommend(users=users, dataset=ds, k=10, filter_viewed=True)
torch.cuda.synchronize()
pr.disable()
print(f"recommend(): {time.perf_counter() - t0:.4f} s for {len(users)} users")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

# the encoder alone, with device synchronisation around it
from rectools_amd.data_preparator import DeviceSequenceStore  # noqa: E402
