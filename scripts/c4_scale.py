"""BASELINE.json configs[3] nearer its stated size (10 M users / 1 M items): the HOST path that has to scale — Dataset.construct,
`process_dataset_train` (data_preparator.py:39-99, :214-284 of the reference), the session store and its upload, one epoch's host
bookkeeping (`_TrainLoop.begin_epoch`) — timed at N users x 1 M items, then a few product steps of HSTU on that store.

    python scripts/c4_scale.py [n_users=2000000] [mean_len=40] [steps=20]      (GPU box; on a CPU-only box it stops before the upload)
"""
import json
import os
import resource
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd
import torch

from rectools_amd import synth
from rectools_amd.dataset import Dataset

n_users = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
mean_len = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n_items, L = 1_000_000, 512
out = {"n_users": n_users, "n_items": n_items, "mean_len": mean_len, "session_max_len": L}
t0 = time.perf_counter()
u, it, ts = synth.gen_interactions(n_users, n_items, mean_len=mean_len, min_len=5, max_len=3000, seed=0, clip_len=L + 1)
out["interactions"] = int(len(u))
out["gen_s"] = round(time.perf_counter() - t0, 2)
t0 = time.perf_counter()
df = pd.DataFrame({"user_id": u, "item_id": it, "weight": 1.0, "datetime": pd.to_datetime(ts, unit="s")})
ds = Dataset.construct(df)
out["dataset_construct_s"] = round(time.perf_counter() - t0, 2)
del df, u, it, ts

from rectools_amd.models import HSTUModel

model = HSTUModel(n_factors=256, n_blocks=2, n_heads=4, session_max_len=L, dropout_rate=0.2, loss="sampled_softmax", n_negatives=128,
                  batch_size=128, lr=1e-3, epochs=1, seed=32, relative_time_attention=True, relative_pos_attention=True,
                  lightning_module_kwargs={"logits_t": 0.05})
t0 = time.perf_counter()
model.data_preparator.process_dataset_train(ds)
out["process_dataset_train_s"] = round(time.perf_counter() - t0, 2)
t0 = time.perf_counter()
store = model.data_preparator.train_store()
out["train_store_s"] = round(time.perf_counter() - t0, 2)
out["sessions"] = int(len(store))
out["store_bytes"] = int(sum(np.asarray(getattr(store, a)).nbytes for a in ("offsets", "items", "weights") if hasattr(store, a))
                         + (np.asarray(store.unix_ts).nbytes if getattr(store, "unix_ts", None) is not None else 0))
out["host_peak_rss_gb"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 2)
if not torch.cuda.is_available():
    print(json.dumps(out))
    sys.exit(0)

torch.cuda.set_device(0)
t0 = time.perf_counter()
model._build_model_from_dataset(ds)        # (processes the dataset once more: what fit() does)
out["build_model_from_dataset_s"] = round(time.perf_counter() - t0, 2)
torch.cuda.synchronize()
t0 = time.perf_counter()
loop = model.training_loop()
torch.cuda.synchronize()
out["device_store_upload_s"] = round(time.perf_counter() - t0, 3)
ds_ = loop.dstore
out["device_store_bytes"] = int(sum(t.numel() * t.element_size() for t in (ds_.offsets, ds_.items, ds_.weights, ds_.unix_ts) if t is not None))
model.lightning_model.train()
t0 = time.perf_counter()
loop.begin_epoch(0)
torch.cuda.synchronize()
out["begin_epoch_s"] = round(time.perf_counter() - t0, 3)
out["steps_per_epoch"] = int(loop.batches_left())
for _ in range(5):
    loop.step()
torch.cuda.synchronize()
s0 = loop.sequences_done
t0 = time.perf_counter()
for _ in range(steps):
    loop.step()
torch.cuda.synchronize()
el = time.perf_counter() - t0
out["train_seqs_per_s"] = round((loop.sequences_done - s0) / el, 1)
out["ms_per_step"] = round(el / steps * 1e3, 3)
out["packed"] = bool(loop.packed)
out["hbm_allocated_gb"] = round(torch.cuda.max_memory_allocated() / 1e9, 2)
out["host_peak_rss_gb"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 2)
print(json.dumps(out))
