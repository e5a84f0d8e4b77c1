"""recommend() through the public API at the C2 shape (16,384 users of a 16,384-user dataset, 26,744 items), best of 3 calls: A/B knob runs
   RT_ENCODE_SESSIONS=4096 python scripts/recommend_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pandas as pd, torch
from rectools_amd import synth
from rectools_amd.dataset import Dataset, Columns
from rectools_amd.models import SASRecModel

n_users, V = 16384, synth.ML_20M["n_items"]
u, it, ts = synth.gen_interactions(n_users, V, mean_len=144.0, min_len=20, max_len=2000, seed=3)
df = pd.DataFrame({Columns.User: u, Columns.Item: it, Columns.Weight: 1.0, Columns.Datetime: pd.to_datetime(ts, unit="s")})
ds = Dataset.construct(df)
model = SASRecModel(n_factors=256, n_blocks=2, n_heads=4, session_max_len=200, dropout_rate=0.2, loss="sampled_softmax",
                    n_negatives=128, batch_size=128, epochs=1, lr=1e-3, verbose=0, deterministic=False)
model._build_model_from_dataset(ds)
model.is_fitted = True
users = ds.user_id_map.external_ids
for _ in range(2):
    model.recommend(users, ds, k=10, filter_viewed=True)
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reco = model.recommend(users, ds, k=10, filter_viewed=True)
    best = min(best, time.perf_counter() - t0)
model.phase_log = {}
model.recommend(users, ds, k=10, filter_viewed=True)
print(f"RT_ENCODE_SESSIONS={os.environ.get('RT_ENCODE_SESSIONS', '1024')}: {len(users) / best:.0f} users/s ({best * 1e3:.2f} ms), phases ms",
      {k: round(v * 1e3, 2) for k, v in model.phase_log.items()})
