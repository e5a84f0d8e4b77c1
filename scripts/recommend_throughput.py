"""End-to-end SASRecModel.recommend() (dataset transform + encode + rank + frame) for all users of a synthetic dataset."""
import os, sys, time, cProfile, pstats, io
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
model.fit(ds)
users = ds.user_id_map.external_ids
model.recommend(users[:256], ds, k=10, filter_viewed=True)   # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter(); reco = model.recommend(users, ds, k=10, filter_viewed=True); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"recommend {len(users)} users: {t1 - t0:.3f} s -> {len(users) / (t1 - t0):.0f} users/s, {len(reco)} rows")
pr = cProfile.Profile(); pr.enable(); model.recommend(users, ds, k=10, filter_viewed=True); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(16); print(s.getvalue()[:4200])
