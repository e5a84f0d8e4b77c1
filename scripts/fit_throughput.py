"""End-to-end SASRecModel.fit() throughput (host collate + H2D + step) vs the resident-batch step loop of bench.py."""
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
t0 = time.perf_counter(); model.fit(ds); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"fit epoch 1 (incl. dataset processing + first-call warm-up): {t1 - t0:.2f} s")
t0 = time.perf_counter(); model.fit_partial(ds, 2, 2); torch.cuda.synchronize(); t1 = time.perf_counter()
n_seq = len(model.data_preparator.train_store()) if hasattr(model.data_preparator, "train_store") else n_users
print(f"fit_partial 2 epochs: {t1 - t0:.2f} s -> {2 * n_seq / (t1 - t0):.0f} seqs/s ({n_seq} sequences/epoch)")
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable(); model.fit_partial(ds, 1, 1); torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
