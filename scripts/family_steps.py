"""Full-size training epochs of the other model families of BASELINE.json (configs 3-5) through the public API: a
robustness + throughput check beside the parity tests, which run at small sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch
from rectools_amd import nn as hnn, synth
from rectools_amd.dataset import Dataset, Columns
from rectools_amd.models import BERT4RecModel, HSTUModel, SASRecModel

def dataset(n_users, n_items, mean_len, max_len, seed, features=False):
    u, it, ts = synth.gen_interactions(n_users, n_items, mean_len=mean_len, min_len=20, max_len=max_len, seed=seed)
    df = pd.DataFrame({Columns.User: u, Columns.Item: it, Columns.Weight: 1.0, Columns.Datetime: pd.to_datetime(ts, unit="s")})
    if not features:
        return Dataset.construct(df)
    # two categorical item features with Zipf-popular values: "genre" (3 per item out of 40) and "studio" (1 out of 1000)
    rng = np.random.default_rng(seed)
    items = np.unique(it)
    pg, ps = 1.0 / np.arange(1, 41), 1.0 / np.arange(1, 1001)
    genre = rng.choice(40, (len(items), 3), p=pg / pg.sum())
    studio = rng.choice(1000, len(items), p=ps / ps.sum())
    feats = pd.concat([pd.DataFrame({"id": np.repeat(items, 3), "feature": "genre", "value": genre.reshape(-1)}),
                       pd.DataFrame({"id": items, "feature": "studio", "value": studio})])
    return Dataset.construct(df, item_features_df=feats, cat_item_features=["genre", "studio"])

cases = [
    ("SASRec d256 nb2 L200 sampled_softmax N=128, ids only (config 2)", lambda: SASRecModel(n_factors=256, n_blocks=2, n_heads=4, session_max_len=200, loss="sampled_softmax", n_negatives=128, batch_size=128, epochs=1, deterministic=False, item_net_block_types=(hnn.IdEmbeddingsItemNet,)), 8192, 26744, 144.0, 2000),
    ("SASRec d256 nb2 L200 sampled_softmax N=128, ids + catfeatures item net (config 2 + item features)", lambda: SASRecModel(n_factors=256, n_blocks=2, n_heads=4, session_max_len=200, loss="sampled_softmax", n_negatives=128, batch_size=128, epochs=1, deterministic=False), 8192, 26744, 144.0, 2000),
    ("BERT4Rec d256 nb2 L200 softmax (config 3)", lambda: BERT4RecModel(n_factors=256, n_blocks=2, n_heads=4, session_max_len=200, loss="softmax", batch_size=128, epochs=1, deterministic=False), 4096, 26744, 144.0, 2000),
    ("HSTU d256 nb2 L512 rel time+pos, sampled_softmax N=128 (config 4 shape)", lambda: HSTUModel(n_factors=256, n_blocks=2, n_heads=4, session_max_len=512, loss="sampled_softmax", n_negatives=128, batch_size=64, epochs=1, deterministic=False), 2048, 100_000, 300.0, 3000),
    ("eSASRec: SASRec + LiGR d512 nb2 L200 sampled_softmax N=128 (config 5 train)", lambda: SASRecModel(n_factors=512, n_blocks=2, n_heads=8, session_max_len=200, loss="sampled_softmax", n_negatives=128, batch_size=128, epochs=1, deterministic=False, transformer_layers_type=hnn.LiGRLayers), 4096, 200_000, 144.0, 2000),
]
only = sys.argv[1:]
for name, make, n_users, n_items, mean_len, max_len in cases:
    if only and not any(o.lower() in name.lower() for o in only):
        continue
    ds = dataset(n_users, n_items, mean_len, max_len, seed=11, features="catfeatures" in name)
    model = make()
    model.fit(ds); torch.cuda.synchronize()                       # epoch 1 includes warm-up / dataset processing
    t0 = time.perf_counter(); model.fit_partial(ds, 1, 1); torch.cuda.synchronize(); t1 = time.perf_counter()
    n_seq = len(model.data_preparator.train_store())
    steps = -(-n_seq // model.batch_size)
    loss = model.history[-1]["train_loss"]
    assert np.isfinite(loss), (name, loss)
    reco = model.recommend(ds.user_id_map.external_ids[:512], ds, k=10, filter_viewed=True,
                           context=(pd.DataFrame({Columns.User: ds.user_id_map.external_ids[:512], Columns.Datetime: pd.Timestamp("2030-01-01")}) if isinstance(model, HSTUModel) else None))
    print(f"{name}: {n_seq} seqs/epoch, {(t1 - t0) / steps * 1e3:.2f} ms/step (incl. ~0.1-0.3 s dataset processing), {n_seq / (t1 - t0):.0f} seqs/s, "
          f"loss {loss:.4f}, recommend rows {len(reco)}")
