#!/usr/bin/env python
"""Trajectory fixture at BASELINE config 1 (SASRec d64, 1 block, 4 heads, L50, full softmax, B128, ML-1M-shaped data):
ONE EPOCH of the UNMODIFIED reference (through oracle/ref_shims.py, CPU fp32), recorded step by step.

    PYTHONPATH=/root/repo python tests/golden/make_golden_trajectory.py      # build container only (needs /root/reference)

Stored in tests/golden/trajectory_c1.npz:
  p0/<name>            state_dict after `on_train_start` (xavier init) — the weights both engines start from
  x, y [S,B,L] int16   the batches in the order the reference's shuffling DataLoader produced them (yw == (y != 0))
  loss [S]             the reference's training_step loss of every step (dropout 0: deterministic)
  rec_x [U,L], filt_indptr/filt_indices, rec_items [U,10], rec_scores [U,10]
                       recommend(users[:U], k=10, filter_viewed=True) of the trained reference model, internal item ids
tests/test_trajectory_gpu.py replays the same batches from the same weights on the HIP engine and compares the loss curve
and the final top-10 lists (SURVEY.md §8c "end-to-end item ids + ranks after training", test_sasrec.py:163-305).
"""
import os
import sys

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_shims  # noqa: E402

ref_shims.install()

from rectools import Columns  # noqa: E402
from rectools.dataset import Dataset  # noqa: E402
from rectools.models import SASRecModel  # noqa: E402

from rectools_amd import synth  # noqa: E402

CFG = dict(n_factors=64, n_blocks=1, n_heads=4, session_max_len=50, dropout_rate=0.0, loss="softmax", batch_size=128, epochs=1,
           lr=1e-3)
N_REC_USERS = 512


def main() -> None:
    u, it, ts = synth.gen_interactions(synth.ML_1M["n_users"], synth.ML_1M["n_items"], mean_len=synth.ML_1M["mean_len"],
                                       min_len=synth.ML_1M["min_len"], max_len=synth.ML_1M["max_len"], seed=0)
    df = pd.DataFrame({Columns.User: u, Columns.Item: it + 1000, Columns.Weight: 1.0,
                       Columns.Datetime: pd.to_datetime(ts, unit="s")})
    ds = Dataset.construct(df)
    ref_shims.seed_all(32)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    model = SASRecModel(deterministic=True, **CFG)
    rec = {"x": [], "y": [], "yw_is_mask": True, "loss": [], "p0": None}

    # record what the reference's own loop sees: the state after xavier init, every batch, every loss
    from rectools.models.nn.transformers.lightning import TransformerLightningModule

    orig_start, orig_step = TransformerLightningModule.on_train_start, TransformerLightningModule.training_step

    def on_train_start(self):
        orig_start(self)
        rec["p0"] = {k: v.detach().clone() for k, v in self.torch_model.state_dict().items()}

    def training_step(self, batch, batch_idx):
        rec["x"].append(batch["x"].clone()); rec["y"].append(batch["y"].clone())
        rec["yw_is_mask"] &= bool(torch.equal(batch["yw"], (batch["y"] != 0).float()))
        loss = orig_step(self, batch, batch_idx)
        rec["loss"].append(float(loss.detach()))
        return loss

    TransformerLightningModule.on_train_start, TransformerLightningModule.training_step = on_train_start, training_step
    try:
        model.fit(ds)
    finally:
        TransformerLightningModule.on_train_start, TransformerLightningModule.training_step = orig_start, orig_step
    assert rec["yw_is_mask"]
    B, L = CFG["batch_size"], CFG["session_max_len"]
    full = [i for i, x in enumerate(rec["x"]) if x.shape[0] == B]      # the last batch of the epoch may be ragged
    assert full == list(range(len(full))) and len(full) >= len(rec["x"]) - 1
    out = {"p0/" + k: v.numpy() for k, v in rec["p0"].items()}
    out["x"] = torch.stack([rec["x"][i] for i in full]).numpy().astype(np.int16)
    out["y"] = torch.stack([rec["y"][i] for i in full]).numpy().astype(np.int16)
    if len(full) < len(rec["x"]):
        out["x_last"] = rec["x"][-1].numpy().astype(np.int16); out["y_last"] = rec["y"][-1].numpy().astype(np.int16)
    out["loss"] = np.asarray(rec["loss"], dtype=np.float64)

    users = np.arange(N_REC_USERS)
    reco = model.recommend(users=users, dataset=ds, k=10, filter_viewed=True)
    dp = model.data_preparator
    assert reco.groupby(Columns.User, sort=False).size().eq(10).all() and reco[Columns.User].values[::10].tolist() == users.tolist()
    out["rec_items"] = dp.item_id_map.convert_to_internal(reco[Columns.Item].values).reshape(N_REC_USERS, 10).astype(np.int32)
    out["rec_scores"] = reco[Columns.Score].values.reshape(N_REC_USERS, 10).astype(np.float32)
    rec_ds = dp.transform_dataset_u2i(ds, users)
    xs = [b["x"] for b in dp.get_dataloader_recommend(rec_ds, 256)]
    out["rec_x"] = torch.cat(xs).numpy().astype(np.int16)
    uid = rec_ds.user_id_map.convert_to_internal(users)
    assert uid.tolist() == sorted(uid.tolist())
    csr = rec_ds.get_user_item_matrix(include_weights=False)[uid]
    csr.sort_indices()
    out["filt_indptr"] = csr.indptr.astype(np.int64); out["filt_indices"] = csr.indices.astype(np.int32)
    out["n_tokens"] = np.array(dp.item_id_map.size)
    out["cfg"] = np.array(repr(CFG))
    np.savez_compressed(os.path.join(HERE, "trajectory_c1.npz"), **out)
    print(f"steps={len(rec['loss'])} (full {len(full)}) loss[0]={rec['loss'][0]:.5f} loss[-1]={rec['loss'][-1]:.5f} "
          f"n_tokens={dp.item_id_map.size} size={os.path.getsize(os.path.join(HERE, 'trajectory_c1.npz')) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
