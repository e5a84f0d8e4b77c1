"""TEST INFRASTRUCTURE ONLY — never imported by the product path (`rectools_amd/`).

The UNMODIFIED reference timed on the host cores (bench.py's `cpu_baseline` leg, kind "reference"): the reference's own
`TransformerLightningModule.training_step` + `backward` + `torch.optim.Adam.step` over batches of its own DataLoader, and
`TorchRanker(device="cpu").rank`.  The package is found in `/root/reference` (build container) or in the staged copy `oracle/_ref`
(GPU box; `oracle/make_ref.py`), imported through `oracle/ref_shims.py` (typeguard / implicit / pytorch_lightning are not installable).
"""
from __future__ import annotations

import os
import time
import typing as tp

import numpy as np


def available() -> bool:
    from . import ref_shims

    return ref_shims.reference_available()


def train_step_rate(interactions: tp.Any, model_kwargs: tp.Dict[str, tp.Any], budget_s: float = 25.0, max_steps: int = 6,
                    threads: tp.Optional[int] = None) -> tp.Tuple[float, int, int, str]:
    """-> (sequences per second, timed steps, batch size, description): the reference's SASRecModel built on `interactions`
    (a pandas frame user_id / item_id / weight / datetime), its own train DataLoader, one untimed step, then whole steps until the
    budget is used (at least 2)."""
    import torch

    from . import ref_shims

    ref_shims.install()
    from rectools.dataset import Dataset
    from rectools.models import SASRecModel

    if threads:
        torch.set_num_threads(threads)
    ref_shims.seed_all(32)
    torch.use_deterministic_algorithms(False)
    model = SASRecModel(**model_kwargs)
    ds = Dataset.construct(interactions)
    model._build_model_from_dataset(ds)          # pylint: disable=protected-access
    lm = model.lightning_model
    loader = model.data_preparator.get_dataloader_train()
    lm.train()
    opt = lm.configure_optimizers()
    it = iter(loader)
    lm.on_train_start()
    n, seqs, t0 = 0, 0, None
    for i, batch in enumerate(it):
        if i == 1:
            t0 = time.perf_counter()
        opt.zero_grad()
        loss = lm.training_step(batch, i)
        loss.backward()
        opt.step()
        if i >= 1:
            n += 1
            seqs += int(batch["x"].shape[0])
            if n >= max_steps or (n >= 2 and time.perf_counter() - t0 > budget_s):
                break
    el = time.perf_counter() - t0
    what = (f"unmodified reference (RecTools SASRecModel, {ref_shims.REFERENCE_ROOT}): TransformerLightningModule.training_step + backward + "
            f"torch.optim.Adam.step on batches of its own DataLoader (collate + negative sampling included), {n} steps of "
            f"{int(batch['x'].shape[0])} sequences after 1 warm-up step, {torch.get_num_threads()} torch threads")
    return seqs / el, n, int(batch["x"].shape[0]), what


def rank_rate(users: np.ndarray, items: np.ndarray, filt: tp.Any, k: int = 10, budget_s: float = 15.0,
              threads: tp.Optional[int] = None) -> tp.Tuple[float, int, str]:
    """-> (users per second, users ranked, description): `TorchRanker(Distance.DOT, device="cpu").rank` with the reference's defaults
    (batch_size 128), filter CSR as given."""
    import torch

    from . import ref_shims

    ref_shims.install()
    from rectools.models.rank import Distance, TorchRanker

    if threads:
        torch.set_num_threads(threads)
    ranker = TorchRanker(distance=Distance.DOT, device="cpu", subjects_factors=users, objects_factors=items)
    n = users.shape[0]
    done, t0 = 0, time.perf_counter()
    chunk = min(n, 1024)
    while True:
        lo = done % max(n - chunk + 1, 1)
        ids = np.arange(lo, lo + chunk)
        ranker.rank(ids, k=k, filter_pairs_csr=filt[ids] if filt is not None else None)
        done += chunk
        el = time.perf_counter() - t0
        if el > budget_s or done >= 8 * chunk:
            break
    what = (f"unmodified reference TorchRanker(device='cpu').rank (rank_torch.py:77-223), k={k}, {done} users in chunks of {chunk}, "
            f"{torch.get_num_threads()} torch threads")
    return done / el, done, what
