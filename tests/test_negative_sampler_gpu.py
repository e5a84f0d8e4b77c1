"""a11 — the on-device uniform negative sampler (`rt_sample_negatives`, CatalogUniformSampler.get_negatives of
negative_sampler.py:58-73): shape, range [n_extra, V), reproducibility, fresh stream per batch, uniformity (chi-square) and
the `negative_sampler_type` plug-in point of the models."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_shape_range_and_streams():
    from rectools_amd.data_preparator import CatalogUniformSampler

    x = torch.zeros((128, 200), dtype=torch.int64, device="cuda")
    s = CatalogUniformSampler(n_negatives=128, seed=7)
    a = s.get_negatives({"x": x}, lowest_id=1, highest_id=26_745)
    b = s.get_negatives({"x": x}, lowest_id=1, highest_id=26_745)
    v = s.get_negatives({"x": x}, lowest_id=2, highest_id=50, session_len_limit=1)
    assert a.shape == (128, 200, 128) and a.dtype == torch.int64 and a.is_cuda
    assert v.shape == (128, 1, 128)
    assert int(a.min()) >= 1 and int(a.max()) < 26_745 and int(a.min()) == 1 and int(a.max()) == 26_744
    assert int(v.min()) == 2 and int(v.max()) == 49
    assert not torch.equal(a, b)                                   # every batch draws from a fresh stream
    s2 = CatalogUniformSampler(n_negatives=128, seed=7)
    assert torch.equal(s2.get_negatives({"x": x}, 1, 26_745), a)   # (seed, batch number) fixes the batch
    s3 = CatalogUniformSampler(n_negatives=128, seed=8)
    assert not torch.equal(s3.get_negatives({"x": x}, 1, 26_745), a)
    # ragged tail (n % 4 != 0) and a tiny range
    t = CatalogUniformSampler(n_negatives=3, seed=1).get_negatives({"x": x[:3, :5]}, 5, 6)
    assert t.shape == (3, 5, 3) and bool((t == 5).all())


@pytest.mark.parametrize("low,high", [(1, 26_745), (2, 1_000_003), (1, 12)])
def test_uniformity_chi_square(low, high):
    """3.3 M draws (one C2 batch): Pearson chi-square over <= 4096 equal-width id bins stays within 5 sigma of its mean,
    and consecutive draws are uncorrelated."""
    from rectools_amd.data_preparator import CatalogUniformSampler

    x = torch.zeros((128, 200), dtype=torch.int64, device="cuda")
    neg = CatalogUniformSampler(n_negatives=128, seed=3).get_negatives({"x": x}, low, high).reshape(-1)
    n, rng = neg.numel(), high - low
    bins = min(rng, 4096)
    edges = (np.arange(bins + 1) * rng) // bins                      # integer bin edges over [0, rng)
    expected = torch.from_numpy(np.diff(edges).astype(np.float64) * n / rng).cuda()
    b = torch.bucketize(neg - low, torch.from_numpy(edges[1:-1]).cuda(), right=True)
    observed = torch.bincount(b, minlength=bins).double()
    chi2 = float(((observed - expected) ** 2 / expected).sum())
    dof = bins - 1
    assert abs(chi2 - dof) < 5.0 * np.sqrt(2.0 * dof), (chi2, dof)
    u = (neg - low).double() / rng
    corr = float(torch.corrcoef(torch.stack([u[:-1], u[1:]]))[0, 1])
    assert abs(corr) < 5.0 / np.sqrt(n)


class _FirstItemsSampler:
    """Custom plug-in: always the first real items (checks that the model routes through `negative_sampler_type`)."""

    def __init__(self, n_negatives, shift=0, **kwargs):
        self.n_negatives, self.shift, self.seen = n_negatives, shift, []

    def get_negatives(self, batch_dict, lowest_id, highest_id, session_len_limit=None, **kwargs):
        x = batch_dict["x"]
        L = session_len_limit if session_len_limit is not None else x.shape[1]
        self.seen.append((lowest_id, highest_id, L))
        ids = lowest_id + self.shift + torch.arange(self.n_negatives, device=x.device)
        return ids.expand(x.shape[0], L, self.n_negatives).contiguous()


def test_models_honour_negative_sampler_type():
    import pandas as pd

    from rectools_amd.dataset import Dataset
    from rectools_amd.models import SASRecModel
    from rectools_amd.utils import leave_one_out_mask

    rng = np.random.default_rng(0)
    df = pd.DataFrame({"user_id": rng.integers(0, 40, 900), "item_id": rng.integers(0, 30, 900) + 5, "weight": 1.0,
                       "datetime": pd.to_datetime("2022-01-01") + pd.to_timedelta(rng.integers(0, 9000, 900), unit="m")})
    model = SASRecModel(n_factors=32, n_blocks=1, n_heads=2, session_max_len=8, batch_size=16, epochs=1, loss="BCE", n_negatives=3,
                        negative_sampler_type=_FirstItemsSampler, negative_sampler_kwargs={"shift": 1},
                        get_val_mask_func=leave_one_out_mask, get_val_mask_func_kwargs={"val_users": 10}, seed=1)
    model.fit(Dataset.construct(df))
    sampler = model.data_preparator.negative_sampler
    assert isinstance(sampler, _FirstItemsSampler) and sampler.shift == 1
    V = model.data_preparator.item_id_map.size
    assert {s[:2] for s in sampler.seen} == {(1, V)}
    assert {s[2] for s in sampler.seen} == {8, 1}          # [B, L, N] for training batches, [B, 1, N] for validation
    assert np.isfinite(model.history[-1]["train_loss"]) and np.isfinite(model.history[-1]["val_loss"])
    cfg = model.get_config()
    assert cfg["negative_sampler_type"].endswith("_FirstItemsSampler") and cfg["negative_sampler_kwargs"] == {"shift": 1}
