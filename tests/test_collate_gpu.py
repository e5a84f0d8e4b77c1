"""Device collation (`rt_collate`) against the host collates, which are pinned to the reference's own
collate outputs (tests/golden/collate_golden.npz, tests/test_host_path.py): integer work, so the bar is bit equality."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _store(n_sessions, max_len, n_items, seed, with_ts):
    from rectools_amd.data_preparator import SequenceStore

    rng = np.random.default_rng(seed)
    lens = rng.integers(1, max_len + 1, n_sessions)
    lens[:4] = [1, 2, max_len, max_len]                     # shortest sessions and ones longer than any window
    offsets = np.r_[0, np.cumsum(lens)].astype(np.int64)
    items = rng.integers(2, n_items, offsets[-1]).astype(np.int64)
    weights = rng.random(offsets[-1]).astype(np.float32) + 0.5
    ts = np.cumsum(rng.integers(1, 10_000, offsets[-1])).astype(np.int64) + 1_400_000_000 if with_ts else None
    return SequenceStore(offsets, items, weights, ts, np.arange(n_sessions))


class _IdMap:
    size = 500


def _prep(cls, L, add_ts=False, **kw):
    from rectools_amd import data_preparator as dpm

    dp = cls.__new__(cls)
    dp.session_max_len, dp.add_unix_ts = L, add_ts
    dp.item_id_map = _IdMap()
    dp.extra_token_ids = {dpm.PADDING_VALUE: 0, dpm.MASKING_VALUE: 1}
    for k, v in kw.items():
        setattr(dp, k, v)
    return dp


@pytest.mark.parametrize("L,with_ts", [(5, False), (50, True), (200, False), (7, True)])
def test_sasrec_device_collates_equal_host(L, with_ts):
    from rectools_amd.data_preparator import DeviceSequenceStore, SASRecDataPreparator

    store = _store(300, 260, 500, seed=L, with_ts=with_ts)
    dp = _prep(SASRecDataPreparator, L, with_ts)
    idx = np.random.default_rng(1).permutation(len(store))[:128]
    dstore = DeviceSequenceStore(store, "cuda")
    idx_t = torch.from_numpy(idx).cuda()
    for host, dev in ((dp.collate_train(store, idx), dp.collate_train_device(dstore, idx_t)),
                      (dp.collate_recommend(store, idx), dp.collate_recommend_device(dstore, idx_t))):
        assert set(host) == set(dev)
        for k in host:
            assert dev[k].dtype == torch.from_numpy(host[k]).dtype
            assert np.array_equal(dev[k].cpu().numpy(), host[k]), k


@pytest.mark.parametrize("L,mask_prob", [(6, 0.5), (50, 0.15), (200, 0.3)])
def test_bert4rec_device_collates_equal_host_given_the_draws(L, mask_prob):
    from rectools_amd import data_preparator as dpm

    store = _store(200, L, 500, seed=10 + L, with_ts=False)    # BERT4Rec train sessions hold at most L items
    dp = _prep(dpm.BERT4RecDataPreparator, L, mask_prob=mask_prob)
    idx = np.random.default_rng(2).permutation(len(store))[:96]
    dstore = dpm.DeviceSequenceStore(store, "cuda")
    idx_t = torch.from_numpy(idx).cuda()
    g = torch.Generator(device="cuda").manual_seed(5)
    probs = torch.rand((len(idx), L), generator=g, device="cuda")
    rand_ids = torch.randint(2, 500, (len(idx), L), generator=g, device="cuda")
    dev = dpm._device_collate(dstore, idx_t, L, 3, False, probs, rand_ids, mask_prob, 1)
    host = dp.collate_train_with_draws(store, idx, probs.cpu().numpy(), rand_ids.cpu().numpy())
    for k in host:
        assert np.array_equal(dev[k].cpu().numpy(), host[k]), k
    # statistical sanity of the product entry point (its own draws): share of target positions ~ mask_prob
    torch.manual_seed(1234)      # the product entry point draws from torch's device generator
    hits, n_real = 0.0, 0.0
    for _ in range(8):           # 8 batches: thousands of Bernoulli draws instead of a few hundred
        out = dp.collate_train_device(dstore, idx_t)
        real = (out["yw"] > 0)
        hits += float(((out["y"] != 0) & real).sum()); n_real += float(real.sum())
    share = hits / n_real
    assert abs(share - mask_prob) < 4.0 * (mask_prob * (1 - mask_prob) / n_real) ** 0.5 + 1e-3, (share, n_real)
    hostr, devr = dp.collate_recommend(store, idx), dp.collate_recommend_device(dstore, idx_t)
    assert np.array_equal(devr["x"].cpu().numpy(), hostr["x"])
