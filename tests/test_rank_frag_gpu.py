"""Fragment-major coarse pass of the two-stage top-k (rt_one_plane_to_fragments + rt_topk_score_two_stage(h_only = 2), opt-in with
RT_TOPK_FRAG=1): the image permutation against its numpy restatement, and the ranker's results against the single-stage kernel — ids,
counts and score bits — on shapes with an odd number of item blocks per workgroup, a viewed filter, cosine, and a block-aligned id offset."""
import os

import numpy as np
import pytest
import torch
from scipy import sparse

pytestmark = [pytest.mark.gpu]


@pytest.fixture(autouse=True)
def _frag_on(monkeypatch):
    """The fragment-major pass is the default from 500 k items up; these shapes are smaller: force it."""
    monkeypatch.setenv("RT_TOPK_FRAG", "1")


def fragments_numpy(img: np.ndarray, rows_pad: int) -> np.ndarray:
    """img [n, d / 2] int32 words (a one-plane image: d bf16 per row) -> [rows_pad * d / 8, 4] units in fragment-major order."""
    n, w = img.shape
    n_units = w // 4
    src = np.zeros((rows_pad, n_units, 4), dtype=np.int32)
    src[:n] = img.reshape(n, n_units, 4)
    out = np.zeros((rows_pad // 32, n_units // 2, 64, 4), dtype=np.int32)
    for u in range(n_units):
        out[:, u // 2, 32 * (u & 1):32 * (u & 1) + 32] = src[:, u].reshape(rows_pad // 32, 32, 4)
    return out.reshape(-1, 4)


@pytest.mark.parametrize("n,d", [(300, 128), (1000, 512), (97, 256)])
def test_fragment_major_image(n, d):
    from rectools_amd import _lib

    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(n)
    img = torch.randint(-2**31, 2**31 - 1, (n, d // 2), dtype=torch.int32, device="cuda", generator=g)
    rows_pad = (n + 127) // 128 * 128
    out = torch.full((rows_pad, d // 2), -1, dtype=torch.int32, device="cuda")
    _lib.check(lib.rt_one_plane_to_fragments(img.data_ptr(), img.stride(0), n, d, out.data_ptr(), rows_pad, _lib.current_stream()),
               "rt_one_plane_to_fragments")
    assert np.array_equal(out.cpu().numpy().reshape(-1, 4), fragments_numpy(img.cpu().numpy(), rows_pad))


@pytest.mark.parametrize("dist", ["dot", "cosine"])
@pytest.mark.parametrize("n_obj,d,n_subj,with_filter", [(200_003, 128, 300, False), (50_000, 256, 1000, True), (131_072, 512, 129, False)])
def test_fragment_major_coarse_pass_returns_the_single_stage_bits(n_obj, d, n_subj, with_filter, dist):
    from rectools_amd.rank import HipRanker

    g = torch.Generator(device="cuda").manual_seed(d)
    obj = torch.randn((n_obj, d), device="cuda", generator=g) * (0.5 + 1.5 * torch.rand((n_obj, 1), device="cuda", generator=g))
    subj = torch.randn((n_subj, d), device="cuda", generator=g)
    filt = sparse.random(n_subj, n_obj, density=0.001, format="csr", random_state=3, dtype=np.float32) if with_filter else None
    exact = HipRanker(dist, "cuda", subj, obj, batch_size=64, two_stage=False)
    fast = HipRanker(dist, "cuda", subj, obj)
    e = exact.rank_device(np.arange(n_subj), 10, filt)
    f = fast.rank_device(np.arange(n_subj), 10, filt)
    assert fast._items_frag is not None and fast.two_stage_stats["h_only_calls"] >= 1
    assert torch.equal(e[2], f[2]) and torch.equal(e[0], f[0]) and torch.equal(e[1].view(torch.int32), f[1].view(torch.int32))
