"""Two-stage exact top-k (include/rectools_hip.h K12c, rank.HipRanker): the hm-image coarse pass + the exact pass over 64 candidates
per user must return what the single-stage kernel returns — ids, order and score BITS (the exact pass replays the 32-wide engine's
instruction chain) — with a per-user proof, and fall back to the single-stage kernel for users where the proof cannot be given."""
import numpy as np
import pytest
import torch
from scipy import sparse

from oracle import ranker_oracle

pytestmark = pytest.mark.gpu


def _factors(n_subj, n_obj, d, seed):
    rng = np.random.default_rng(seed)
    return rng.normal(size=(n_subj, d)).astype(np.float32), (rng.normal(size=(n_obj, d)) * rng.uniform(0.5, 2.0, (n_obj, 1))).astype(np.float32)


CASES = [
    # d, n_obj, n_subj, batch (users per pass), k, filter, whitelist
    (64, 20_000, 140, 64, 10, False, None),
    (256, 131_109, 300, 128, 10, True, None),
    (128, 50_003, 133, 64, 5, True, "sparse"),
    (512, 30_000, 200, 128, 16, False, "range"),
    (64, 9_000, 130, 128, 1, True, "range"),
    (96, 26_744, 1000, None, 10, True, "range"),       # the recommend() shape: library defaults
    (32, 4_000, 70, 32, 3, False, None),
]


@pytest.mark.parametrize("plane", ["h", "hm"])
@pytest.mark.parametrize("dist", ["dot", "cosine"])
@pytest.mark.parametrize("d,n_obj,n_subj,batch,k,with_filter,wl_kind", CASES)
def test_two_stage_returns_the_single_stage_result_bit_for_bit(d, n_obj, n_subj, batch, k, with_filter, wl_kind, dist, plane):
    """plane "h": the one-plane coarse image (the default where the row width allows it: d % 64 == 0); "hm": the two-plane image."""
    from rectools_amd.rank import HipRanker

    subj, obj = _factors(n_subj, n_obj, d, seed=n_obj % 97)
    rng = np.random.default_rng(5)
    ids = rng.permutation(n_subj)[: max(1, n_subj - 3)]
    filt = None
    if with_filter:
        filt = sparse.random(len(ids), n_obj, density=0.002, format="csr", random_state=7, dtype=np.float32)
    wl = None
    if wl_kind == "sparse":
        wl = np.sort(rng.permutation(n_obj)[: n_obj // 2])
    elif wl_kind == "range":
        wl = np.arange(1000, n_obj - 500)
    exact = HipRanker(dist, "cuda", subj, obj, batch_size=batch if batch else 64, two_stage=False)
    fast = HipRanker(dist, "cuda", subj, obj, batch_size=batch, two_stage=True)
    if plane == "hm":
        fast._h_only_off = True
    elif d % 64 != 0:
        pytest.skip("the one-plane image needs d % 64 == 0: this shape runs on the (h, m) image either way")
    e_ids, e_sc, e_cnt, _ = exact.rank_device(ids, k, filt, wl)
    f_ids, f_sc, f_cnt, _ = fast.rank_device(ids, k, filt, wl)
    st = fast.two_stage_stats
    if plane == "hm":
        assert st["calls"] == 1 and st["unproven_users"] == 0 and st["h_only_calls"] == 0, st
    else:       # the one-plane bound is 2^7 wider: a user it leaves unproven is re-ranked (same result below), most are proven
        assert st["h_only_calls"] == 1 and st["unproven_users"] <= len(ids) // 8, st
    assert torch.equal(e_cnt, f_cnt)
    valid = torch.arange(e_ids.shape[1], device="cuda")[None, :] < e_cnt[:, None]
    assert torch.equal(e_ids[valid], f_ids[valid])
    assert torch.equal(f_sc[valid].view(torch.int32), e_sc[valid].view(torch.int32))          # the same bits
    s_o, i_o, sc_o = ranker_oracle.rank(subj, obj, ids[:5], k=k, filter_pairs_csr=None if filt is None else filt[:5],
                                        sorted_object_whitelist=wl, distance=dist)
    assert f_ids[:5][valid[:5]].cpu().numpy().tolist() == np.asarray(i_o).tolist()              # and the CPU oracle's order


def test_two_stage_is_the_default_for_many_users_with_dot_and_cosine():
    from rectools_amd.rank import HipRanker

    subj, obj = _factors(300, 20_000, 64, 3)
    r = HipRanker("dot", "cuda", subj, obj)
    r.rank_device(np.arange(300), 10)
    n_calls = r.two_stage_stats["calls"]
    assert n_calls >= 1
    r.rank_device(np.arange(12), 10)                        # few users: the HBM-bound 16-user tile of the single-stage kernel
    assert r.two_stage_stats["calls"] == n_calls
    c = HipRanker("cosine", "cuda", subj, obj)
    c.rank_device(np.arange(300), 10)
    assert c.two_stage_stats["calls"] >= 1                  # cosine too (unit-row images, exact cosine in the second stage)
    e = HipRanker("euclidean", "cuda", subj, obj, two_stage=True)
    e.rank_device(np.arange(300), 10)
    assert e.two_stage_stats["calls"] == 0


@pytest.mark.parametrize("plane", ["h", "hm"])
def test_users_whose_result_cannot_be_proven_are_ranked_by_the_single_stage_kernel(plane):
    """Exact duplicates of the best items (ties at the k-th place) and a block of near-duplicates inside the coarse error window: the
    proof fails for the affected users — flagged per user, re-ranked by the single-stage kernel, same bits in the end."""
    from rectools_amd.rank import HipRanker

    rng = np.random.default_rng(1)
    d, n_obj = 64, 30_000
    obj = rng.normal(size=(n_obj, d)).astype(np.float32)
    subj = rng.normal(size=(200, d)).astype(np.float32)
    # users 0..9 point at item 17's direction; items 100..229 are 130 copies of item 17 (more than the 64 candidates): their top-10 ties
    obj[100:230] = obj[17]
    subj[:10] = obj[17][None, :] * rng.uniform(0.5, 2.0, (10, 1)).astype(np.float32)
    # users 20..24: 150 near-duplicates (relative spread 1e-6, far inside the coarse error) of their best direction
    obj[1000:1150] = obj[33][None, :] * (1.0 + 1e-6 * rng.normal(size=(150, 1))).astype(np.float32)
    subj[20:25] = obj[33][None, :] * rng.uniform(0.5, 2.0, (5, 1)).astype(np.float32)
    exact = HipRanker("dot", "cuda", subj, obj, batch_size=64, two_stage=False)
    fast = HipRanker("dot", "cuda", subj, obj, batch_size=64, two_stage=True)
    fast._h_only_off = plane == "hm"
    e_ids, e_sc, e_cnt, _ = exact.rank_device(np.arange(200), 10)
    f_ids, f_sc, f_cnt, _ = fast.rank_device(np.arange(200), 10)
    st = fast.two_stage_stats
    if plane == "hm":
        assert st["calls"] == 1 and st["fallbacks"] == 0 and 15 <= st["unproven_users"] <= 40, st
    else:   # the one-plane bound is 2^8 wider: whatever it leaves unproven is re-ranked (by runs, or by the (h, m) pass again)
        assert st["h_only_calls"] == 1 and st["unproven_users"] >= 15, st
    assert torch.equal(e_ids, f_ids) and torch.equal(e_sc.view(torch.int32), f_sc.view(torch.int32)) and torch.equal(e_cnt, f_cnt)
    # a catalog made of near-duplicates only defeats the coarse pass for everybody: the whole call goes to the single-stage kernel
    base = rng.normal(size=(50, d)).astype(np.float32)
    obj2 = np.repeat(base, 400, axis=0) * (1.0 + 1e-6 * rng.normal(size=(20_000, 1))).astype(np.float32)
    exact2 = HipRanker("dot", "cuda", subj, obj2, batch_size=64, two_stage=False)
    fast2 = HipRanker("dot", "cuda", subj, obj2, batch_size=64, two_stage=True)
    fast2._h_only_off = plane == "hm"
    e2 = exact2.rank_device(np.arange(200), 10)
    f2 = fast2.rank_device(np.arange(200), 10)
    # "h": one-plane pass -> (h, m) pass -> single-stage kernel, and the one-plane image is not tried again on this catalog
    assert fast2.two_stage_stats["fallbacks"] == (1 if plane == "hm" else 2) and fast2._h_only_off
    assert torch.equal(e2[0], f2[0]) and torch.equal(e2[1].view(torch.int32), f2[1].view(torch.int32))


def test_unsettled_calls_are_repaired_by_one_settle():
    """`rank_device(..., settle=False)` returns without reading the proof flags (calls queue back to back on the stream);
    `settle()` reads the flags of every such call at once and re-ranks the unproven users INTO the tensors the calls returned."""
    from rectools_amd.rank import HipRanker

    rng = np.random.default_rng(1)
    d, n_obj = 64, 30_000
    obj = rng.normal(size=(n_obj, d)).astype(np.float32)
    subj = rng.normal(size=(200, d)).astype(np.float32)
    obj[100:230] = obj[17]                                   # ties at the k-th place for users 0..9 (see the test above)
    subj[:10] = obj[17][None, :] * rng.uniform(0.5, 2.0, (10, 1)).astype(np.float32)
    exact = HipRanker("dot", "cuda", subj, obj, batch_size=64, two_stage=False)
    fast = HipRanker("dot", "cuda", subj, obj, batch_size=64, two_stage=True)
    fast._h_only_off = True
    users = [np.arange(200), np.arange(100, 200), np.arange(0, 50)]      # the second call has no unproven user
    want = [exact.rank_device(u, 10) for u in users]
    got = [fast.rank_device(u, 10, settle=False) for u in users]
    assert len(fast._unsettled) == 3 and fast.two_stage_stats["unproven_users"] == 0
    repaired = fast.settle()
    assert repaired >= 20 and fast.two_stage_stats["unproven_users"] == repaired and not fast._unsettled and fast.settle() == 0
    for (e_ids, e_sc, e_cnt, _), (f_ids, f_sc, f_cnt, _) in zip(want, got):
        assert torch.equal(e_ids, f_ids) and torch.equal(e_sc.view(torch.int32), f_sc.view(torch.int32)) and torch.equal(e_cnt, f_cnt)
    # the default keeps the call self-contained
    f_ids = fast.rank_device(users[0], 10)[0]
    assert not fast._unsettled and torch.equal(f_ids, want[0][0])


def test_hm_image_kernel():
    """rt_to_hm_rows: word = (h << 16) | m with h, m the bf16 truncations of x and x - h; x - h - m below 2^-15 |x|; norms; gather."""
    from rectools_amd import _lib

    lib = _lib.load()
    torch.manual_seed(0)
    x = (torch.randn(300, 192) * torch.logspace(-3, 3, 300)[:, None]).cuda()
    x[5, :7] = 0.0
    rows = torch.randperm(300)[:77].cuda()
    img = torch.empty(77, 192, dtype=torch.int32, device="cuda")
    norms = torch.empty(77, device="cuda")
    _lib.check(lib.rt_to_hm_rows(x.data_ptr(), x.stride(0), rows.data_ptr(), 77, 192, 0, img.data_ptr(), 192, norms.data_ptr(),
                                 _lib.current_stream()), "rt_to_hm_rows")
    src = x[rows]
    h = (img & -65536).view(torch.float32)
    m = (img << 16).view(torch.float32)
    assert torch.equal(h, (src.view(torch.int32) & -65536).view(torch.float32))
    r = src - h
    assert torch.equal(m, (r.view(torch.int32) & -65536).view(torch.float32))
    assert bool(((src - h - m).abs() <= src.abs() * 2.0 ** -15).all())
    torch.testing.assert_close(norms, src.double().norm(dim=1).float(), rtol=1e-5, atol=0)
    unit = torch.empty_like(img)                             # normalize = 1: the image of the unit rows
    _lib.check(lib.rt_to_hm_rows(x.data_ptr(), x.stride(0), rows.data_ptr(), 77, 192, 1, unit.data_ptr(), 192, norms.data_ptr(),
                                 _lib.current_stream()), "rt_to_hm_rows")
    uh, um = (unit & -65536).view(torch.float32), (unit << 16).view(torch.float32)
    torch.testing.assert_close((uh + um).double().norm(dim=1), torch.ones(77, dtype=torch.float64, device="cuda"), rtol=1e-4, atol=0)


def test_two_stage_with_fewer_than_k_candidates_left_by_the_filter():
    """Users whose viewed-items filter leaves fewer than k (or exactly zero) candidates: counts, ids and score bits as the single-stage
    kernel — nothing was dropped, so the short result is proven complete."""
    from rectools_amd.rank import HipRanker

    rng = np.random.default_rng(3)
    n_obj, d, n_subj, k = 2_000, 64, 150, 10
    subj, obj = _factors(n_subj, n_obj, d, 5)
    rows, cols = [], []
    for u in range(n_subj):
        keep = rng.integers(0, 25) if u % 3 else 0          # every third user keeps nothing, the others 0..24 items
        seen = rng.permutation(n_obj)[: n_obj - keep]
        rows.append(np.full(len(seen), u)); cols.append(seen)
    filt = sparse.csr_matrix((np.ones(sum(len(c) for c in cols), np.float32), (np.concatenate(rows), np.concatenate(cols))),
                             shape=(n_subj, n_obj))
    exact = HipRanker("dot", "cuda", subj, obj, batch_size=64, two_stage=False)
    fast = HipRanker("dot", "cuda", subj, obj, batch_size=64, two_stage=True)
    e_ids, e_sc, e_cnt, _ = exact.rank_device(np.arange(n_subj), k, filt)
    f_ids, f_sc, f_cnt, _ = fast.rank_device(np.arange(n_subj), k, filt)
    assert fast.two_stage_stats["calls"] == 1 and fast.two_stage_stats["fallbacks"] == 0
    assert torch.equal(e_cnt, f_cnt) and int(e_cnt.min()) == 0 and 0 < int((e_cnt < k).sum()) < n_subj
    valid = torch.arange(k, device="cuda")[None, :] < e_cnt[:, None]
    assert torch.equal(e_ids[valid], f_ids[valid]) and torch.equal(e_sc[valid].view(torch.int32), f_sc[valid].view(torch.int32))
    # the numpy front end drops the empty rows the same way in both modes
    a, b = exact.rank(np.arange(n_subj), k, filt), fast.rank(np.arange(n_subj), k, filt)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("dist", ["dot", "cosine"])
@pytest.mark.parametrize("n_subj,with_filter", [(16, False), (32, True), (5, False)])
def test_h_only_coarse_pass_for_a_few_users_returns_the_single_stage_bits(n_subj, with_filter, dist):
    """The HBM-bound regime (a few users against a catalog that does not fit the caches): the coarse pass streams a ONE-plane bf16 image
    (half the catalog bytes, coarse error 2^-7 |u| |v|), the exact pass and the proof are the same — ids, order and score bits of the
    32-wide single-stage engine."""
    from rectools_amd.rank import HipRanker

    n_obj, d, k = 600_000, 128, 10                         # 307 MB of fp32 rows
    g = torch.Generator(device="cuda").manual_seed(n_subj)
    obj = torch.randn((n_obj, d), device="cuda", generator=g) * (0.5 + 1.5 * torch.rand((n_obj, 1), device="cuda", generator=g))
    subj = torch.randn((n_subj, d), device="cuda", generator=g)
    filt = sparse.random(n_subj, n_obj, density=0.0005, format="csr", random_state=3, dtype=np.float32) if with_filter else None
    exact = HipRanker(dist, "cuda", subj, obj, batch_size=32, two_stage=False)
    fast = HipRanker(dist, "cuda", subj, obj)
    e_ids, e_sc, e_cnt, _ = exact.rank_device(np.arange(n_subj), k, filt)
    f_ids, f_sc, f_cnt, _ = fast.rank_device(np.arange(n_subj), k, filt)
    st = fast.two_stage_stats
    assert st["calls"] == 1 and st["h_only_calls"] == 1 and st["unproven_users"] <= n_subj // 8, st
    assert torch.equal(e_cnt, f_cnt) and torch.equal(e_ids, f_ids) and torch.equal(e_sc.view(torch.int32), f_sc.view(torch.int32))


def test_one_plane_pass_with_many_users_falls_back_to_the_two_plane_pass():
    """Many users, a catalog of 100-fold near-duplicates (relative spread 1e-3): more look-alikes than the 64 candidates inside the
    one-plane bound (2^-7 |u| |v|), but well apart under the (h, m) bound (2^-14): the call is repeated over the (h, m) image — proven
    there — and the ranker keeps to that image afterwards."""
    from rectools_amd.rank import HipRanker

    g = torch.Generator(device="cuda").manual_seed(2)
    base = torch.randn((600, 128), device="cuda", generator=g)
    obj = base.repeat_interleave(100, dim=0) * (1.0 + 1e-3 * torch.randn((60_000, 1), device="cuda", generator=g))
    subj = torch.randn((256, 128), device="cuda", generator=g)
    exact = HipRanker("dot", "cuda", subj, obj, batch_size=64, two_stage=False)
    fast = HipRanker("dot", "cuda", subj, obj)
    e = exact.rank_device(np.arange(256), 10)
    f = fast.rank_device(np.arange(256), 10)
    st = fast.two_stage_stats
    assert st["h_only_calls"] == 1 and st["calls"] == 2 and st["fallbacks"] == 1 and fast._h_only_off, st
    assert st["unproven_users"] >= 200, st        # all of them from the one-plane pass
    assert torch.equal(e[0], f[0]) and torch.equal(e[1].view(torch.int32), f[1].view(torch.int32)) and torch.equal(e[2], f[2])
    fast.rank_device(np.arange(256), 10)
    assert fast.two_stage_stats["h_only_calls"] == 1 and fast.two_stage_stats["calls"] == 3


def test_h_only_is_dropped_for_a_catalog_that_defeats_its_bound():
    """Near-duplicate items closer than the one-plane bound: the call falls back to the single-stage kernel (same result) and the ranker
    stops using the one-plane image; the (h, m) image still serves larger calls."""
    from rectools_amd.rank import HipRanker

    g = torch.Generator(device="cuda").manual_seed(1)
    base = torch.randn((300, 128), device="cuda", generator=g)
    obj = base.repeat_interleave(2000, dim=0) * (1.0 + 1e-4 * torch.randn((600_000, 1), device="cuda", generator=g))
    subj = torch.randn((8, 128), device="cuda", generator=g)
    exact = HipRanker("dot", "cuda", subj, obj, batch_size=32, two_stage=False)
    fast = HipRanker("dot", "cuda", subj, obj)
    e = exact.rank_device(np.arange(8), 10)
    f = fast.rank_device(np.arange(8), 10)
    assert fast.two_stage_stats["h_only_calls"] == 1 and fast.two_stage_stats["fallbacks"] == 1 and fast._h_only_off
    assert torch.equal(e[0], f[0]) and torch.equal(e[1].view(torch.int32), f[1].view(torch.int32))
    fast.rank_device(np.arange(8), 10)
    assert fast.two_stage_stats["h_only_calls"] == 1          # not tried again


@pytest.mark.parametrize("n_subj", [16, 4096])
def test_two_stage_at_the_baseline_catalog_5m_x_512(n_subj):
    """BASELINE.json configs[4] at its stated size (5,000,000 x 512 fp32 = 10.24 GB), 16 users (the HBM-bound launch) and 4,096 users
    (the matrix-bound one; fragment-major coarse pass), viewed-items filter on: the two-stage path returns the single-stage kernel's
    ids, counts and score BITS for every user, and the CPU oracle's ids on a 200,000-row slice for 32 users (this
    equality existed only in builder-side scripts)."""
    from rectools_amd.rank import HipRanker

    if torch.cuda.mem_get_info()[0] < 60 << 30:
        pytest.skip("needs ~40 GB of free HBM (catalog + two images + workspaces)")
    n_obj, d, k = 5_000_000, 512, 10
    g = torch.Generator(device="cuda").manual_seed(11)
    obj = torch.randn((n_obj, d), device="cuda", generator=g)
    subj = torch.randn((n_subj, d), device="cuda", generator=g)
    ids = np.arange(n_subj)
    nnz_per_user = 50
    rng = np.random.default_rng(5)
    cols = rng.integers(0, n_obj, size=(n_subj, nnz_per_user))
    filt = sparse.csr_matrix((np.ones(n_subj * nnz_per_user, np.float32), (np.repeat(ids, nnz_per_user), cols.reshape(-1))), shape=(n_subj, n_obj))
    filt.sum_duplicates()
    fast = HipRanker("dot", "cuda", subj, obj)
    f_ids, f_sc, f_cnt, _ = fast.rank_device(ids, k, filt)
    st = dict(fast.two_stage_stats)
    assert st["calls"] >= 1 and st["h_only_calls"] >= 1 and st["fallbacks"] == 0, st
    if n_subj >= 128:
        assert fast._items_frag is not None          # the fragment-major coarse pass is the default at this catalog size
    # (the exact pass replays the 32- / 64-user engines' instruction chain: those are the single-stage kernels whose BITS it returns; the
    # 16-user tile accumulates in another order — same ids, scores a rounding unit apart)
    exact = HipRanker("dot", "cuda", subj, obj, batch_size=32 if n_subj <= 32 else 64, two_stage=False)
    e_ids, e_sc, e_cnt, _ = exact.rank_device(ids, k, filt)
    assert torch.equal(e_cnt, f_cnt) and int(e_cnt.min()) == k
    assert torch.equal(e_ids, f_ids)
    assert torch.equal(e_sc.view(torch.int32), f_sc.view(torch.int32))
    del fast, exact
    # the oracle on a slice of the catalog for (up to) 32 users: same ids, same order
    m, n_slice = min(32, n_subj), 200_000
    sub = HipRanker("dot", "cuda", subj[:m], obj[:n_slice], two_stage=True)
    s_ids, _, s_cnt, _ = sub.rank_device(np.arange(m), k, filt[:m, :n_slice])
    _, i_o, _ = ranker_oracle.rank(subj[:m].cpu().numpy(), obj[:n_slice].cpu().numpy(), np.arange(m), k=k, filter_pairs_csr=filt[:m, :n_slice])
    assert s_ids.cpu().numpy().reshape(-1).tolist() == np.asarray(i_o).tolist()
