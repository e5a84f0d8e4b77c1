"""Generate golden fixtures by running the UNMODIFIED reference (RecTools v0.17.0, /root/reference).

Run in the build container only (the reference tree does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [ranker] [transformer [only=<substring>]] [collate] [checkpoints]

Outputs (committed):  tests/golden/ranker_*.npz, transformer_*.npz, collate_*.npz, ckpt_*.ckpt
Every file stores the exact inputs next to the reference's outputs, so the oracle (`oracle/`) and the HIP
path can both be checked against them without the reference being present.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_shims  # noqa: E402

ref_shims.install()

import torch  # noqa: E402
from scipy import sparse  # noqa: E402


# --------------------------------------------------------------------------------------------------
# Ranker (reference: rectools/models/rank/rank_torch.py; KAT inputs: tests/models/rank/test_rank.py:51-64)
# --------------------------------------------------------------------------------------------------
def _csr_parts(csr):
    if csr is None:
        return dict(has_filter=np.array(0))
    csr = sparse.csr_matrix(csr)
    return dict(
        has_filter=np.array(1),
        f_indptr=csr.indptr.astype(np.int64),
        f_indices=csr.indices.astype(np.int64),
        f_data=csr.data.astype(np.float32),
        f_shape=np.array(csr.shape, dtype=np.int64),
    )


def make_ranker() -> None:
    from rectools.models.rank import Distance, TorchRanker

    cases = []

    # --- the reference's own known-answer inputs (3x3 integer factors) in every rank() mode it tests
    subj = np.array([[-4, 0, 3], [0, 1, 2]], dtype=np.float32)
    obj = np.array([[-4, 0, 3], [0, 2, 4], [1, 10, 100]], dtype=np.float32)
    kat_modes = [
        dict(name="plain", k=3, filt=None, wl=None, sids=[0, 1]),
        dict(name="filter", k=3, filt=[[0, 1, 0], [0, 0, 0]], wl=None, sids=[0, 1]),
        dict(name="whitelist", k=3, filt=None, wl=[0, 2], sids=[0, 1]),
        dict(name="wl_filter", k=3, filt=[[1, 1, 0], [0, 0, 0]], wl=[0, 2], sids=[0, 1]),
        dict(name="k2", k=2, filt=None, wl=None, sids=[0, 1]),
        dict(name="k1", k=1, filt=None, wl=None, sids=[0, 1]),
        dict(name="knone", k=None, filt=None, wl=None, sids=[0, 1]),
        dict(name="subject1", k=3, filt=None, wl=None, sids=[1]),
        dict(name="subject_rev", k=2, filt=[[0, 0, 1], [1, 0, 0]], wl=None, sids=[1, 0]),
        dict(name="all_filtered", k=3, filt=[[1, 1, 1], [0, 1, 0]], wl=None, sids=[0, 1]),
    ]
    for dist in ("dot", "cosine", "euclidean"):
        for m in kat_modes:
            cases.append(dict(tag=f"kat_{dist}_{m['name']}", distance=dist, users=subj, items=obj, **m))

    # --- seeded random, tie-free w.p. 1 (U~N(0,1) seed 1, I~N(0,1) seed 2)
    def rnd(seed, n, d):
        g = torch.Generator().manual_seed(seed)
        return torch.randn(n, d, generator=g, dtype=torch.float32).numpy()

    rs = np.random.RandomState(7)
    for (nu, ni, d, k, tag) in [(37, 301, 24, 10, "r_small"), (130, 1000, 64, 10, "r_mid"), (9, 77, 7, 77, "r_odd_d")]:
        U, I = rnd(1, nu, d), rnd(2, ni, d)
        filt = sparse.random(nu, ni, density=0.05, format="csr", random_state=rs, data_rvs=lambda n: np.ones(n))
        wl = np.sort(rs.choice(ni, size=ni * 2 // 3, replace=False))
        sids = np.arange(nu)
        for dist in ("dot", "cosine"):
            cases.append(dict(tag=f"{tag}_{dist}_plain", distance=dist, users=U, items=I, k=k, filt=None, wl=None, sids=sids))
            cases.append(dict(tag=f"{tag}_{dist}_filter", distance=dist, users=U, items=I, k=k, filt=filt, wl=None, sids=sids))
            cases.append(dict(tag=f"{tag}_{dist}_wl_filter", distance=dist, users=U, items=I, k=k, filt=filt, wl=wl, sids=sids))
        cases.append(dict(tag=f"{tag}_euclidean_wl", distance="euclidean", users=U, items=I, k=k, filt=None, wl=wl, sids=sids))
    # subject subset / permuted subjects with filter rows aligned to the subset
    U, I = rnd(1, 50, 16), rnd(2, 200, 16)
    sids = np.array([49, 3, 17, 3 + 20, 0])
    filt = sparse.random(len(sids), 200, density=0.2, format="csr", random_state=rs, data_rvs=lambda n: np.ones(n))
    cases.append(dict(tag="r_subset_dot_filter", distance="dot", users=U, items=I, k=5, filt=filt, wl=None, sids=sids))

    out = {}
    for c in cases:
        ranker = TorchRanker(distance=Distance(c["distance"]), device="cpu", subjects_factors=c["users"],
                             objects_factors=c["items"])
        filt = None if c["filt"] is None else sparse.csr_matrix(c["filt"])
        wl = None if c["wl"] is None else np.asarray(c["wl"])
        su, it, sc = ranker.rank(subject_ids=np.asarray(c["sids"]), k=c["k"], filter_pairs_csr=filt,
                                 sorted_object_whitelist=wl)
        p = c["tag"] + "/"
        # inputs are shared between cases: store each distinct (users, items) pair once
        dkey = f"data_{c['users'].shape[0]}x{c['items'].shape[0]}x{c['users'].shape[1]}"
        out[dkey + "/users"] = np.asarray(c["users"], np.float32)
        out[dkey + "/items"] = np.asarray(c["items"], np.float32)
        out[p + "data"] = np.array(dkey)
        out[p + "distance"] = np.array(c["distance"])
        out[p + "k"] = np.array(-1 if c["k"] is None else c["k"])
        out[p + "sids"] = np.asarray(c["sids"], np.int64)
        out[p + "has_wl"] = np.array(0 if wl is None else 1)
        if wl is not None:
            out[p + "wl"] = wl.astype(np.int64)
        for kk, v in _csr_parts(filt).items():
            out[p + kk] = v
        out[p + "ref_subjects"] = np.asarray(su, np.int64)
        out[p + "ref_items"] = np.asarray(it, np.int64)
        out[p + "ref_scores"] = np.asarray(sc, np.float32)
    out["__cases__"] = np.array([c["tag"] for c in cases])
    np.savez_compressed(os.path.join(HERE, "ranker_golden.npz"), **out)
    print(f"ranker: {len(cases)} cases")


if __name__ == "__main__":
    what = sys.argv[1:] or ["ranker", "transformer", "collate", "checkpoints"]
    only = next((w.split("=", 1)[1] for w in what if w.startswith("only=")), "")   # e.g. `transformer only=catfeat`
    if "ranker" in what:
        make_ranker()
    if "transformer" in what:
        from make_golden_transformer import make_transformer  # type: ignore

        make_transformer(only)
    if "collate" in what:
        from make_golden_transformer import make_collate  # type: ignore

        make_collate()
    if "checkpoints" in what:
        from make_golden_transformer import make_checkpoints  # type: ignore

        make_checkpoints()
