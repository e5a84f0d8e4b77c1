"""pytest configuration: registers the `gpu` marker and shared golden-fixture helpers."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need a device; skip (not fail) them when collected without one and without -m."""
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_ranker_golden():
    """-> list of dict cases from tests/golden/ranker_golden.npz (made by tests/golden/make_golden.py)."""
    from scipy import sparse

    z = np.load(os.path.join(GOLDEN_DIR, "ranker_golden.npz"), allow_pickle=False)
    cases = []
    for tag in z["__cases__"]:
        tag = str(tag)
        p = tag + "/"
        dkey = str(z[p + "data"])
        c = dict(
            tag=tag,
            users=z[dkey + "/users"],
            items=z[dkey + "/items"],
            distance=str(z[p + "distance"]),
            k=None if int(z[p + "k"]) < 0 else int(z[p + "k"]),
            sids=z[p + "sids"],
            wl=z[p + "wl"] if int(z[p + "has_wl"]) else None,
            filt=None,
            ref_subjects=z[p + "ref_subjects"],
            ref_items=z[p + "ref_items"],
            ref_scores=z[p + "ref_scores"],
        )
        if int(z[p + "has_filter"]):
            shape = tuple(int(x) for x in z[p + "f_shape"])
            c["filt"] = sparse.csr_matrix((z[p + "f_data"], z[p + "f_indices"], z[p + "f_indptr"]), shape=shape)
        cases.append(c)
    return cases


def list_transformer_golden():
    import glob

    return sorted(os.path.basename(p)[len("transformer_"):-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "transformer_*.npz")))


def load_transformer_golden(name):
    """-> (cfg dict, params0, grads, params1, params2, batch, extras) as torch CPU tensors."""
    import json

    import torch

    z = np.load(os.path.join(GOLDEN_DIR, f"transformer_{name}.npz"), allow_pickle=False)
    cfg = json.loads(str(z["config"]))

    def grab(prefix):
        return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}

    extras = {"loss": float(z["loss"]), "loss2": float(z["loss2"]), "enc": torch.from_numpy(z["enc"])}
    if "logits" in z.files:
        extras["logits"] = torch.from_numpy(z["logits"])
    return cfg, grab("p0/"), grab("g/"), grab("p1/"), grab("p2/"), grab("b/"), extras
