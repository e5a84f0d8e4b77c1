"""Numerical model of the planned two-stage exact top-k: bf16 coarse scores + margin -> candidate set ->
exact fp32 rescoring.  CPU / numpy only.  Checks (a) the error bound |a_bf16 - a_fp32| <= c * |u| * |v| with c = 2^-8 + slack,
(b) that the candidate set always contains the exact top-k, (c) how many candidates survive per user."""
import sys

import numpy as np


def to_bf16(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even float32 -> bfloat16, returned as float32 values."""
    b = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    rounded = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16) << 16
    return rounded.astype(np.uint32).view(np.float32)


def run(n_items, d, n_users, k, kind, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "gauss":
        items = rng.normal(size=(n_items, d)).astype(np.float32)
        users = rng.normal(size=(n_users, d)).astype(np.float32)
    else:   # "trained-like": low-rank structure + popularity-dependent norms, users inside the item span
        basis = rng.normal(size=(16, d)).astype(np.float32)
        items = (rng.normal(size=(n_items, 16)).astype(np.float32) @ basis) * (0.2 + rng.pareto(3.0, (n_items, 1))).astype(np.float32)
        items += 0.1 * rng.normal(size=(n_items, d)).astype(np.float32)
        users = rng.normal(size=(n_users, 16)).astype(np.float32) @ basis
    exact = users.astype(np.float64) @ items.astype(np.float64).T
    coarse = (to_bf16(users) @ to_bf16(items).T).astype(np.float64)        # fp32 accumulate of exact bf16 products
    un, vn = np.linalg.norm(users, axis=1), np.linalg.norm(items, axis=1)
    ratio = np.abs(coarse - exact) / (un[:, None] * vn[None, :])
    c = 2.0 ** -8 + 2.0 ** -18 + d * 2.0 ** -24
    eps = c * un * vn.max()                                                  # per-user margin with the global max item norm
    eps_item = c * un[:, None] * vn[None, :]                                 # per-item margin (needs item norms in the shadow)
    kth = np.sort(coarse, axis=1)[:, -k]
    cand = coarse >= (kth - 2 * eps)[:, None]
    cand_item = coarse + eps_item >= (np.sort(coarse - eps_item, axis=1)[:, -k])[:, None]
    top = np.argsort(-exact, axis=1)[:, :k]
    ok = all(cand[u, top[u]].all() for u in range(n_users))
    ok_item = all(cand_item[u, top[u]].all() for u in range(n_users))
    print(f"{kind:12s} items={n_items} d={d} users={n_users} k={k}: max err/(|u||v|) = {ratio.max():.2e} (bound c = {c:.2e}); "
          f"top-k inside candidates: {ok} / {ok_item}; candidates per user: global-norm margin mean {cand.sum(1).mean():.0f} max {cand.sum(1).max()}, "
          f"per-item margin mean {cand_item.sum(1).mean():.0f} max {cand_item.sum(1).max()}")
    assert ratio.max() <= c and ok and ok_item


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    for kind in ("gauss", "trained-like"):
        for d in (64, 256, 512):
            run(n, d, 16, 10, kind)


def append_form(n_items, d, n_users, k, prefix, kind, seed=0):
    """Second form: how many items pass `coarse >= seeded bound - 2 eps` (the candidate buffer load per user) when the bound
    is the k-th best coarse score of a catalog prefix (the threshold-seeding phase of the kernel)."""
    rng = np.random.default_rng(seed)
    if kind == "gauss":
        items = rng.normal(size=(n_items, d)).astype(np.float32)
        users = rng.normal(size=(n_users, d)).astype(np.float32)
    else:
        basis = rng.normal(size=(16, d)).astype(np.float32)
        items = (rng.normal(size=(n_items, 16)).astype(np.float32) @ basis) * (0.2 + rng.pareto(3.0, (n_items, 1))).astype(np.float32)
        items += 0.1 * rng.normal(size=(n_items, d)).astype(np.float32)
        users = rng.normal(size=(n_users, 16)).astype(np.float32) @ basis
    coarse = to_bf16(users) @ to_bf16(items).T
    c = 2.0 ** -8 + 2.0 ** -18 + d * 2.0 ** -24
    eps = c * np.linalg.norm(users, axis=1) * np.linalg.norm(items, axis=1).max()
    seed_thr = np.sort(coarse[:, :prefix], axis=1)[:, -k]
    final_thr = np.sort(coarse, axis=1)[:, -k]
    upper = (coarse >= (seed_thr - 2 * eps)[:, None]).sum(1)     # if the bound never improved after seeding
    lower = (coarse >= (final_thr - 2 * eps)[:, None]).sum(1)    # with the final bound from the start
    eps_item = c * np.linalg.norm(users, axis=1)[:, None] * np.linalg.norm(items, axis=1)[None, :]     # per-item margin
    upper_item = (coarse + 2 * eps_item >= seed_thr[:, None]).sum(1)
    print(f"append {kind:12s} items={n_items} d={d} prefix={prefix}: candidates per user between {lower.mean():.0f} (final bound) and "
          f"{upper.mean():.0f} mean / {upper.max()} max (seeded bound only, global-norm margin); per-item margin with the seeded bound: "
          f"{upper_item.mean():.0f} mean / {upper_item.max()} max")


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "append":
    for kind in ("gauss", "trained-like"):
        for d in (64, 512):
            append_form(n, d, 8, 10, max(n // 76, 1000), kind)      # 65 k of 5 M = 1/76 of the catalog seeds the bound
