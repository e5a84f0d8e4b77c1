"""End-to-end TRAINED-model parity at BASELINE config 1 (SURVEY.md §8c "end-to-end item ids + ranks after training",
test_sasrec.py:163-305): the HIP engine replays one epoch of the reference — same initial state_dict, same batch sequence
(tests/golden/trajectory_c1.npz, recorded from the unmodified reference by tests/golden/make_golden_trajectory.py), dropout 0,
full softmax, Adam(1e-3, (0.9, 0.98)) — and must follow the reference's loss curve step by step and end with the same
top-10 recommendations.

Tolerances: per-step loss rtol 1e-4 (fp32; 48 Adam steps compound the summation-order differences of every kernel);
final top-10: >= 99 % of the (user, item) pairs shared with the reference, scores rtol 2e-3.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import transformer_oracle as T

FIX = os.path.join(GOLDEN_DIR, "trajectory_c1.npz")
CFG = dict(V=3706, L=50, d=64, H=4, n_blocks=1, N=None, loss="softmax", dist="dot", logits_t=1.0, causal=True, keypad=False,
           layers="sasrec", n_extra=1, gbce_t=0.2, lr=1e-3)


def _load():
    z = np.load(FIX)
    p0 = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("p0/")}
    batches = [(z["x"][i], z["y"][i]) for i in range(z["x"].shape[0])]
    if "x_last" in z.files:
        batches.append((z["x_last"], z["y_last"]))
    return z, p0, batches


def _batch(x, y):
    y = torch.from_numpy(y.astype(np.int64))
    return {"x": torch.from_numpy(x.astype(np.int64)), "y": y, "yw": (y != 0).float()}


def test_oracle_follows_the_reference_trajectory():
    """CPU: the oracle (restated step + Adam) against the recorded reference losses for the first 6 steps — pins the
    oracle's optimiser loop, not only single steps."""
    z, p0, batches = _load()
    assert int(z["n_tokens"]) == CFG["V"] + 1 and len(batches) == len(z["loss"]) == 48
    params = {k: v.clone() for k, v in p0.items()}
    adam = T.AdamState(lr=CFG["lr"])
    for s in range(6):
        b = _batch(*batches[s])
        loss, grads = T.loss_and_grads(dict(CFG, B=b["x"].shape[0]), params, b)
        assert abs(float(loss) - z["loss"][s]) <= 2e-5 * z["loss"][s], (s, float(loss), z["loss"][s])
        params = adam.step(params, grads)


@pytest.mark.gpu
def test_engine_follows_the_reference_trajectory_and_recommends_the_same_items():
    from rectools_amd import lightning as hl
    from rectools_amd import ops
    from rectools_amd.rank import DeviceCSR, HipRanker
    from test_transformer_gpu import build_hip_model

    z, p0, batches = _load()
    lm = build_hip_model(dict(CFG, N=None), p0)
    lm.train()
    opt = hl.FlatAdam(lm.torch_model, lr=CFG["lr"], betas=(0.9, 0.98))
    losses = []
    for x, y in batches:
        batch = {k: v.cuda() for k, v in _batch(x, y).items()}
        ops.RNG.next_step()
        opt.zero_grad()
        loss = lm.training_loss(batch)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(losses, z["loss"], rtol=1e-4)
    # recommend(users[:512], k=10, filter_viewed=True) of the trained model, on the reference's own recommend batches
    lm.eval()
    with torch.no_grad():
        table = lm.torch_model.item_model.get_all_embeddings()
        rec_x = torch.from_numpy(z["rec_x"].astype(np.int64)).cuda()
        users = torch.cat([lm.torch_model.encode_sessions({"x": rec_x[i:i + 256]}, table)[:, -1, :] for i in range(0, len(rec_x), 256)])
    U = users.shape[0]
    filt = DeviceCSR(torch.from_numpy(z["filt_indptr"]).cuda(), torch.from_numpy(z["filt_indices"]).cuda(), (U, int(z["n_tokens"])))
    ranker = HipRanker("dot", "cuda", users, table)
    ids, scores, counts, _ = ranker.rank_device(np.arange(U), k=10, filter_pairs_csr=filt,
                                                sorted_object_whitelist=np.arange(1, int(z["n_tokens"])))
    ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
    assert bool((counts.cpu().numpy() == 10).all())
    ref_ids, ref_scores = z["rec_items"], z["rec_scores"]
    shared = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(ids, ref_ids))
    same_rank = float((ids == ref_ids).mean())
    loss_err = float(np.max(np.abs(np.asarray(losses) - z["loss"]) / np.abs(z["loss"])))
    # the actual numbers, so that a slide from 99.9 % to 99.0 % is visible in the test output (pytest -s; profiles/r3_parity_numbers.txt)
    print(f"[trajectory C1] 48 steps: max relative loss error {loss_err:.2e}; trained model vs the reference's frames: "
          f"{shared / ref_ids.size:.4%} of the (user, item) top-10 pairs shared, {same_rank:.4%} at the same rank")
    assert shared >= 0.99 * ref_ids.size, shared / ref_ids.size
    assert same_rank >= 0.97, same_rank
    m = ids == ref_ids
    np.testing.assert_allclose(scores[m], ref_scores[m], rtol=2e-3, atol=2e-3)
