"""Pins `oracle/transformer_oracle.py` to vectors produced by the UNMODIFIED reference modules
(tests/golden/transformer_*.npz, generator: tests/golden/make_golden_transformer.py): logits, loss, every
parameter gradient, parameters after one and two Adam steps, and eval-mode session encodings, for SASRec
(all four losses, dot/cosine, causal / key-padding / merged masks), BERT4Rec (Pre-LN), eSASRec (LiGR x3
activations) and HSTU (all four relative-bias variants).  CPU-only.  fp32 tolerances: rtol 1e-4 / atol 1e-5
on logits and loss, rtol 2e-3 / atol 2e-6 on gradients (different but equivalent op order).
"""
import pytest
import torch

from conftest import list_transformer_golden, load_transformer_golden
from oracle import transformer_oracle as T

NAMES = list_transformer_golden()


@pytest.mark.parametrize("name", NAMES)
def test_forward_loss_grads_adam(name):
    cfg, p0, g_ref, p1_ref, p2_ref, batch, ex = load_transformer_golden(name)
    with torch.no_grad():
        if "logits" in ex:
            logits = T.batch_logits(cfg, p0, batch)
            torch.testing.assert_close(logits, ex["logits"], rtol=1e-4, atol=2e-5)
        enc = T.encode_sessions(cfg, p0, batch)
        torch.testing.assert_close(enc, ex["enc"], rtol=1e-4, atol=2e-5)
    loss, grads = T.loss_and_grads(cfg, p0, batch)
    assert abs(float(loss) - ex["loss"]) <= 1e-5 + 1e-5 * abs(ex["loss"])
    assert set(grads) == set(g_ref)
    for k in g_ref:
        torch.testing.assert_close(grads[k], g_ref[k], rtol=2e-3, atol=2e-6, msg=lambda m, k=k: f"{k}: {m}")
    # Adam restatement: fed with the REFERENCE's gradients it must reproduce the reference's parameters.
    # (Fed with its own gradients it cannot, for parameters whose true gradient is zero — e.g. the key bias of
    # softmax attention: there |g| ~ 1e-9 is rounding noise and Adam's g / (|g| + eps) amplifies it to O(lr).)
    adam = T.AdamState(lr=cfg["lr"])
    p1 = adam.step(p0, g_ref)
    for k in p1_ref:
        torch.testing.assert_close(p1[k], p1_ref[k], rtol=1e-6, atol=1e-7, msg=lambda m, k=k: f"p1 {k}: {m}")
    loss2, grads2 = T.loss_and_grads(cfg, p1_ref, batch)
    assert abs(float(loss2) - ex["loss2"]) <= 2e-5 + 2e-5 * abs(ex["loss2"])
    p2 = adam.step(p1_ref, grads2)
    for k in p2_ref:
        if k not in g_ref:      # integer buffers of the item-feature structure: unchanged by the optimiser
            assert torch.equal(p2[k], p2_ref[k])
            continue
        solid = (grads2[k].abs() > 1e-6) & (g_ref[k].abs() > 1e-6)  # skip noise-dominated coordinates
        torch.testing.assert_close(p2[k][solid], p2_ref[k][solid], rtol=1e-4, atol=2e-5,
                                   msg=lambda m, k=k: f"p2 {k}: {m}")
