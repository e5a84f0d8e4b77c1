"""`lightning.NativeSasrecStep` (csrc/rt_step.hip: the stock packed SASRec training step behind one compiled call) against the autograd
path it restates — same entry points, order, streams and dropout draws — and its eligibility rules.  Two runs of EITHER path differ in
the last bits of the item table's gradient (the sampled loss ranks the pairs of a candidate with atomics, the row reducer sums them in
rank order) and of the bias gradients (partial column sums meet in atomicAdds: colsum_kernel) — scripts/debug/grad_repro.py; scripts/debug/native_step_diff.py
shows autograd vs autograd, compiled vs compiled and compiled vs autograd differing alike, ~3e-7 after 14 steps), so the comparison is
at that level: a wrong dropout stream, a missing gradient or a wrong Adam segment moves a parameter by the learning rate (4e-3) per step.
Reference: lightning.py:311-321 (training_step), sasrec.py:271-304, lightning.py:164-212 (sampled losses)."""
import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dataset(seed=3, n_users=260, n_items=180, n=9000):
    from rectools_amd.dataset import Dataset

    rng = np.random.default_rng(seed)
    df = pd.DataFrame({"user_id": rng.integers(0, n_users, n), "item_id": rng.integers(0, n_items, n) + 100, "weight": 1.0,
                       "datetime": pd.to_datetime("2022-01-01") + pd.to_timedelta(rng.integers(0, 500_000, n), unit="m")})
    return Dataset.construct(df)


def _run(monkeypatch, native, steps, **kw):
    from rectools_amd import ops
    from rectools_amd.models import SASRecModel

    monkeypatch.setenv("RT_NATIVE_STEP", "1" if native else "0")
    ops.RNG.__init__(0)
    args = dict(n_factors=64, n_blocks=2, n_heads=2, session_max_len=32, lr=0.004, batch_size=48, dropout_rate=0.2, loss="sampled_softmax",
                n_negatives=7, seed=11, epochs=1)
    args.update(kw)
    m = SASRecModel(**args)
    m._build_model_from_dataset(_dataset())
    loop = m.training_loop()
    m.lightning_model.train()
    loop.begin_epoch(0)
    losses = [loop.step() for _ in range(steps)]       # (crosses the epoch's end: a short last batch, a re-plan)
    torch.cuda.synchronize()
    return loop, m, (torch.stack([l.detach() for l in losses]).cpu() if losses else torch.zeros(0)), m.optimizer


def _first_params(monkeypatch, kw):
    """The parameters before any step (what a step that did nothing would leave)."""
    _, _, _, opt = _run(monkeypatch, False, 0, **kw)
    return opt.flat_p


@pytest.mark.parametrize("kw", [
    {},                                                          # the benchmarked form: sampled softmax, dropout, positional rows, pad keys
    {"loss": "BCE", "dropout_rate": 0.0},
    {"loss": "gBCE", "n_negatives": 5},
    {"use_pos_emb": False, "n_blocks": 1},
    {"use_key_padding_mask": True, "n_blocks": 3, "n_heads": 1},
], ids=["sampled_softmax", "bce_p0", "gbce", "no_pos_1_block", "keypad_3_blocks"])
def test_native_step_is_the_autograd_step(monkeypatch, kw):
    steps = 14
    loop_n, m_n, loss_n, opt_n = _run(monkeypatch, True, steps, **kw)
    assert loop_n._native is not None, "the stock configuration must take the compiled step"
    assert all(p.grad is None for p in opt_n.params)             # its gradients live in the step's arena
    loop_a, m_a, loss_a, opt_a = _run(monkeypatch, False, steps, **kw)
    assert loop_a._native is None
    torch.testing.assert_close(loss_n, loss_a, rtol=1e-5, atol=0)
    assert opt_n.step_count == opt_a.step_count == steps
    _, m_a2, _, _ = _run(monkeypatch, False, steps, **kw)          # the yardstick: a second autograd run
    _assert_same_parameters(m_n, m_a, m_a2)
    moved = float((opt_n.flat_p - _first_params(monkeypatch, kw)).abs().max())
    assert moved > 1e-2, moved                                   # (14 Adam steps of 4e-3: the comparison above is not between two idle models)


def _far_apart(m_n, m_a):
    """-> (elements further apart than rtol 1e-4 / atol 2e-5, all elements, the largest difference), over every parameter but the KEY
    bias: that one has no gradient (it shifts every logit of a query alike), what the kernels produce for it is rounding noise around
    zero, and Adam turns noise into steps of the learning rate — two runs of the same path disagree there too (as
    tests/test_packed_gpu.py::test_packed_train_loop_takes_the_steps_of_the_padded_loop)."""
    sd_n, sd_a = m_n.torch_model.state_dict(), m_a.torch_model.state_dict()
    assert sd_n.keys() == sd_a.keys()
    far = total = 0
    worst = 0.0
    for k, a in sd_n.items():
        b = sd_a[k]
        if k.endswith("in_proj_bias"):
            d = a.numel() // 3
            a, b = torch.cat([a[:d], a[2 * d:]]), torch.cat([b[:d], b[2 * d:]])
        diff = (a - b).abs()
        far += int((diff > 2e-5 + 1e-4 * b.abs()).sum())
        total += a.numel()
        worst = max(worst, float(diff.max()))
    return far, total, worst


def _assert_same_parameters(m_n, m_a, m_a2=None):
    """The compiled run against an autograd run at the run-to-run noise level.  Any element whose gradient passes close to zero in some
    step sits in an ill-conditioned spot of Adam's m / sqrt(v): a percent of the elements may end further apart than the tolerance — between
    two autograd runs (m_a, m_a2: the yardstick when given) as between a compiled and an autograd run.  A systematic difference (a dropout
    stream, a missing gradient, a wrong Adam segment) moves EVERY element of a tensor by ~ the learning rate (4e-3) per step."""
    far, total, worst = _far_apart(m_n, m_a)
    assert worst < 1e-3, worst
    assert far <= total // 50, (far, total)
    if m_a2 is not None:
        far0, _, worst0 = _far_apart(m_a, m_a2)
        assert far <= 3 * far0 + total // 500, f"compiled vs autograd: {far} elements apart (max {worst}); autograd vs autograd: {far0} (max {worst0})"


def test_native_step_cosine_with_temperature(monkeypatch):
    """The C4 loss geometry (cosine similarity, logits_t 0.05) through the compiled step."""
    kw = dict(similarity_module_kwargs={"distance": "cosine"}, lightning_module_kwargs={"logits_t": 0.05})
    loop_n, m_n, loss_n, _ = _run(monkeypatch, True, 8, **kw)
    assert loop_n._native is not None and loop_n.lm.cosine and abs(loop_n.lm.logits_t - 0.05) < 1e-9
    _, m_a, loss_a, _ = _run(monkeypatch, False, 8, **kw)
    torch.testing.assert_close(loss_n, loss_a, rtol=1e-5, atol=0)
    _assert_same_parameters(m_n, m_a)


def test_what_the_compiled_step_does_not_restate_keeps_the_autograd_path(monkeypatch):
    from rectools_amd import lightning as hl
    from rectools_amd import ops

    # the full-catalog softmax
    loop, *_ = _run(monkeypatch, True, 2, loss="softmax", n_negatives=None)
    assert loop._native is None

    # a subclassed loss module with its own loss
    class MyLoss(hl.TransformerLossModule):
        def _loss_from_sessions(self, table, sess2d, y, w, negatives, n_targets=None):
            loss, logits = super()._loss_from_sessions(table, sess2d, y, w, negatives)
            return loss * 2.0, logits

    loop, _, losses, _ = _run(monkeypatch, True, 2, lightning_module_type=MyLoss)
    assert loop._native is None and torch.isfinite(losses).all()

    # instrumentation (bench.py's kernel breakdown itemises the autograd path's calls): the step falls back for as long as it is on
    loop, m, _, opt = _run(monkeypatch, True, 2)
    assert loop._native is not None
    ops.start_timing()
    try:
        loop.step()
        assert any(p.grad is not None for p in opt.params)       # the autograd path ran
    finally:
        ops.stop_timing()
    loop.step()
    torch.cuda.synchronize()
    assert opt.step_count == 4


def test_native_step_follows_moved_buffers(monkeypatch):
    """The descriptor's pointers are re-read when the optimiser's buffers move (a checkpoint load, `.to()`): here the moments are
    swapped for clones between steps — the steps behind the swap update the NEW buffers, on both paths alike."""
    models = {}
    for native in (True, False):
        loop, m, _, opt = _run(monkeypatch, native, 5)
        old_m, old_v = opt.m, opt.v
        opt.m, opt.v = opt.m.clone(), opt.v.clone()
        frozen = old_m.clone()
        for _ in range(4):
            loop.step()
        torch.cuda.synchronize()
        assert (loop._native is not None) == native and opt.step_count == 9
        assert torch.equal(old_m, frozen) and not torch.equal(opt.m, frozen)
        models[native] = m
        del old_v
    _assert_same_parameters(models[True], models[False])


@pytest.mark.parametrize("seed", range(10))
def test_compiled_step_fuzz_against_autograd(monkeypatch, seed):
    """Random stock configurations (width, heads, window, depth, negatives, batch size, loss, dropout, positional rows, key-padding
    masks): ONE step from the same parameters, batch and dropout streams through autograd and through the compiled call.  Same entry
    points, same order, same arguments: the loss and every weight MATRIX's gradient agree BIT FOR BIT; the item table's gradient is summed
    in the order the loss's atomics ranked a candidate's pairs and the bias gradients' partial column sums meet in atomicAdds (last bits);
    the Adam step behind them likewise."""
    import random

    from rectools_amd import lightning as hl
    from rectools_amd import ops
    from rectools_amd.models import SASRecModel

    r = random.Random(1000 + seed)
    d = r.choice([32, 64, 128, 256])
    hd = r.choice([h for h in (32, 64) if d % h == 0])
    kw = dict(n_factors=d, n_heads=d // hd, session_max_len=r.choice([16, 50, 200]), n_blocks=r.choice([1, 2, 4]), n_negatives=r.choice([1, 16, 128]),
              batch_size=r.choice([7, 32, 128]), loss=r.choice(["BCE", "gBCE", "sampled_softmax"]), dropout_rate=r.choice([0.0, 0.3]),
              use_pos_emb=r.choice([True, False]), use_key_padding_mask=r.choice([True, False]))
    monkeypatch.setenv("RT_NATIVE_STEP", "0")
    m = SASRecModel(lr=0.004, seed=11, epochs=1, **kw)
    m._build_model_from_dataset(_dataset())
    loop = m.training_loop()
    if not loop.packed:
        pytest.skip(f"no packed loop for {kw}")
    m.lightning_model.train()
    loop.begin_epoch(0)
    for _ in range(r.choice([0, 3])):      # (3: not the first batch, moments in place)
        loop.step()
    batch = loop._cut_batch()
    opt, lm = loop.opt, loop.lm
    step0 = ops.RNG.step + 1

    def rewind():
        ops.RNG.step, ops.RNG._stream = step0, 0

    rewind()
    opt.zero_grad()
    loss_a = lm.training_loss_packed(batch)
    loss_a.backward()
    ops.join_side_streams()
    grads_a = [None if p.grad is None else p.grad.detach().clone() for p in opt.params]
    monkeypatch.setenv("RT_NATIVE_STEP", "1")
    native = hl.NativeSasrecStep.plan(lm, opt)
    assert native is not None and native.ready(batch), kw
    rewind()
    loss_n = native.forward_backward(batch)
    grads_n = native.gradients()
    assert torch.equal(loss_n.detach(), loss_a.detach()), (kw, float(loss_n), float(loss_a))
    table = lm.torch_model.item_model.table
    names = {id(p): n for n, p in lm.torch_model.named_parameters()}
    for p, ga, gn in zip(opt.params, grads_a, grads_n):
        assert (ga is None) == (gn is None), names[id(p)]
        if ga is None:
            continue
        if p is table:
            torch.testing.assert_close(gn, ga, rtol=0, atol=2e-6 * float(ga.abs().max()) + 1e-12, msg=lambda s: f"{kw}: table gradient: {s}")
        elif p.dim() == 1:
            # bias gradients are column sums whose partial sums meet in an atomicAdd (colsum_kernel, rt_gemm.hip): last bits vary with the
            # order the workgroups arrive in; the KEY bias third has no gradient at all (cancellation noise around zero)
            torch.testing.assert_close(gn, ga, rtol=0, atol=1e-6 * float(ga.abs().max()) + 1e-12, msg=lambda s, p=p: f"{kw}: {names[id(p)]}: {s}")
        else:
            bad = (gn != ga).nonzero()
            assert bad.numel() == 0, (f"{kw}: {names[id(p)]}: {bad.shape[0]} of {ga.numel()} elements differ, "
                                      f"max {float((gn - ga).abs().max()):.2e} of {float(ga.abs().max()):.2e}")
    # the Adam step: the compiled phase 2 against FlatAdam.step on copies of the same state
    state = [t.clone() for t in (opt.flat_p, opt.m, opt.v)]
    count = opt.step_count
    native.adam()
    torch.cuda.synchronize()
    after_n = [t.clone() for t in (opt.flat_p, opt.m, opt.v)]
    for t, s0 in zip((opt.flat_p, opt.m, opt.v), state):
        t.copy_(s0)
    opt.step_count = count
    opt.step(1)          # (reads p.grad: the autograd pass's tensors)
    torch.cuda.synchronize()
    for i, p in enumerate(opt.params):      # matrices but the table: the same bits; biases and the table: to their gradients' last bits
        lo, hi = opt._offsets[i], opt._offsets[i] + p.numel()
        for t_n, t_a in zip(after_n, (opt.flat_p, opt.m, opt.v)):
            if p.dim() == 2 and p is not table:
                assert torch.equal(t_n[lo:hi], t_a[lo:hi]), (kw, names[id(p)])
            else:
                torch.testing.assert_close(t_n[lo:hi], t_a[lo:hi], rtol=1e-3, atol=1e-5)
