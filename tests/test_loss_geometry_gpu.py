"""Parity at the BENCHMARKED loss geometry (BASELINE configs C4 and C5-train): the sampled loss + the embedding lookup's backward + Adam
over a 1,000,001-row item table, N = 128 negatives, Zipf targets, M = 8,192 positions — the regime of 217 k-workgroup row reducers,
`pairs_scatter_kernel` and `adam_segs_kernel` over a 1–2 GB table that the bench legs run and the V = 600 .. 26,744 parity tests never enter.

HIP (`ops.embed` -> `ops.sampled_loss` -> backward -> `lightning.FlatAdam.step`) against the oracle's formulas (oracle/transformer_oracle.py:
`_l2norm`, `sampled_softmax_loss`, `AdamState` — similarity.py:92-100, lightning.py:207-218, hstu.py:701-702) evaluated with autograd on the
TOUCHED table rows only (the oracle cannot hold [M, 1 + N, d] gathers of the whole table cheaply; an untouched row's gradient must be
an exact zero and its parameter bit-identical after the step — checked on the device over all rows).  `home`: the table's gradient is
produced in its segment of the optimiser's flat gradient buffer (`ops._TABLE_GRAD_HOME`, the data-parallel layout) and the flat Adam
kernel `rt_adam_step` runs; else autograd's own tensor and `rt_adam_step_segments`."""
import numpy as np
import pytest
import torch

from oracle import transformer_oracle as T

pytestmark = pytest.mark.gpu

CASES = [
    ("C4_V1M_d256_cosine_t0.05", 1_000_001, 256, True, 0.05),      # hstu.py:701-702 / tutorial: cosine, logits_t 0.05, N = 128
    ("C5train_V1M_d512_dot", 1_000_001, 512, False, 1.0),
]


def _inputs(V, d, M, N, L, seed):
    from rectools_amd import synth

    rng = np.random.default_rng(seed)
    zipf = synth.zipf_item_sampler(V - 1, rng)
    x = zipf(M).astype(np.int64) + 1
    y = zipf(M).astype(np.int64) + 1
    # left padding: the first positions of some sessions are pads (id 0, no target), as the collate cuts them (sasrec.py:86-104)
    x2, y2 = x.reshape(-1, L), y.reshape(-1, L)
    for b in range(0, x2.shape[0], 3):
        k = int(rng.integers(1, L // 2))
        x2[b, :k] = 0
        y2[b, :k] = 0
    neg = rng.integers(1, V, size=(M, N)).astype(np.int64)           # negative_sampler.py:58-73: uniform over the real items, no rejection
    w = (0.5 + rng.random(M)).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(neg), torch.from_numpy(w)


@pytest.mark.parametrize("home", [False, True], ids=["own_gradient_tensor", "flat_buffer_home"])
@pytest.mark.parametrize("name,V,d,cosine,logits_t", CASES, ids=[c[0] for c in CASES])
def test_sampled_loss_embedding_backward_and_adam_over_a_million_row_table(name, V, d, cosine, logits_t, home):
    from rectools_amd import lightning as hl
    from rectools_amd import ops

    M, N, L, lr = 8192, 128, 128, 1e-3
    x, y, neg, w = _inputs(V, d, M, N, L, 7)
    g = torch.Generator().manual_seed(11)
    table0 = torch.randn(V, d, generator=g) * 0.05
    table0[0] = 0                                                    # the PAD row: zero at init, never trained (item_net.py:260-264)
    holder = torch.nn.Module()
    holder.weight = torch.nn.Parameter(table0.clone().cuda())
    opt = hl.FlatAdam(holder, lr=lr)
    table = holder.weight
    xd, yd, negd, wd = x.cuda(), y.cuda(), neg.cuda(), w.cuda()

    def forward_backward():
        opt.zero_grad()
        sess = ops.embed(table, None, xd, L, 1.0, 0.0)
        loss, logits = ops.sampled_loss(sess, table, yd, negd, wd, ops.LOSS_SAMPLED_SOFTMAX, cosine, logits_t)
        loss.backward()
        ops.join_side_streams()
        return loss.detach(), logits

    if home:
        forward_backward()
        opt.gather_gradients()                                       # a data-parallel step's pack registers the segment as the table's home
        assert table.data_ptr() in ops._TABLE_GRAD_HOME
    loss, logits = forward_backward()
    if home:
        seg = opt.flat_g[opt._offsets[0]:opt._offsets[0] + table.numel()]
        assert table.grad.data_ptr() == seg.data_ptr()               # produced in place: the pack skips it
    grad = table.grad.detach().clone()
    before = table.detach().clone()

    # ---- the oracle on the touched rows --------------------------------------------------------------------------------------------
    touched = torch.unique(torch.cat([x, y, neg.reshape(-1)]))
    remap = lambda ids: torch.searchsorted(touched, ids)             # noqa: E731
    sub = table0[touched].clone().requires_grad_(True)
    xs = sub[remap(x)]
    xs = torch.where((x != 0).unsqueeze(-1), xs, xs.detach())        # padding_idx = 0: no gradient through the lookup of a pad
    s, it = (T._l2norm(xs), T._l2norm(sub)) if cosine else (xs, sub)
    cand = torch.cat([remap(y).unsqueeze(-1), remap(neg)], dim=-1)   # [M, 1 + N]
    ref_logits = (it[cand] @ s.unsqueeze(-1)).squeeze(-1) / logits_t
    ref_loss = T.sampled_softmax_loss(ref_logits[None], y[None], w[None])
    ref_loss.backward()
    ref_grad = sub.grad.clone()
    assert int(touched[0]) == 0
    ref_grad[0] = 0                                                  # the PAD row receives nothing (oracle.loss_and_grads does the same)

    assert abs(float(loss) - float(ref_loss)) <= 5e-5 * abs(float(ref_loss)) + 5e-6, (float(loss), float(ref_loss))
    act = y != 0
    torch.testing.assert_close(logits.cpu()[act], ref_logits.detach()[act], rtol=2e-4, atol=2e-4 * float(ref_logits.detach().abs().max()))
    got = grad[touched.cuda()].cpu()
    scale = float(ref_grad.abs().max())
    torch.testing.assert_close(got, ref_grad, rtol=1e-2, atol=2e-5 * scale)
    # every row no candidate and no lookup touched: an exact zero (all 1,000,001 rows checked on the device)
    hit = torch.zeros(V, dtype=torch.bool, device="cuda")
    hit[touched.cuda()] = True
    assert float(grad[~hit].abs().max()) == 0.0
    assert float(grad[0].abs().max()) == 0.0

    # ---- one Adam step (lightning.py:214-218) ----------------------------------------------------------------------------------------
    if home:      # what FlatAdam.step(world_size > 1) runs behind the collective: the pack (skips the table) + the flat kernel
        fg = opt.gather_gradients()
        opt.step_count += 1
        opt._adam_flat(opt.flat_p, fg, opt.m, opt.v, (opt.step_count, lr, 0.9, 0.98, 1e-8, 1.0))
    else:
        opt.step()
    torch.cuda.synchronize()
    after = table.detach()
    got_after = after[touched.cuda()].cpu()
    # the update kernel itself: the oracle's Adam on the gradient the device produced (rounding of one fp32 update)
    want_k = T.AdamState(lr).step({"w": table0[touched]}, {"w": got})["w"]
    torch.testing.assert_close(got_after, want_k, rtol=0, atol=6e-8)      # (two ulps of a 0.25 .. 0.5 parameter)
    # end to end: the oracle's Adam on the oracle's gradient (the gradients sit near Adam's eps = 1e-8 here — M = 8,192 positions in the
    # normaliser — so a relative gradient error e moves the update by e * eps / (|g| + eps) of a step)
    want = T.AdamState(lr).step({"w": table0[touched]}, {"w": ref_grad})["w"]
    torch.testing.assert_close(got_after, want, rtol=0, atol=2e-2 * lr)
    assert float((got_after - table0[touched]).abs().max()) <= lr * 1.0001                  # |first update| <= lr
    assert bool(torch.equal(after[~hit], before[~hit]))                                      # untouched rows: bit-identical
