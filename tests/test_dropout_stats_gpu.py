"""Dropout masks are where parity with the reference is DISTRIBUTIONAL (nn.Dropout draws from torch's Philox stream, `sasrec.py:225-229`,
`torch_backbone.py:247`, `nn.MultiheadAttention(dropout=...)`): the engine regenerates its masks from counter hashes (element masks:
`csrc/rt_common.h` fmix32 + lowbias32, 16 bits per element, four elements per hash pair; attention probabilities: `csrc/rt_varlen.h`).
What a dropout mask has to be is Bernoulli(1 - p) per element and independent of every other element, stream and step.  Tested here, on
the masks the KERNELS produce (not a restatement of the hash), with the harness of tests/test_negative_sampler_gpu.py:

  * keep rate against 1 - p (z-score; p is quantised to 16 bits, `rt_drop_thr16`),
  * the 16 keep / drop patterns of a float4 group against the product distribution (chi-square, 15 dof) — the four fields of a group
    come from two dependent hash words,
  * correlation between the fields of a group, between neighbouring elements and neighbouring groups, between two streams of one step,
    between the same stream of two steps, between two model seeds,
  * the same for the attention-probability masks of the padded and the packed (bf16-plane) attention kernels.
Bounds: 5 sigma of the null distribution (false-alarm rate < 1e-5 per run over all assertions)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_ROWS, WIDTH = 4096, 1024          # 4.2 M elements per mask: one C2 activation array is 3.4 M


def _element_mask(p, step, stream, seed=20240531):
    """[N_ROWS * WIDTH] bool: the mask `rt_act_dropout_fwd` applies for (seed, step, stream)."""
    from rectools_amd import ops

    ops.RNG.seed, ops.RNG.step, ops.RNG._stream = seed, step, stream - 1
    x = torch.ones((N_ROWS, WIDTH), device="cuda")
    y = ops.dropout(x, p)
    kept = y != 0
    torch.testing.assert_close(y[kept], torch.full_like(y[kept], 1.0 / (1.0 - p)), rtol=1e-6, atol=0)     # kept values are scaled
    return kept.reshape(-1)


def _corr(a, b):
    a, b = a.double(), b.double()
    return float(torch.corrcoef(torch.stack([a, b]))[0, 1])


def _assert_bernoulli(kept, q, what):
    n = kept.numel()
    z = (float(kept.double().mean()) - q) / np.sqrt(q * (1 - q) / n)
    assert abs(z) < 5.0, f"{what}: keep rate {float(kept.double().mean()):.6f} vs {q:.6f} (z = {z:.2f})"


@pytest.mark.parametrize("p", [0.1, 0.2, 0.5])
def test_element_masks_are_bernoulli_and_independent(p):
    q = 1.0 - int(np.float32(p) * np.float32(65536.0)) / 65536.0          # the 16-bit threshold the kernels compare with
    m = _element_mask(p, step=7, stream=1)
    n = m.numel()
    bound = 5.0 / np.sqrt(n)
    _assert_bernoulli(m, q, f"p={p}")
    g = m.view(-1, 4)                                                     # the four fields of a float4 group
    pattern = (g.long() * torch.tensor([1, 2, 4, 8], device="cuda")).sum(1)
    observed = torch.bincount(pattern, minlength=16).double().cpu().numpy()
    prob = np.array([np.prod([q if (i >> j) & 1 else 1 - q for j in range(4)]) for i in range(16)])
    chi2 = float(((observed - g.shape[0] * prob) ** 2 / (g.shape[0] * prob)).sum())
    assert abs(chi2 - 15) < 5.0 * np.sqrt(2.0 * 15) + 15, f"group patterns: chi2 = {chi2:.1f} (15 dof)"
    for i in range(4):
        for j in range(i + 1, 4):
            assert abs(_corr(g[:, i], g[:, j])) < 5.0 / np.sqrt(g.shape[0]), f"fields {i}, {j} of a group"
    assert abs(_corr(m[:-1], m[1:])) < bound, "neighbouring elements"
    assert abs(_corr(m[:-4], m[4:])) < bound, "same field of neighbouring groups"
    assert abs(_corr(m[:-WIDTH], m[WIDTH:])) < bound, "same column of neighbouring rows"
    # the same mask again (forward and backward regenerate it), another stream, another step, another seed
    assert torch.equal(m, _element_mask(p, step=7, stream=1))
    for what, other in (("stream", _element_mask(p, 7, 2)), ("step", _element_mask(p, 8, 1)), ("far step", _element_mask(p, 5000, 1)),
                        ("seed", _element_mask(p, 7, 1, seed=20240532))):
        _assert_bernoulli(other, q, what)
        assert abs(_corr(m, other)) < bound, f"masks of two {what}s are correlated"


def _attention_masks(kind, p, step):
    """Keep / drop decisions of the attention-probability dropout, read off the output: with q = k = 0 every allowed key of a query
    has probability 1 / (number of allowed keys), with v = identity column j of the output row i is P[i, j] * keep / (1 - p)."""
    from rectools_amd import ops

    B, H, L = 192, 2, 64
    hd = L
    d = H * hd
    ops.RNG.seed, ops.RNG.step, ops.RNG._stream = 99, step, 0
    eye = torch.eye(L, device="cuda").repeat(B, H)                          # [B*L, H*hd]: v of (b, i, h) = e_i
    zeros = torch.zeros((B * L, d), device="cuda")
    if kind == "padded":
        ids = torch.ones((B, L), dtype=torch.int64, device="cuda")
        o = ops.mha(zeros, zeros, eye, ids, B, H, L, True, False, p)
    else:
        cu = torch.arange(B + 1, dtype=torch.int64, device="cuda") * L      # full sessions: no pad keys in the window
        o = ops.mha_varlen(zeros, torch.cat([zeros, eye], 1).contiguous(), None, None, cu, B, H, L, p)
    o = o.view(B, L, H, hd).permute(0, 2, 1, 3)                             # [B, H, query, key]
    allowed = torch.tril(torch.ones(L, L, dtype=torch.bool, device="cuda"))
    kept = (o != 0)[:, :, allowed]                                          # [B, H, L (L + 1) / 2]
    counts = torch.arange(1, L + 1, device="cuda", dtype=torch.float32)     # allowed keys of query i
    want = (1.0 / counts / (1.0 - p))[:, None].expand(L, L)[allowed]
    got = o[:, :, allowed]
    torch.testing.assert_close(got[kept], want.expand_as(got)[kept], rtol=2e-3, atol=0)
    return kept


@pytest.mark.parametrize("kind", ["padded", "packed"])
@pytest.mark.parametrize("p", [0.2, 0.5])
def test_attention_probability_masks_are_bernoulli_and_independent(kind, p):
    m = _attention_masks(kind, p, step=3)                                   # 192 x 2 x 2080 = 0.8 M decisions
    n = m.numel()
    bound = 5.0 / np.sqrt(n)
    q = float(m.double().mean())
    assert abs(q - (1.0 - p)) < 5.0 * np.sqrt(p * (1 - p) / n) + 2.0 ** -15, (q, 1.0 - p)     # (thresholds are 16-bit)
    flat = m.reshape(-1)
    assert abs(_corr(flat[:-1], flat[1:])) < bound, "neighbouring (query, key) pairs"
    assert abs(_corr(m[:, 0].reshape(-1), m[:, 1].reshape(-1))) < 5.0 / np.sqrt(n / 2), "the two heads of a session"
    assert abs(_corr(m[:-1].reshape(-1), m[1:].reshape(-1))) < bound, "neighbouring sessions"
    other = _attention_masks(kind, p, step=4)
    assert abs(_corr(flat, other.reshape(-1))) < bound, "two steps"
    assert torch.equal(m, _attention_masks(kind, p, step=3))
