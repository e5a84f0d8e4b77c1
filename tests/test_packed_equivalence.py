"""Packed (padding-free) SASRec stack == the reference's padded computation, on the CPU oracle.

The reference runs every block on the full left-padded [B, L] window (sasrec.py:186-231, :300).  Pad QUERY rows never reach a
real row or the loss; pad KEY rows are the same in every session and block (the block input is masked to 0, so K_pad = b_k and
V_pad = b_v) and, with left padding + a causal mask, every real query sees all `n_pad` of them.  So attention over the real keys
plus ONE virtual key per query — logit q.b_k / sqrt(hd), value b_v, multiplicity n_pad — reproduces the padded softmax.  This test
pins that argument (outputs at the real positions and every parameter gradient of a loss over them) before any kernel is built on
it; the packed side is written with plain tensor ops per session.
"""
import math

import pytest
import torch

from oracle import transformer_oracle as T


def _params(d, H, n_blocks, L, V, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: (torch.randn(*s, generator=g, dtype=torch.float64) * 0.3).requires_grad_(True)   # noqa: E731
    p = {"emb": r(V, d), T.POS_EMB: r(L, d)}
    for i in range(n_blocks):
        pre = f"transformer_layers.transformer_blocks.{i}."
        for name, shape in (("q_layer_norm.weight", (d,)), ("q_layer_norm.bias", (d,)), ("ff_layer_norm.weight", (d,)),
                            ("ff_layer_norm.bias", (d,)), ("multi_head_attn.in_proj_weight", (3 * d, d)),
                            ("multi_head_attn.in_proj_bias", (3 * d,)), ("multi_head_attn.out_proj.weight", (d, d)),
                            ("multi_head_attn.out_proj.bias", (d,)), ("feed_forward.ff_linear_1.weight", (d, d)),
                            ("feed_forward.ff_linear_1.bias", (d,)), ("feed_forward.ff_linear_2.weight", (d, d)),
                            ("feed_forward.ff_linear_2.bias", (d,))):
            p[pre + name] = r(*shape)
    p["transformer_layers.last_layernorm.weight"], p["transformer_layers.last_layernorm.bias"] = r(d), r(d)
    return p


def _padded(p, x, n_blocks, H, keypad=False):
    """The reference-shaped computation (oracle): embeddings + inverse positions, SASRec blocks on the padded window."""
    L = x.shape[1]
    seqs = p["emb"][x] + p[T.POS_EMB][torch.arange(L - 1, -1, -1)][None]
    tl = (x != 0).unsqueeze(-1).to(seqs.dtype)
    mask = T.attention_mask(x, True, keypad).to(seqs.dtype)
    return T.sasrec_layers(seqs, tl, mask, p, n_blocks, H)


def _packed(p, x, n_blocks, H, keypad=False):
    """Only the real rows of every session; the pads enter as one virtual key with multiplicity n_pad (none under key-padding
    masks: the reference hides them from every real query, torch_backbone.py:254)."""
    B, L = x.shape
    d = p["emb"].shape[1]
    hd = d // H
    outs = []
    for b in range(B):
        ids = x[b][x[b] != 0]
        n, n_pad = len(ids), (0 if keypad else L - len(ids))
        if n == 0:
            outs.append(torch.zeros(0, d, dtype=p["emb"].dtype))
            continue
        s = p["emb"][ids] + p[T.POS_EMB][torch.arange(n - 1, -1, -1)]          # position = distance from the end
        for i in range(n_blocks):
            pre = f"transformer_layers.transformer_blocks.{i}."
            w, bias = p[pre + "multi_head_attn.in_proj_weight"], p[pre + "multi_head_attn.in_proj_bias"]
            qn = T.layer_norm(s, p[pre + "q_layer_norm.weight"], p[pre + "q_layer_norm.bias"], 1e-5)
            q = (qn @ w[:d].T + bias[:d]).view(n, H, hd).transpose(0, 1)          # [H, n, hd]
            k = (s @ w[d:2 * d].T + bias[d:2 * d]).view(n, H, hd).transpose(0, 1)
            v = (s @ w[2 * d:].T + bias[2 * d:]).view(n, H, hd).transpose(0, 1)
            bk, bv = bias[d:2 * d].view(H, 1, hd), bias[2 * d:].view(H, 1, hd)
            sc = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
            sc = sc.masked_fill(~torch.tril(torch.ones(n, n, dtype=torch.bool)), float("-inf"))
            sp = (q * bk).sum(-1, keepdim=True) / math.sqrt(hd)                    # the virtual pad key's logit, [H, n, 1]
            m = torch.maximum(sc.max(-1, keepdim=True).values, sp)
            e, ep = torch.exp(sc - m), n_pad * torch.exp(sp - m)
            att = (e @ v + ep * bv) / (e.sum(-1, keepdim=True) + ep)
            att = att.transpose(0, 1).reshape(n, d) @ p[pre + "multi_head_attn.out_proj.weight"].T \
                + p[pre + "multi_head_attn.out_proj.bias"]
            s = qn + att
            f = T.layer_norm(s, p[pre + "ff_layer_norm.weight"], p[pre + "ff_layer_norm.bias"], 1e-5)
            s = T.ffn(f, p, pre + "feed_forward.", "relu") + f
        outs.append(T.layer_norm(s, p["transformer_layers.last_layernorm.weight"],
                                 p["transformer_layers.last_layernorm.bias"], 1e-8))
    return outs


@pytest.mark.parametrize("keypad", [False, True])
@pytest.mark.parametrize("seed,H,n_blocks", [(0, 2, 2), (1, 1, 1), (2, 4, 3)])
def test_packed_sasrec_stack_equals_padded_reference(seed, H, n_blocks, keypad):
    d, L, V, B = 16, 12, 30, 6
    g = torch.Generator().manual_seed(100 + seed)
    lens = [L, 1, 5, 0, 9, 2]                                      # full, single item, typical, empty, ...
    x = torch.zeros(B, L, dtype=torch.int64)
    for b, n in enumerate(lens):
        if n:
            x[b, L - n:] = torch.randint(1, V, (n,), generator=g)
    p = _params(d, H, n_blocks, L, V, seed)
    gout = torch.randn(B, L, d, generator=g, dtype=torch.float64)

    full = _padded(p, x, n_blocks, H, keypad)
    real = x != 0
    (full * gout)[real].sum().backward()
    grads_padded = {k: v.grad.clone() for k, v in p.items()}
    for v in p.values():
        v.grad = None

    packed = _packed(p, x, n_blocks, H, keypad)
    loss = sum((o * gout[b][real[b]]).sum() for b, o in enumerate(packed))
    loss.backward()
    for b, o in enumerate(packed):
        torch.testing.assert_close(o, full[b][real[b]], rtol=1e-10, atol=1e-10, msg=f"session {b}")
    for k, v in p.items():
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        torch.testing.assert_close(got, grads_padded[k], rtol=1e-9, atol=1e-10, msg=f"gradient of {k}")
