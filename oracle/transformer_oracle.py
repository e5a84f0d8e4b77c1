"""Oracle (plain torch ops, fp32, CPU) for the transformer fit()/encode hot path.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, as explicit math on a flat `{reference parameter name: tensor}` dict:

  item table             item_net.py:101-132 (CatFeaturesItemNet: EmbeddingBag sum), 266-281, 463-482 (sum of blocks)
  embed + positions      net_blocks.py:374-400 ; torch_backbone.py:241-247
  masks                  torch_backbone.py:172-218,249-257  (causal / key-padding / merged with zero diagonal)
  multi-head attention   torch.nn.MultiheadAttention as called at sasrec.py:221-224, net_blocks.py:247-255,
                         ligr.py:90-98  (packed in_proj, 1/sqrt(hd) scaling, additive -inf mask, out_proj)
  SASRec block(s)        sasrec.py:197-230, 271-304
  Pre-LN block(s)        net_blocks.py:223-261, 305-335        (BERT4Rec)
  LiGR block(s)          ligr.py:66-106, 161-191               (eSASRec)
  STU block(s) + bias    hstu.py:84-153, 225-295, 364-399      (HSTU)
  logits                 similarity.py:84-115                   (dot / cosine, full catalog or candidates)
  losses                 lightning.py:144-212                   (softmax, BCE, gBCE, sampled_softmax)
  Adam                   torch.optim.Adam(lr, betas=(0.9, 0.98), eps=1e-8) — lightning.py:214-218

Dropout is the identity here (parity runs use dropout_rate = 0 or eval mode).
Gradients come from torch autograd over these plain ops.
"""
from __future__ import annotations

import math
import typing as tp

import torch
import torch.nn.functional as F

Params = tp.Dict[str, torch.Tensor]
Batch = tp.Dict[str, torch.Tensor]

ITEM_EMB = "item_model.item_net_blocks.0.ids_emb.weight"
POS_EMB = "pos_encoding_layer.pos_emb.weight"
CAT_BLOCK = "item_model.item_net_blocks.1."          # CatFeaturesItemNet as the second block (the models' default order)
CAT_EMB = CAT_BLOCK + "embedding_bag.weight"


# ----------------------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------------------
def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)  # biased variance, as nn.LayerNorm
    return (x - mu) / torch.sqrt(var + eps) * w + b


def item_table(p: Params) -> torch.Tensor:
    """`SumOfEmbeddingsConstructor.get_all_embeddings()` (item_net.py:361-368,463-482): id embeddings, plus — when the
    parameter dict holds a CatFeaturesItemNet block — the sum of the embeddings of every (feature, value) id the item
    carries: `EmbeddingBag(mode="sum")` over `emb_bag_inputs[offsets[i] : offsets[i] + input_lengths[i]]`
    (item_net.py:101-132; values / weights of the feature matrix are not used, only its pattern)."""
    emb = p[ITEM_EMB]
    if CAT_EMB not in p:
        return emb
    inputs, offsets, lens = p[CAT_BLOCK + "emb_bag_inputs"], p[CAT_BLOCK + "offsets"], p[CAT_BLOCK + "input_lengths"]
    rows = []
    for i in range(emb.shape[0]):      # plain loop: this is the checker, catalogues here are small
        vals = inputs[int(offsets[i]): int(offsets[i]) + int(lens[i])]
        rows.append(p[CAT_EMB][vals].sum(dim=0))
    return emb + torch.stack(rows)


def embed_sessions(p: Params, x: torch.Tensor, use_scale: bool, table: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
    """item_embs[sessions] (* sqrt(d)) + pos_emb[L-1-j]  (torch_backbone.py:245-246, net_blocks.py:388-399)."""
    emb = item_table(p) if table is None else table
    seqs = emb[x]
    L, d = x.shape[1], emb.shape[1]
    if use_scale:
        seqs = seqs * (d ** 0.5)
    if POS_EMB in p:
        positions = torch.arange(L - 1, -1, -1)
        seqs = seqs + p[POS_EMB][positions][None, :, :]
    return seqs


def attention_mask(x: torch.Tensor, causal: bool, keypad: bool) -> tp.Optional[torch.Tensor]:
    """Additive float mask [B, L, L] (0 / -inf), or None.

    causal: `~tril` (torch_backbone.py:249-252); key padding: `sessions == 0` (:254); both: merged with the
    diagonal forced to 0 (:172-218) so fully padded query rows attend to themselves.
    """
    B, L = x.shape
    if not causal and not keypad:
        return None
    m = torch.zeros(B, L, L)
    if causal:
        m = m.masked_fill(~torch.tril(torch.ones(L, L, dtype=torch.bool))[None], float("-inf"))
    if keypad:
        m = m.masked_fill((x == 0)[:, None, :], float("-inf"))
        if causal:
            idx = torch.arange(L)
            m[:, idx, idx] = 0.0
    return m


def mha(q_in: torch.Tensor, kv_in: torch.Tensor, p: Params, prefix: str, n_heads: int,
        mask: tp.Optional[torch.Tensor]) -> torch.Tensor:
    """torch.nn.MultiheadAttention(batch_first=True, need_weights=False) restated."""
    B, L, d = q_in.shape
    hd = d // n_heads
    w, bias = p[prefix + "in_proj_weight"], p[prefix + "in_proj_bias"]
    q = q_in @ w[:d].T + bias[:d]
    k = kv_in @ w[d:2 * d].T + bias[d:2 * d]
    v = kv_in @ w[2 * d:].T + bias[2 * d:]
    q = q.view(B, L, n_heads, hd).transpose(1, 2)  # [B,H,L,hd]
    k = k.view(B, L, n_heads, hd).transpose(1, 2)
    v = v.view(B, L, n_heads, hd).transpose(1, 2)
    scores = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if mask is not None:
        scores = scores + mask[:, None, :, :]
    attn = torch.softmax(scores, dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, L, d)
    return out @ p[prefix + "out_proj.weight"].T + p[prefix + "out_proj.bias"]


def ffn(x: torch.Tensor, p: Params, prefix: str, activation: str) -> torch.Tensor:
    """PointWiseFeedForward / SwigluFeedForward (net_blocks.py:21-111)."""
    def lin(name: str, t: torch.Tensor) -> torch.Tensor:
        y = t @ p[prefix + name + ".weight"].T
        if (prefix + name + ".bias") in p:
            y = y + p[prefix + name + ".bias"]
        return y

    if activation == "swiglu":
        h = F.silu(lin("ff_linear_1", x)) * lin("ff_linear_3", x)
    elif activation == "relu":
        h = torch.relu(lin("ff_linear_1", x))
    elif activation == "gelu":
        h = F.gelu(lin("ff_linear_1", x))  # exact erf GELU (torch.nn.GELU default)
    else:
        raise ValueError(activation)
    return lin("ff_linear_2", h)


# ----------------------------------------------------------------------------------------------------
# layer stacks
# ----------------------------------------------------------------------------------------------------
def sasrec_layers(seqs, tl_mask, mask, p: Params, n_blocks: int, n_heads: int) -> torch.Tensor:
    for i in range(n_blocks):
        pre = f"transformer_layers.transformer_blocks.{i}."
        seqs = seqs * tl_mask
        q = layer_norm(seqs, p[pre + "q_layer_norm.weight"], p[pre + "q_layer_norm.bias"], 1e-5)
        seqs = q + mha(q, seqs, p, pre + "multi_head_attn.", n_heads, mask)  # Q from LN(x), K/V from raw x
        ff_in = layer_norm(seqs, p[pre + "ff_layer_norm.weight"], p[pre + "ff_layer_norm.bias"], 1e-5)
        seqs = ffn(ff_in, p, pre + "feed_forward.", "relu") + ff_in
    seqs = seqs * tl_mask
    return layer_norm(seqs, p["transformer_layers.last_layernorm.weight"],
                      p["transformer_layers.last_layernorm.bias"], 1e-8)


def preln_layers(seqs, tl_mask, mask, p: Params, n_blocks: int, n_heads: int) -> torch.Tensor:
    for i in range(n_blocks):
        pre = f"transformer_layers.transformer_blocks.{i}."
        h = layer_norm(seqs, p[pre + "layer_norm_1.weight"], p[pre + "layer_norm_1.bias"], 1e-5)
        seqs = seqs + mha(h, h, p, pre + "multi_head_attn.", n_heads, mask)
        f_in = layer_norm(seqs, p[pre + "layer_norm_2.weight"], p[pre + "layer_norm_2.bias"], 1e-5)
        seqs = seqs + ffn(f_in, p, pre + "feed_forward.", "gelu")
    return seqs


def ligr_layers(seqs, tl_mask, mask, p: Params, n_blocks: int, n_heads: int, ff_activation: str) -> torch.Tensor:
    for i in range(n_blocks):
        pre = f"transformer_layers.transformer_blocks.{i}."
        h = layer_norm(seqs, p[pre + "layer_norm_1.weight"], p[pre + "layer_norm_1.bias"], 1e-5)
        a = mha(h, h, p, pre + "multi_head_attn.", n_heads, mask)
        g1 = torch.sigmoid(seqs @ p[pre + "gating_linear_1.weight"].T + p[pre + "gating_linear_1.bias"])
        seqs = seqs + g1 * a
        f_in = layer_norm(seqs, p[pre + "layer_norm_2.weight"], p[pre + "layer_norm_2.bias"], 1e-5)
        f_out = ffn(f_in, p, pre + "feed_forward.", ff_activation)
        g2 = torch.sigmoid(seqs @ p[pre + "gating_linear_2.weight"].T + p[pre + "gating_linear_2.bias"])
        seqs = seqs + g2 * f_out
    return seqs


def time_buckets(unix_ts: torch.Tensor, num_buckets: int = 128) -> torch.Tensor:
    """[B, L, L] int64 bucket of (t_{i+1} - t_j)  (hstu.py:84-113).

    bucket = clamp(trunc(log(max(1, |dt|)) / 0.301), 0, num_buckets), in float32 like the reference.
    """
    ext = torch.cat([unix_ts, unix_ts[:, -1:]], dim=1)
    diff = ext[:, 1:].unsqueeze(2) - ext[:, :-1].unsqueeze(1)  # [B, L+1, L+1]
    b = (torch.log(torch.abs(diff).clamp(min=1)) / 0.301).long()
    return torch.clamp(b, 0, num_buckets)[:, :-1, :-1]


def rel_attn_bias(p: Params, prefix: str, batch: Batch, L: int) -> torch.Tensor:
    B = batch["x"].shape[0]
    rab = torch.zeros(B, L, L)
    if (prefix + "time_weights") in p:     # num_buckets = the length of the weight vector - 1 (hstu.py:75-78)
        rab = rab + p[prefix + "time_weights"][time_buckets(batch["unix_ts"], p[prefix + "time_weights"].numel() - 1)]
    if (prefix + "pos_weights") in p:
        i = torch.arange(L)[:, None]
        j = torch.arange(L)[None, :]
        rab = rab + p[prefix + "pos_weights"][(L - 1) + j - i][None]  # Toeplitz (hstu.py:115-128)
    return rab


def stu_layers(seqs, tl_mask, batch: Batch, p: Params, n_blocks: int, n_heads: int, lin: tp.Optional[int] = None,
               att: tp.Optional[int] = None) -> torch.Tensor:
    """hstu.py:225-295,364-399.  lin / att: linear_hidden_dim (u, v) and attention_dim (q, k) per head; HSTUModel sets both to
    n_factors // n_heads (hstu.py:662-669)."""
    B, L, d = seqs.shape
    lin = d // n_heads if lin is None else lin
    att = d // n_heads if att is None else att
    causal = torch.tril(torch.ones(L, L))  # (~attn_mask).int() of the ~tril mask (hstu.py:394)
    for i in range(n_blocks):
        pre = f"transformer_layers.stu_blocks.{i}."
        seqs = seqs * tl_mask
        normed = layer_norm(seqs, p[pre + "norm_input.weight"], p[pre + "norm_input.bias"], 1e-6) * tl_mask
        uvqk = F.silu(normed @ p[pre + "uvqk_proj"])
        u, v, q, k = torch.split(uvqk, [lin * n_heads, lin * n_heads, att * n_heads, att * n_heads], dim=-1)   # hstu.py:259-268
        qh = q.view(B, L, n_heads, att)
        kh = k.view(B, L, n_heads, att)
        vh = v.reshape(B, L, n_heads, lin)
        qk = torch.einsum("bnhd,bmhd->bhnm", qh, kh) + rel_attn_bias(p, pre + "rel_attn.", batch, L)[:, None]
        qk = F.silu(qk) / L
        pad2 = tl_mask.squeeze(-1)[:, None, :] * tl_mask  # [B, L, L]: m_i * m_j
        qk = qk * causal[None, None] * pad2[:, None]
        attn = torch.einsum("bhnm,bmhd->bnhd", qk, vh).reshape(B, L, n_heads * lin)
        o_in = u * layer_norm(attn, p[pre + "norm_attn_output.weight"], p[pre + "norm_attn_output.bias"], 1e-6) * tl_mask
        seqs = o_in @ p[pre + "output_mlp.weight"].T + p[pre + "output_mlp.bias"] + seqs
    return seqs * tl_mask


# ----------------------------------------------------------------------------------------------------
# backbone / logits / losses
# ----------------------------------------------------------------------------------------------------
def encode_sessions(cfg: dict, p: Params, batch: Batch, table: tp.Optional[torch.Tensor] = None) -> torch.Tensor:
    """TransformerTorchBackbone.encode_sessions (torch_backbone.py:220-260)."""
    x = batch["x"]
    tl_mask = (x != 0).unsqueeze(-1).float()
    seqs = embed_sessions(p, x, cfg.get("use_scale", False), table)
    kind = cfg["layers"]
    if kind == "stu":
        return stu_layers(seqs, tl_mask, batch, p, cfg["n_blocks"], cfg["H"], cfg.get("linear_hidden_dim"), cfg.get("attention_dim"))
    mask = attention_mask(x, cfg["causal"], cfg["keypad"])
    if kind == "sasrec":
        return sasrec_layers(seqs, tl_mask, mask, p, cfg["n_blocks"], cfg["H"])
    if kind == "preln":
        return preln_layers(seqs, tl_mask, mask, p, cfg["n_blocks"], cfg["H"])
    if kind == "ligr":
        return ligr_layers(seqs, tl_mask, mask, p, cfg["n_blocks"], cfg["H"],
                           cfg["layer_kwargs"].get("ff_activation", "swiglu"))
    raise ValueError(kind)


def _l2norm(e: torch.Tensor) -> torch.Tensor:
    # similarity.py:97-100.  torch.norm (not sqrt(sum(e*e))): its backward is 0 for an all-zero row, and HSTU
    # feeds exactly-zero session rows (padded slots are multiplied by the timeline mask) through here.
    n = torch.norm(e, p=2, dim=-1, keepdim=True)
    return e / torch.max(n, torch.tensor([1e-8]))


def batch_logits(cfg: dict, p: Params, batch: Batch) -> torch.Tensor:
    """`get_batch_logits` (lightning.py:301-309): logits / logits_t."""
    items = item_table(p)        # once per forward, shared by the encoder and the logits (torch_backbone.py:290-293)
    sess = encode_sessions(cfg, p, batch, items)
    if cfg["dist"] == "cosine":
        sess, items = _l2norm(sess), _l2norm(items)
    if cfg["loss"] == "softmax":
        logits = sess @ items.T
    else:
        cand = torch.cat([batch["y"].unsqueeze(-1), batch["negatives"]], dim=-1)
        logits = (items[cand] @ sess.unsqueeze(-1)).squeeze(-1)
    return logits / cfg.get("logits_t", 1.0)


def softmax_loss(logits: torch.Tensor, y: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """lightning.py:145-162.  NB: the divisor counts positions whose weighted loss is > 0."""
    lse = torch.logsumexp(logits, dim=-1)
    picked = torch.gather(logits, -1, y.unsqueeze(-1)).squeeze(-1)
    ce = torch.where(y != 0, lse - picked, torch.zeros_like(lse))  # ignore_index = 0
    loss = ce * w
    return loss.sum() / (loss > 0).to(loss.dtype).sum()


def bce_loss(logits: torch.Tensor, y: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """lightning.py:189-199."""
    mask = (y != 0)
    target = torch.zeros_like(logits)
    target[:, :, 0] = 1
    loss = F.softplus(logits) - logits * target  # == bce_with_logits(logits, target)
    loss = loss.mean(-1) * mask * w
    return loss.sum() / mask.sum()


def gbce_logits(logits: torch.Tensor, n_items: int, n_negatives: int, gbce_t: float) -> torch.Tensor:
    """lightning.py:164-186 (float64 transform of the positive logit)."""
    alpha = n_negatives / (n_items - 1)
    beta = alpha * (gbce_t * (1 - 1 / alpha) + 1 / alpha)
    pos = logits[:, :, 0:1].to(torch.float64)
    neg = logits[:, :, 1:].to(torch.float64)
    eps = 1e-10
    fmax = torch.finfo(torch.float64).max
    probs = torch.clamp(torch.sigmoid(pos), eps, 1 - eps)
    adj = torch.clamp(probs.pow(-beta), 1 + eps, fmax)
    adj = torch.clamp(1.0 / (adj - 1), eps, fmax)
    return torch.cat([torch.log(adj), neg], dim=-1)


def sampled_softmax_loss(logits: torch.Tensor, y: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """lightning.py:207-212: positive moved to class 1, class 0 is ignore_index."""
    swapped = torch.cat([logits[:, :, 1:2], logits[:, :, 0:1], logits[:, :, 2:]], dim=-1)
    return softmax_loss(swapped, (y != 0).long(), w)


def training_loss(cfg: dict, p: Params, batch: Batch) -> torch.Tensor:
    logits = batch_logits(cfg, p, batch)
    y, w = batch["y"], batch["yw"]
    loss = cfg["loss"]
    if loss == "softmax":
        return softmax_loss(logits, y, w)
    if loss == "BCE":
        return bce_loss(logits, y, w)
    if loss == "gBCE":
        n_items = p[ITEM_EMB].shape[0] - cfg["n_extra"]  # lightning.py:202
        return bce_loss(gbce_logits(logits, n_items, cfg["N"], cfg["gbce_t"]), y, w)
    if loss == "sampled_softmax":
        return sampled_softmax_loss(logits, y, w)
    raise ValueError(f"loss {loss} is not supported")  # lightning.py:328


def loss_and_grads(cfg: dict, params: Params, batch: Batch) -> tp.Tuple[torch.Tensor, Params]:
    p = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in params.items()}
    loss = training_loss(cfg, p, batch)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items() if v.is_floating_point()}
    # nn.Embedding(padding_idx=0): the PAD row receives no gradient from the *gather* (item_net.py:260-264)
    # — but it does from the full-catalog / candidate logits, exactly as autograd computes here? No:
    # padding_idx zeroes only the gradient flowing through `ids_emb(items)`; the reference materialises the
    # whole table THROUGH ids_emb (get_all_embeddings), so row 0 never receives any gradient at all.
    grads[ITEM_EMB] = grads[ITEM_EMB].clone()
    grads[ITEM_EMB][0] = 0
    return loss.detach(), grads


class AdamState:
    """torch.optim.Adam(lr, betas, eps=1e-8, weight_decay=0) restated (dense, fp32)."""

    def __init__(self, lr: float, betas=(0.9, 0.98), eps: float = 1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, betas[0], betas[1], eps
        self.t = 0
        self.m: Params = {}
        self.v: Params = {}

    def step(self, params: Params, grads: Params) -> Params:
        self.t += 1
        out = {}
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for k, p in params.items():
            if k not in grads:          # integer buffers (item-feature structure) are not optimised
                out[k] = p
                continue
            g = grads[k]
            m = self.m.get(k, torch.zeros_like(p)) * self.b1 + (1 - self.b1) * g
            v = self.v.get(k, torch.zeros_like(p)) * self.b2 + (1 - self.b2) * g * g
            self.m[k], self.v[k] = m, v
            denom = v.sqrt() / math.sqrt(bc2) + self.eps
            out[k] = p - (self.lr / bc1) * (m / denom)
        return out
