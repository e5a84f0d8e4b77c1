"""One tiny SASRec training step (forward + loss + backward + fused Adam) on cuda:0, checked against the CPU oracle.
Called by `__graft_entry__.smoke()`; the oracle import lives here only as the checker."""
from __future__ import annotations

import torch


def run() -> None:
    from oracle import transformer_oracle as T  # checker only
    from rectools_amd import lightning as hl
    from rectools_amd import nn as hnn

    cfg = dict(V=300, B=4, L=50, d=64, H=2, n_blocks=2, N=8, loss="sampled_softmax", dist="dot", logits_t=1.0, causal=True,
               keypad=False, layers="sasrec", n_extra=1, gbce_t=0.2, lr=1e-3)
    n_tokens = cfg["V"] + 1
    item_model = hnn.SumOfEmbeddingsConstructor(n_tokens, [hnn.IdEmbeddingsItemNet(cfg["d"], n_tokens, 0.0)])
    pos = hnn.LearnableInversePositionalEncoding(True, cfg["L"], cfg["d"])
    layers = hnn.SASRecTransformerLayers(cfg["n_blocks"], cfg["d"], cfg["H"], 0.0)
    bb = hnn.TransformerTorchBackbone(cfg["H"], 0.0, item_model, pos, layers, hnn.DistanceSimilarityModule("dot"), True, False)
    lm = hl.TransformerLossModule(bb, "sampled_softmax", cfg["N"], 0.2, 1.0, 1).to("cuda:0")
    torch.manual_seed(0)
    hl.xavier_normal_init(lm.torch_model)
    g = torch.Generator().manual_seed(1)
    x = torch.randint(1, n_tokens, (cfg["B"], cfg["L"]), generator=g)
    x[1, :30] = 0
    y = torch.roll(x, -1, dims=1)
    y[:, -1] = torch.randint(1, n_tokens, (cfg["B"],), generator=g)
    y[x == 0] = 0
    batch = {"x": x, "y": y, "yw": (y != 0).float(), "negatives": torch.randint(1, n_tokens, (cfg["B"], cfg["L"], cfg["N"]), generator=g)}
    params = {k: v.detach().cpu().clone() for k, v in lm.torch_model.state_dict().items()}
    loss_ref, g_ref = T.loss_and_grads(cfg, params, batch)
    lm.train()
    opt = hl.FlatAdam(lm.torch_model, lr=cfg["lr"])
    opt.zero_grad()
    loss = lm.training_loss({k: v.to("cuda:0") for k, v in batch.items()})
    loss.backward()
    assert abs(float(loss.detach()) - float(loss_ref)) < 1e-4 * abs(float(loss_ref)), (float(loss.detach()), float(loss_ref))
    for n, p in lm.torch_model.named_parameters():
        ref = g_ref[n]
        tol = 1e-2 * float(ref.abs().max()) + 1e-7
        assert float((p.grad.detach().cpu() - ref).abs().max()) <= tol, n
    opt.step()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in lm.torch_model.parameters())
    print(f"[smoke] SASRec train step (HIP) == oracle: loss {float(loss.detach()):.6f}, {len(g_ref)} gradients ok")
