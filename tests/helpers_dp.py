"""Small seeded SASRec model + pre-collated batches for the multi-process tests (not the bench: `bench.py` times the
product loop of `SASRecModel.fit()`)."""
import numpy as np
import torch


def make_sasrec(V, d, H, n_blocks, L, dropout, loss, n_neg, device="cuda"):
    from rectools_amd import lightning as hl
    from rectools_amd import nn as hnn

    n_tokens = V + 1
    torch.manual_seed(31)  # before construction: biases / LayerNorm / embeddings take their default init from this stream
    item_model = hnn.SumOfEmbeddingsConstructor(n_tokens, [hnn.IdEmbeddingsItemNet(d, n_tokens, 0.0)])
    pos = hnn.LearnableInversePositionalEncoding(True, L, d)
    layers = hnn.SASRecTransformerLayers(n_blocks, d, H, dropout)
    bb = hnn.TransformerTorchBackbone(H, dropout, item_model, pos, layers, hnn.DistanceSimilarityModule("dot"), True, False)
    lm = hl.TransformerLossModule(bb, loss, n_neg, 0.2, 1.0, 1).to(device)
    torch.manual_seed(32)
    hl.xavier_normal_init(lm.torch_model)
    return lm


def make_train_batches(n_batches, B, L, V, n_neg, rank, seed=0):
    """SASRec training batches (x, y, yw, negatives) from ML-20M-shaped synthetic histories, collated exactly as
    SASRecDataPreparator._collate_fn_train does (sasrec.py:86-104): last L+1 items, left padding, shift by one."""
    from rectools_amd import synth

    n_users = n_batches * B
    u, it, _ = synth.gen_interactions(n_users, V, mean_len=144.0, min_len=20, max_len=9254, seed=seed + 17 * rank)
    it = it + 1  # internal ids: 0 is PAD
    bounds = np.concatenate([[0], np.cumsum(np.bincount(u, minlength=n_users))])
    x = np.zeros((n_users, L), np.int64)
    y = np.zeros((n_users, L), np.int64)
    for i in range(n_users):
        ses = it[bounds[i]:bounds[i + 1]][-(L + 1):]
        x[i, L - (len(ses) - 1):] = ses[:-1]
        y[i, L - (len(ses) - 1):] = ses[1:]
    rng = np.random.default_rng(seed + 5 + rank)
    out = []
    for b in range(n_batches):
        sl = slice(b * B, (b + 1) * B)
        yb = torch.from_numpy(y[sl])
        batch = {"x": torch.from_numpy(x[sl]).cuda(), "y": yb.cuda(), "yw": (yb != 0).float().cuda()}
        if n_neg:
            batch["negatives"] = torch.from_numpy(rng.integers(1, V + 1, size=(B, L, n_neg))).cuda()
        out.append(batch)
    return out
