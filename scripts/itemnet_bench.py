"""K1b micro-benchmark: item-row producer (ids_emb + category bag sums) forward / backward, HIP-event timed.
usage: python scripts/itemnet_bench.py [V d F per_item]..."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rectools_amd import ops  # noqa: E402


def run(V, d, F, per_item, reps=10):
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 2 * per_item + 1, V)
    lens[0] = 0
    # Zipf-like popularity of the category values (a few values tag a large share of the catalog)
    probs = 1.0 / np.arange(1, F + 1)
    probs /= probs.sum()
    inputs = rng.choice(F, int(lens.sum()), p=probs)
    dev = torch.device("cuda:0")
    offsets = torch.from_numpy(np.cumsum(lens) - lens).to(dev)
    bag = ops.BagStructure(torch.from_numpy(inputs).to(dev), offsets, torch.from_numpy(lens).to(dev), F)
    ids_w = torch.randn(V, d, device=dev)
    cat_w = torch.randn(F, d, device=dev).requires_grad_(True)
    ids_w.requires_grad_(True)
    g = torch.randn(V, d, device=dev)
    for p in (0.0, 0.2):
        f_ms, b_ms = [], []
        for it in range(reps + 2):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            out = ops.item_table(ids_w, cat_w, bag, p)
            e[1].record()
            out.backward(g)
            e[2].record()
            torch.cuda.synchronize()
            ids_w.grad = cat_w.grad = None
            if it >= 2:
                f_ms.append(e[0].elapsed_time(e[1])); b_ms.append(e[1].elapsed_time(e[2]))
        fwd_bytes = 2 * V * d * 4 + len(inputs) * 8
        bwd_bytes = len(inputs) * (d * 4 + 8) + bag.n_chunks * d * 8
        print(f"V={V} d={d} F={F} nnz={len(inputs)} chunks={bag.n_chunks} p={p}: fwd {np.mean(f_ms)*1e3:.1f} us "
              f"({fwd_bytes/np.mean(f_ms)/1e6:.0f} GB/s algorithmic)  bwd {np.mean(b_ms)*1e3:.1f} us "
              f"({bwd_bytes/np.mean(b_ms)/1e6:.0f} GB/s over gathered gradient rows)", flush=True)


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    cases = [a[i:i + 4] for i in range(0, len(a), 4)] or [[26745, 256, 1000, 4], [1_000_001, 256, 5000, 4], [5_000_001, 512, 20000, 3]]
    for c in cases:
        run(*c)
