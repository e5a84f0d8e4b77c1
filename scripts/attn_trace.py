"""Timeline of one wave of the resident forward attention kernel (library built with RT_EXTRA_HIPCC_FLAGS=-DRT_ATTN_TRACE)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rectools_amd import _lib, ops

dev = torch.device("cuda:0")
B, H, L, d = 128, 4, 200, 256
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(B * L, d, generator=g).to(dev) for _ in range(3))
ids = torch.randint(1, 1000, (B, L), generator=g).to(dev)
lib = _lib.load()
for p in (0.2, 0.0):
    for _ in range(3):
        ops.mha(q, k, v, ids, B, H, L, True, False, p)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 1024)()
    ctypes.memset(buf, 0, 8192)
    lib.rt_debug_attn_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rc = lib.rt_debug_attn_trace(buf, 1024)
    t = np.array(buf, dtype=np.uint64).astype(np.int64)
    print(f"p={p}: bh=300 starts {t[256] - t[0]} ticks after bh=1 starts; bh=1 exits at {t[5] - t[0]}, bh=300 exits at {t[261] - t[0]}")
    for base, name in ((0, "bh=1 (first round)"), (256, "bh=300 (second round)")):
        m = t[base:base + 6] - t[base]
        print(f"p={p} {name}: s_memtime ticks (100 MHz => x24 core cycles at 2.4 GHz): start 0 | K,V staged {m[1]} | barrier {m[2]} | "
              f"pairs done {m[3]} | stored {m[4]} | exit {m[5]}")
        rows = []
        for kt in range(7):
            x = t[base + 16 + kt * 4: base + 16 + kt * 4 + 4]
            nxt = t[base + 16 + (kt + 1) * 4]
            rows.append((int(x[0] - t[base]), int(x[1] - x[0]), int(x[3] - x[1]), int(x[2] - x[3]), int(nxt - x[2])))
        print("   per step (start, stage1, stage2, tail, to-next-top):", rows)
    for base, name in ((512, "ring bh=1"), (768, "ring bh=300")):
        if t[base] == 0:
            continue
        m = t[base:base + 5] - t[base]
        print(f"p={p} {name}: start 0 | loop entry {m[1]} | loop done {m[3]} | stored {m[4]}")
        rows = []
        for kt in range(7):
            x = t[base + 16 + kt * 8: base + 16 + kt * 8 + 8]
            rows.append(dict(top=int(x[4] - t[base]), vmwait=int(x[5] - x[4]), barrier=int(x[6] - x[5]), issue=int(x[0] - x[6]), S=int(x[1] - x[0]),
                             softmax=int(x[2] - x[1]), PV=int(x[3] - x[2])))
        for r in rows:
            print("     ", r)
