"""How many workgroups with a given LDS footprint are resident on one CU at the same time?  (library built with -DRT_ATTN_TRACE)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rectools_amd import _lib

torch.zeros(1, device="cuda")
lib = _lib.load()
lib.rt_debug_occupancy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
for threads, lds in ((256, 1024), (256, 32 * 1024), (256, 50112), (256, 64 * 1024), (256, 80 * 1024), (512, 122 * 1024), (256, 40 * 1024)):
    n = 2048
    buf = (ctypes.c_ulonglong * (4 * n))()
    rc = lib.rt_debug_occupancy(n, threads, lds, 200000, buf)
    a = np.array(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
    t0, t1, hw, xcc = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    cu = ((xcc & 0xF) << 16) | (hw & 0xFF00)          # xcc | se/sh/cu bits of HW_ID (bits 8..15)
    best = 0
    per_cu = []
    for c in np.unique(cu):
        m = cu == c
        ev = sorted([(x, 1) for x in t0[m]] + [(x, -1) for x in t1[m]])
        cur = mx = 0
        for _, d in ev:
            cur += d
            mx = max(mx, cur)
        per_cu.append(mx)
    span = (t1.max() - t0.min())
    print(f"threads {threads} lds {lds:6d} B: rc {rc}  distinct CUs {len(per_cu)}  max resident per CU: min {min(per_cu)} median {int(np.median(per_cu))} "
          f"max {max(per_cu)}   kernel span {span} ticks for {n} WGs x 200000 ticks")
