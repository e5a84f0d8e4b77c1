"""The streamed forward attention kernel's launch as the device saw it: every workgroup's s_memtime stamps (ablation build, RT_V2_ABLATE
bit 1024; rt_attention_v3.hip) on the C2 batch of scripts/attn_ablate.py — when workgroups start and end, how long a chunk's staging and
its products take under co-residency, how many workgroups are resident over the launch.

   RT_LIB_PATH=rectools_amd/librectools_hip_ablation.so python scripts/attn_trace_v3.py [out.md]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_V2_ABLATE"] = "0"
import numpy as np
import torch

import attn_ablate as A            # noqa: E402   (the batch and the two launchers; its sweep runs under __main__ only)
from rectools_amd import _lib      # noqa: E402

for _ in range(10):
    A.fwd()
torch.cuda.synchronize()

lib = _lib.load()
lib.rt_v3_trace_read.restype = ctypes.c_int
lib.rt_v3_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros((8192, 16), np.uint64)
lib.rt_v3_trace_read(buf.ctypes.data, buf.nbytes)          # clear
os.environ["RT_V2_ABLATE"] = "1024"
A.fwd()
torch.cuda.synchronize()
os.environ["RT_V2_ABLATE"] = "0"
assert lib.rt_v3_trace_read(buf.ctypes.data, buf.nbytes) == 0
n_wg = A.B * A.H * ((A.L + 63) // 64)
t = buf[:n_wg].astype(np.int64)
n = (buf[:n_wg, 15] >> np.uint64(32)).astype(np.int64)
ob = (buf[:n_wg, 15] & np.uint64(0xFFFFFFFF)).astype(np.int64)
# s_memtime (slots 0 .. 11, 14) counts shader cycles per XCD: good for durations inside a workgroup; the launch's timeline comes from the
# device-wide 100 MHz clock (slots 13 / 12: entry / exit, 10 ns ticks)
life_cyc = np.where(t[:, 14] > 0, t[:, 14], t[:, 1]) - t[:, 0]
r0 = t[:, 13].min()
start, end = (t[:, 13] - r0) / 100.0, (t[:, 12] - r0) / 100.0      # microseconds
active = ob * 64 < n
out = []
out.append(f"{n_wg} workgroups launched, {int(active.sum())} with rows; the launch spans {end.max():.1f} us from the first workgroup's entry to the last "
           f"one's exit; durations inside a workgroup in shader cycles (s_memtime)")
span = float(end.max())
out.append("")
out.append("| owner block | workgroups | start us median / max | end us median / max | lifetime cycles median / max | per chunk: staged (load + split + barrier) median | computed median |")
out.append("|---|---|---|---|---|---|---|")
for o in range(int(ob.max()), -1, -1):
    m = active & (ob == o)
    if not m.any():
        continue
    life = life_cyc[m]
    st, cp = [], []
    for c in range(o + 1):
        prev = t[m, 1] if c == 0 else t[m, 3 + 2 * (c - 1)]
        st.append(t[m, 2 + 2 * c] - prev)
        cp.append(t[m, 3 + 2 * c] - t[m, 2 + 2 * c])
    st, cp = np.concatenate(st), np.concatenate(cp)
    out.append(f"| {o} | {int(m.sum())} | {np.median(start[m]):.1f} / {start[m].max():.1f} | {np.median(end[m]):.1f} / {end[m].max():.1f} | {int(np.median(life))} / {int(life.max())} | "
               f"{int(np.median(st))} | {int(np.median(cp))} |")
idle = ~active
out.append(f"| (no rows) | {int(idle.sum())} | {np.median(start[idle]):.1f} / {start[idle].max():.1f} | {np.median(end[idle]):.1f} / {end[idle].max():.1f} | {int(np.median(life_cyc[idle]))} | | |")
# residency over the launch: how many workgroups with rows are alive in each 1/20 of the span
edges = np.linspace(0, span, 21)
res = [(int(((start[active] < b) & (end[active] > a)).sum())) for a, b in zip(edges[:-1], edges[1:])]
out.append("")
out.append("workgroups with rows alive per 1/20 of the launch: " + " ".join(str(r) for r in res) + "   (768 = three per CU)")
started = [int((start < b).sum()) for b in edges[1:]]
out.append("workgroups (all 2,048) STARTED by the end of each 1/20:    " + " ".join(str(r) for r in started))
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 1 and not sys.argv[1].startswith("--"):
    open(sys.argv[1], "w").write(txt + "\n")
