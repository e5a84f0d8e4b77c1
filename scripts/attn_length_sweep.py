"""How the streamed attention's matrix-pipe share moves with the session length (docs/kernels/attention_streamed.md §5: the C2 batch's
0.10 / 0.11 / 0.14 is a property of its 113-row sessions, not of the kernels).  Uniform sessions of L rows, B = 16,384 / L of them, 4 heads
of 64, p = 0.2; kernel time = HIP events around 100 back-to-back launches; executed 16 x 32 tile steps counted as the kernels skip them
(whole tiles up to the diagonal); matrix-pipe floor = steps x MFMAs per step x 16 cycles / 1,024 SIMDs / 2.4 GHz.
   python scripts/attn_length_sweep.py [out.md]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rectools_amd import ops, synth

H, hd = 4, 64
d = H * hd
p, seed = 0.2, 1234
MF = {"fwd": 48, "bwd": 72 + 96}


def run(lens, window):
    B = len(lens)
    cu_h = np.zeros(B + 1, np.int64); cu_h[1:] = np.cumsum(lens)
    n = int(cu_h[-1]); rows = (n + 127) // 128 * 128
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)   # noqa: E731
    q, kv, do = r(rows, d), r(rows, 2 * d), r(rows, d)
    bkv = r(2 * d) * 0.1
    cu = torch.from_numpy(cu_h).cuda()
    o, lse = torch.empty(rows, d, device="cuda"), torch.empty(rows, H, device="cuda")
    dq, dkv, delta, part = torch.empty(rows, d, device="cuda"), torch.empty(rows, 2 * d, device="cuda"), torch.empty(rows, H, device="cuda"), torch.empty(B, d, device="cuda")

    def fwd():
        ops._c("rt_mha_varlen_train_fwd", q, d, kv, 2 * d, kv[:, d:], 2 * d, cu, bkv, bkv[d:], B, H, hd, window, window, p, seed, o, d, lse)

    def bwd():
        ops._c("rt_mha_varlen_bwd", q, d, kv, 2 * d, kv[:, d:], 2 * d, o, d, do, d, lse, cu, bkv, bkv[d:], B, H, hd, window, window, p, seed, dq, d, dkv, 2 * d,
               dkv[:, d:], 2 * d, delta, part)

    def time_it(fn, reps=100):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    tiles = 0
    for m in lens:
        for qt in range((int(m) + 15) // 16):
            tiles += ((qt * 16 + 15) >> 5) + 1
    tiles *= H
    waves = sum((int(m) + 15) // 16 for m in lens) * H
    tf, tb = time_it(fwd), time_it(bwd)
    floor = {k: tiles * v * 16 / 1024 / 2.4e3 for k, v in MF.items()}
    return n, tiles / waves, tf, tb, floor["fwd"] / tf, floor["bwd"] / tb


def main():
    rng = np.random.default_rng(0)
    c2 = np.clip(synth.gen_lengths(128, 144.0, 20, 9254, rng) - 1, 1, 200)
    rows = [("C2 batch (ML-20M-shaped, window 200)", c2, 200)]
    for L in (200, 512, 1024, 2048, 4096):
        rows.append((f"uniform {L}", np.full(max(1, 16384 // L), L), L))
    out = ["| sessions | rows | tile steps per wave | fwd µs | bwd (dQ + dK/dV) µs | matrix-pipe floor / time: fwd | bwd |", "|---|---|---|---|---|---|---|"]
    for name, lens, window in rows:
        n, spw, tf, tb, bf, bb = run(lens, window)
        out.append(f"| {name} | {n} | {spw:.1f} | {tf:.1f} | {tb:.1f} | {bf:.3f} | {bb:.3f} |")
    out.append("")
    out.append("(floor / time assumes 2.4 GHz; the SQ counters of the C2 batch put the same launches at 0.098 fwd — the clock under load is lower)")
    txt = "\n".join(out)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
