"""Where does the packed softmax attention's time go?  (rt_attention_v2.hip: v2_fwd / v2_bwd_dq / v2_bwd_dkv on one C2 batch.)

Needs the diagnostic twin of the library (`python -m rectools_amd.build --ablation`, loaded through RT_LIB_PATH): its kernels read
RT_V2_ABLATE bits at every launch and leave parts of themselves out — 1 staging, 2 the tile loop, 4 softmax / mask / dropout arithmetic,
8 the three-way split of the probabilities, 16 the products (LDS fragment reads + MFMA), 32 the barriers of the chunk loop (v3),
64 the owner-row loads, 128 the stores,
256 / 512 the dQ / the dK,dV launch.  One C2-like batch (128 sessions, ML-20M-shaped lengths, longest first, 4 heads of 64, window 200,
p = 0.2); kernel time = HIP events around 200 back-to-back launches (nothing else on the device).

   RT_LIB_PATH=rectools_amd/librectools_hip_ablation.so python scripts/attn_ablate.py [out.md]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from rectools_amd import ops, synth

B, H, hd, L = 128, 4, 64, 200
d = H * hd
rng = np.random.default_rng(0)
lens = np.clip(synth.gen_lengths(B, 144.0, 20, 9254, rng) - 1, 1, L)
lens = np.sort(lens)[::-1].copy()
cu_h = np.zeros(B + 1, np.int64); cu_h[1:] = np.cumsum(lens)
n = int(cu_h[-1]); rows = (n + 127) // 128 * 128
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)   # noqa: E731
q, kv, do = r(rows, d), r(rows, 2 * d), r(rows, d)
bkv = r(2 * d) * 0.1
cu = torch.from_numpy(cu_h).cuda()
o, lse = torch.empty(rows, d, device="cuda"), torch.empty(rows, H, device="cuda")
dq, dkv, delta, part = torch.empty(rows, d, device="cuda"), torch.empty(rows, 2 * d, device="cuda"), torch.empty(rows, H, device="cuda"), torch.empty(B, d, device="cuda")
p, seed = 0.2, 1234


def fwd():
    ops._c("rt_mha_varlen_train_fwd", q, d, kv, 2 * d, kv[:, d:], 2 * d, cu, bkv, bkv[d:], B, H, hd, L, L, p, seed, o, d, lse)


def bwd():
    ops._c("rt_mha_varlen_bwd", q, d, kv, 2 * d, kv[:, d:], 2 * d, o, d, do, d, lse, cu, bkv, bkv[d:], B, H, hd, L, L, p, seed, dq, d, dkv, 2 * d,
           dkv[:, d:], 2 * d, delta, part)


def time_it(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


# useful work of one launch: the causal triangle of every (session, head); a product = 2 n (n + 1) / 2 hd flops
tri = float((lens.astype(np.float64) * (lens + 1) / 2).sum()) * H
flop = {"fwd": 2 * 2 * tri * hd, "dq": 3 * 2 * tri * hd, "dkv": 4 * 2 * tri * hd}
# executed tiles (16 owner rows x 32 partner rows, whole tiles up to the diagonal)
tiles = 0
for m in lens:
    for qt in range((int(m) + 15) // 16):
        tiles += ((qt * 16 + 15) >> 5) + 1
tiles *= H
byts = {"fwd": 4 * n * d * 4, "dq": 6 * n * d * 4, "dkv": 6 * n * d * 4}

MODES = [
    (0, "as built"),
    (2, "staging only (two images split into the LDS, no tile loop)"),
    (1, "no staging (tile loop on whatever the LDS holds)"),
    (1 | 64, "no staging, no owner-row loads (tile loop alone)"),
    (1 | 64 | 128, "... and no result stores"),
    (1 | 64 | 4, "tile loop without the softmax / mask / dropout arithmetic"),
    (1 | 64 | 4 | 8, "... and without the split of the probabilities (LDS reads + MFMA only)"),
    (1 | 64 | 16, "tile loop without the products (VALU only)"),
    (4, "everything but the softmax arithmetic"),
    (16, "everything but the products"),
    (4 | 8 | 16, "memory skeleton: staging + owner loads + stores, empty tile loop"),
    (1 | 64 | 4 | 8 | 16, "loop skeleton + stores: no staging, no owner loads, no arithmetic, no products"),
    (1 | 64 | 4 | 8 | 16 | 128, "loop skeleton alone (launch, session offsets, barriers, loop control)"),
    (1 | 64 | 4 | 8 | 16 | 128 | 32, "... without the barriers (v3 only)"),
    (1 | 2 | 64, "launch + session offsets only (every workgroup leaves after its first barrier)"),
]
def main():
    if "--once" in sys.argv:      # a counter pass's target (rocprofv3 --pmc): 20 launches of each kernel as built, nothing else
        for _ in range(20):
            fwd(); bwd()
        torch.cuda.synchronize()
        print(f"rows {n} tiles {tiles}")
        sys.exit(0)
    out = []
    out.append(f"kernels: RT_VARLEN_IMPL={os.environ.get('RT_VARLEN_IMPL', '(default: v3, streamed chunks)')}")
    out.append(f"one C2 batch: {B} sessions x {H} heads, {n} rows (mean {n / B:.0f}), {tiles} executed 16x32 tiles per product pair; "
               f"useful GFLOP fwd {flop['fwd'] / 1e9:.2f} dq {flop['dq'] / 1e9:.2f} dkv {flop['dkv'] / 1e9:.2f}; "
               f"algorithmic MB fwd {byts['fwd'] / 1e6:.0f} dq {byts['dq'] / 1e6:.0f} dkv {byts['dkv'] / 1e6:.0f}")
    out.append("")
    out.append("| RT_V2_ABLATE | what runs | v2_fwd µs | v2_bwd_dq µs | v2_bwd_dkv µs |")
    out.append("|---|---|---|---|---|")
    os.environ["RT_V2_ABLATE"] = "0"
    fwd(); bwd(); torch.cuda.synchronize()
    for bits, what in MODES:
        os.environ["RT_V2_ABLATE"] = str(bits)
        tf = time_it(fwd)
        os.environ["RT_V2_ABLATE"] = str(bits | 512)
        tq = time_it(bwd)
        os.environ["RT_V2_ABLATE"] = str(bits | 256)
        tk = time_it(bwd)
        out.append(f"| {bits} | {what} | {tf:.1f} | {tq:.1f} | {tk:.1f} |")
    os.environ["RT_V2_ABLATE"] = "0"
    peak6 = 2500e12 / 6
    t0 = [float(x) for x in out[5].split("|")[3:6]]
    out.append("")
    out.append("as built, useful flops / time against 2500 / 6 TF: fwd %.3f  dq %.3f  dkv %.3f; algorithmic bytes / time: %.2f / %.2f / %.2f TB/s" % (
        flop["fwd"] / t0[0] / 1e-6 / peak6, flop["dq"] / t0[1] / 1e-6 / peak6, flop["dkv"] / t0[2] / 1e-6 / peak6,
        byts["fwd"] / t0[0] / 1e6, byts["dq"] / t0[1] / 1e6, byts["dkv"] / t0[2] / 1e6))
    # matrix-pipe floor: executed MFMA cycles (16 per v_mfma_f32_16x16x32_bf16 and SIMD, MI355X_MICROARCH.md) spread over 1,024 SIMDs at 2.4 GHz
    mf = {"fwd": 48, "dq": 72, "dkv": 96}
    out.append("matrix-pipe floor (executed tiles x MFMAs per tile x 16 cycles / 1024 SIMDs / 2.4 GHz): " +
               "  ".join(f"{k} {tiles * v * 16 / 1024 / 2.4e3:.1f} µs" for k, v in mf.items()))
    txt = "\n".join(out)
    print(txt)
    if len(sys.argv) > 1 and not sys.argv[1].startswith("--"):
        open(sys.argv[1], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
