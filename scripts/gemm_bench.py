"""rt_gemm microbenchmark on the products of the C2 training step (+ one large square).  Usage on the GPU box:
   RT_GEMM_IMPL={0,2,3,4} python scripts/gemm_bench.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rectools_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
def rnd(*s): return torch.randn(*s, generator=g).to(dev)

def time_it(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us

out = {}
SHAPES = [(25600, 256, 256), (25600, 512, 256), (25600, 768, 256), (8192, 8192, 8192), (25600, 1024, 256)]
if len(sys.argv) > 1:      # "MxNxK,MxNxK,..."
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in sys.argv[1].split(",")]
for (M, N, K) in SHAPES:
    x, w, b = rnd(M, K), rnd(N, K), rnd(N)
    dy = rnd(M, N)
    y = torch.empty(M, N, device=dev); dx = torch.empty(M, K, device=dev); dw = torch.empty(N, K, device=dev)
    fl = 2.0 * M * N * K
    def fwd(): ops._gemm(x, K, 1, w, K, 1, y, N, b, None, 0, M, N, K, 0)
    def dgrad(): ops._gemm(dy, N, 1, w, K, 0, dx, K, None, None, 0, M, K, N)
    sp = ops._wgrad_splits(M)
    def wgrad(): ops._gemm(dy, N, 0, x, K, 0, dw, K, None, None, 0, N, K, M, 0, sp)
    t = {}
    for name, fn in (("fwd", fwd), ("dgrad", dgrad), ("wgrad", wgrad)):
        us = time_it(fn, 5 if M == 8192 else 20)
        t[name] = (round(us, 1), round(fl / us / 1e6, 1))
    torch.cuda.synchronize()
    # correctness vs fp64 on a row/col sample
    idx = torch.arange(0, M, max(1, M // 97), device=dev)
    ref = (x[idx].double() @ w.double().T + b.double())
    e_f = float((y[idx].double() - ref).abs().max() / ref.abs().max())
    ref = dy[idx].double() @ w.double()
    e_d = float((dx[idx].double() - ref).abs().max() / ref.abs().max())
    jdx = torch.arange(0, N, max(1, N // 31), device=dev)
    ref = dy[:, jdx].double().T @ x.double()
    e_w = float((dw[jdx].double() - ref).abs().max() / ref.abs().max())
    out[f"{M}x{N}x{K}"] = {"us,TF": t, "relerr": [f"{e_f:.1e}", f"{e_d:.1e}", f"{e_w:.1e}"], "splits": sp}
print("RT_GEMM_IMPL=" + os.environ.get("RT_GEMM_IMPL", "default"))
for k, v in out.items(): print(" ", k, json.dumps(v))
