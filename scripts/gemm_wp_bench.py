"""rt_gemm (both operands split in registers) vs rt_gemm_wp (pre-split weight planes) on the shapes of the C2 step / the recommend encoder.
   python scripts/gemm_wp_bench.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rectools_amd import _lib, ops

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)   # noqa: E731


def time_it(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


SHAPES = [(18432, 256, 256), (147456, 256, 256), (147456, 512, 256), (8192, 8192, 8192)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in t.split('x')) for t in sys.argv[1].split(',')]
for (M, N, K) in SHAPES:
    x, w, dy = rnd(M, K), rnd(N, K), rnd(M, N)
    y, dx = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev)
    n = w.numel(); stride = (n + 7) // 8 * 8
    planes = torch.empty(3 * stride, dtype=torch.int16, device=dev)
    ops._c("rt_split_planes", w, n, planes, stride)
    arr = (_lib.GemmWpProblem * 1)()

    def wp(A, C, Mq, Nq, Kq, ldw, tr):
        q = arr[0]
        q.A, q.lda, q.W, q.plane_stride, q.ldw, q.C, q.ldc, q.bias, q.R, q.ldr = A.data_ptr(), A.stride(0), planes.data_ptr(), stride, ldw, C.data_ptr(), C.stride(0), None, None, 0
        q.M, q.N, q.K, q.relu = Mq, Nq, Kq, 0
        ops._c("rt_gemm_wp", ctypes.cast(arr, ctypes.c_void_p), 1, tr)

    fl = 2.0 * M * N * K
    reps = 5 if M == 8192 else 20
    t = {
        "fwd rt_gemm": time_it(lambda: ops._gemm(x, K, 1, w, K, 1, y, N, None, None, 0, M, N, K, 0), reps),
        "fwd rt_gemm_wp": time_it(lambda: wp(x, y, M, N, K, K, 0), reps),
        "dgrad rt_gemm": time_it(lambda: ops._gemm(dy, N, 1, w, K, 0, dx, K, None, None, 0, M, K, N), reps),
        "dgrad rt_gemm_wp": time_it(lambda: wp(dy, dx, M, K, N, K, 1), reps),
    }
    ref = x[:64].double() @ w.double().T
    wp(x, y, M, N, K, K, 0); torch.cuda.synchronize()
    e1 = float((y[:64].double() - ref).abs().max() / ref.abs().max())
    ref = dy[:64].double() @ w.double()
    wp(dy, dx, M, K, N, K, 1); torch.cuda.synchronize()
    e2 = float((dx[:64].double() - ref).abs().max() / ref.abs().max())
    print(f"{M}x{N}x{K}: " + "  ".join(f"{k} {v:.1f} us = {fl / v / 1e6:.0f} TF" for k, v in t.items()) + f"  relerr fwd {e1:.1e} dgrad {e2:.1e}")
