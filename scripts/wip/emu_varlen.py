"""Lane-level emulation of attn_varlen_fwd_kernel (index math only) against a dense reference with the virtual pad key."""
import numpy as np
rng = np.random.default_rng(0)
def row_of(r, half): return (r & 3) + 8 * (r >> 2) + 4 * half
def mfma(a, b, acc):
    # a[l] = A[row=l&31][k=l>>5]; b[l] = B[k=l>>5][col=l&31]; acc[l][r] <-> D[row_of(r,l>>5)][l&31]
    A = np.zeros((32, 2)); Bm = np.zeros((2, 32))
    for l in range(64): A[l & 31, l >> 5] = a[l]; Bm[l >> 5, l & 31] = b[l]
    D = A @ Bm
    out = acc.copy()
    for l in range(64):
        for r in range(16): out[l, r] += D[row_of(r, l >> 5), l & 31]
    return out
def kernel(Q, K, V, bk, bv, n, window, HD):
    scale = 1 / np.sqrt(HD); KS = HD + 1; NS = HD // 2; NCB = HD // 32
    n32 = (n + 31) // 32 * 32
    Ks = np.zeros((n32, KS)); Vs = np.zeros((n32, KS)); Ks[:n, :HD] = K; Vs[:n, :HD] = V
    O = np.full((n, HD), np.nan)
    n_pad = max(window - n, 0)
    for qt in range(n32 // 32):
        qrow = [qt * 32 + (l & 31) for l in range(64)]
        qf = np.array([[Q[min(qrow[l], n - 1), 2 * s + (l >> 5)] * scale for s in range(NS)] for l in range(64)])
        m = np.full(64, -np.inf); lsum = np.zeros(64); oT = np.zeros((NCB, 64, 16))
        for kt in range(qt + 1):
            sT = np.zeros((64, 16))
            for s in range(NS):
                a = np.array([Ks[kt * 32 + (l & 31), 2 * s + (l >> 5)] for l in range(64)])
                sT = mfma(a, qf[:, s], sT)
            mx = m.copy()
            for l in range(64):
                for r in range(16):
                    jr = kt * 32 + row_of(r, l >> 5)
                    if not (jr <= qrow[l] and jr < n): sT[l, r] = -np.inf
                    mx[l] = max(mx[l], sT[l, r])
            mx = np.maximum(mx, mx[np.arange(64) ^ 32])
            alpha = np.where(np.isinf(m), 0.0, np.exp(m - mx))
            p = np.where(np.isinf(sT), 0.0, np.exp(sT - mx[:, None]))
            ps = p.sum(1); ps = ps + ps[np.arange(64) ^ 32]
            lsum = lsum * alpha + ps; m = mx
            oT *= alpha[None, :, None]
            for r in range(16):
                for cb in range(NCB):
                    a = np.array([Vs[kt * 32 + row_of(r, l >> 5), cb * 32 + (l & 31)] for l in range(64)])
                    oT[cb] = mfma(a, p[:, r], oT[cb])
        if bk is not None and n_pad > 0:
            dp = np.array([sum(qf[l, s] * bk[2 * s + (l >> 5)] for s in range(NS)) for l in range(64)])
            dp = dp + dp[np.arange(64) ^ 32]
            mx = np.maximum(m, dp); alpha = np.where(np.isinf(m), 0.0, np.exp(m - mx)); w = n_pad * np.exp(dp - mx)
            lsum = lsum * alpha + w
            for cb in range(NCB):
                for l in range(64):
                    for r in range(16): oT[cb, l, r] = oT[cb, l, r] * alpha[l] + w[l] * bv[cb * 32 + row_of(r, l >> 5)]
        for l in range(64):
            if qrow[l] < n:
                for cb in range(NCB):
                    for r in range(16): O[qrow[l], cb * 32 + row_of(r, l >> 5)] = oT[cb, l, r] / lsum[l]
    return O
def reference(Q, K, V, bk, bv, n, window, HD):
    scale = 1 / np.sqrt(HD); n_pad = max(window - n, 0)
    O = np.zeros((n, HD))
    for i in range(n):
        lg = (K[:i + 1] @ Q[i]) * scale; vals = V[:i + 1]
        if bk is not None and n_pad > 0:
            lg = np.r_[lg, np.full(n_pad, (bk @ Q[i]) * scale)]; vals = np.vstack([vals, np.tile(bv, (n_pad, 1))])
        e = np.exp(lg - lg.max()); O[i] = (e[:, None] * vals).sum(0) / e.sum()
    return O
for HD in (32, 64):
    for n, window, pads in ((1, 10, True), (37, 50, True), (64, 64, True), (70, 200, False), (33, 40, True)):
        Q, K, V = rng.standard_normal((n, HD)), rng.standard_normal((n, HD)), rng.standard_normal((n, HD))
        bk, bv = (rng.standard_normal(HD), rng.standard_normal(HD)) if pads else (None, None)
        got, ref = kernel(Q, K, V, bk, bv, n, window, HD), reference(Q, K, V, bk, bv, n, window, HD)
        print(HD, n, window, pads, "max err", np.abs(got - ref).max())
