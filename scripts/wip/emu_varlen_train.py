"""Lane-level numpy emulation of the three training kernels of rt_attention_varlen.hip (forward with dropout + lse, dQ pass with
delta and the pad keys' d_bv share, dK/dV pass) against a dense restatement of the PADDED window with explicit pad keys.
Checks the MFMA operand / accumulator index math, the masks, the dropout-mask keys and the virtual-key terms — not timing."""
import numpy as np

M32 = 0xFFFFFFFF
def row_of(r, half): return (r & 3) + 8 * (r >> 2) + 4 * half
def drop_hash(seed, bh, q, kp):
    x = (seed & M32) ^ ((q * 0x9E3779B1) & M32) ^ ((kp * 0x85EBCA77) & M32) ^ ((bh * 0xC2B2AE3D) & M32) ^ ((seed >> 32) & M32)
    x ^= x >> 16; x = (x * 0x7FEB352D) & M32; x ^= x >> 15
    return x
def kept(seed, bh, q, key, thr16): return ((drop_hash(seed, bh, q, key >> 1) >> (16 * (key & 1))) & 0xFFFF) >= thr16
def mfma(a, b, acc):
    A = np.zeros((32, 2)); Bm = np.zeros((2, 32))
    for l in range(64): A[l & 31, l >> 5] = a[l]; Bm[l >> 5, l & 31] = b[l]
    D = A @ Bm
    out = acc.copy()
    for l in range(64):
        for r in range(16): out[l, r] += D[row_of(r, l >> 5), l & 31]
    return out
L64 = np.arange(64)

def fwd_kernel(Q, K, V, bk, bv, n, window, HD, p_drop, seed, bh):
    scale = 1 / np.sqrt(HD); NS = HD // 2; NCB = HD // 32
    n32 = (n + 31) // 32 * 32
    Ks = np.zeros((n32, HD)); Vs = np.zeros((n32, HD)); Ks[:n] = K; Vs[:n] = V
    O = np.zeros((n, HD)); LSE = np.zeros(n)
    n_pad = max(window - n, 0); pads = bk is not None and n_pad > 0
    thr16 = int(np.float32(p_drop) * np.float32(65536.0)); inv_keep = 1 / (1 - p_drop) if p_drop > 0 else 1.0
    for qt in range(n32 // 32):
        qrow = qt * 32 + (L64 & 31)
        qf = np.array([[Q[min(qrow[l], n - 1), 2 * s + (l >> 5)] * scale for s in range(NS)] for l in range(64)])
        m = np.full(64, -np.inf); lsum = np.zeros(64); oT = np.zeros((NCB, 64, 16))
        for kt in range(qt + 1):
            sT = np.zeros((64, 16))
            for s in range(NS):
                sT = mfma(np.array([Ks[kt * 32 + (l & 31), 2 * s + (l >> 5)] for l in range(64)]), qf[:, s], sT)
            mx = m.copy()
            for l in range(64):
                for r in range(16):
                    jr = kt * 32 + row_of(r, l >> 5)
                    if not (jr <= qrow[l] and jr < n): sT[l, r] = -np.inf
                    mx[l] = max(mx[l], sT[l, r])
            mx = np.maximum(mx, mx[L64 ^ 32])
            alpha = np.where(np.isinf(m), 0.0, np.exp(m - mx))
            p = np.where(np.isinf(sT), 0.0, np.exp(sT - mx[:, None]))
            ps = p.sum(1); ps = ps + ps[L64 ^ 32]
            lsum = lsum * alpha + ps; m = mx
            if thr16:
                for l in range(64):
                    for r in range(16):
                        p[l, r] = p[l, r] * inv_keep if kept(seed, bh, int(qrow[l]), kt * 32 + row_of(r, l >> 5), thr16) else 0.0
            oT *= alpha[None, :, None]
            for r in range(16):
                for cb in range(NCB):
                    oT[cb] = mfma(np.array([Vs[kt * 32 + row_of(r, l >> 5), cb * 32 + (l & 31)] for l in range(64)]), p[:, r], oT[cb])
        if pads:
            dp = np.array([sum(qf[l, s] * bk[2 * s + (l >> 5)] for s in range(NS)) for l in range(64)]); dp = dp + dp[L64 ^ 32]
            mx = np.maximum(m, dp); alpha = np.where(np.isinf(m), 0.0, np.exp(m - mx)); e = np.exp(dp - mx)
            lsum = lsum * alpha + n_pad * e; m = mx
            if thr16: wv = np.array([sum(kept(seed, bh, int(qrow[l]), kk, thr16) for kk in range(n, n + n_pad)) for l in range(64)]) * inv_keep * e
            else: wv = n_pad * e
            for cb in range(NCB):
                for l in range(64):
                    for r in range(16): oT[cb, l, r] = oT[cb, l, r] * alpha[l] + wv[l] * bv[cb * 32 + row_of(r, l >> 5)]
        for l in range(64):
            if qrow[l] < n:
                LSE[qrow[l]] = m[l] + np.log(lsum[l])
                for cb in range(NCB):
                    for r in range(16): O[qrow[l], cb * 32 + row_of(r, l >> 5)] = oT[cb, l, r] / lsum[l]
    return O, LSE

def dq_kernel(Q, K, V, O, dO, LSE, bk, bv, n, window, HD, p_drop, seed, bh):
    scale = 1 / np.sqrt(HD); NS = HD // 2; NCB = HD // 32
    n32 = (n + 31) // 32 * 32
    Ks = np.zeros((n32, HD)); Vs = np.zeros((n32, HD)); Ks[:n] = K; Vs[:n] = V
    dQ = np.zeros((n, HD)); DELTA = np.zeros(n); dbv = np.zeros(HD)
    n_pad = max(window - n, 0); pads = bk is not None and n_pad > 0
    thr16 = int(np.float32(p_drop) * np.float32(65536.0)); inv_keep = 1 / (1 - p_drop) if p_drop > 0 else 1.0
    for qt in range(n32 // 32):
        qrow = qt * 32 + (L64 & 31); qok = qrow < n; g = np.minimum(qrow, n - 1)
        qf = np.array([[Q[g[l], 2 * s + (l >> 5)] * scale for s in range(NS)] for l in range(64)])
        dof = np.array([[dO[g[l], 2 * s + (l >> 5)] if qok[l] else 0.0 for s in range(NS)] for l in range(64)])
        dl = np.array([sum(dof[l, s] * O[g[l], 2 * s + (l >> 5)] for s in range(NS)) for l in range(64)]); dl = dl + dl[L64 ^ 32]
        lse = LSE[g]
        for l in range(64):
            if qok[l]: DELTA[qrow[l]] = dl[l]
        dqT = np.zeros((NCB, 64, 16))
        for kt in range(qt + 1):
            sT = np.zeros((64, 16)); dpT = np.zeros((64, 16))
            for s in range(NS):
                sT = mfma(np.array([Ks[kt * 32 + (l & 31), 2 * s + (l >> 5)] for l in range(64)]), qf[:, s], sT)
                dpT = mfma(np.array([Vs[kt * 32 + (l & 31), 2 * s + (l >> 5)] for l in range(64)]), dof[:, s], dpT)
            dsT = np.zeros((64, 16))
            for l in range(64):
                for r in range(16):
                    jr = kt * 32 + row_of(r, l >> 5)
                    ok = jr <= qrow[l] and jr < n
                    p = np.exp(sT[l, r] - lse[l]) if ok else 0.0
                    dp = dpT[l, r]
                    if thr16: dp = dp * inv_keep if kept(seed, bh, int(qrow[l]), jr, thr16) else 0.0
                    dsT[l, r] = p * (dp - dl[l])
            for r in range(16):
                for cb in range(NCB):
                    dqT[cb] = mfma(np.array([Ks[kt * 32 + row_of(r, l >> 5), cb * 32 + (l & 31)] for l in range(64)]), dsT[:, r], dqT[cb])
        if pads:
            sp = np.array([sum(qf[l, s] * bk[2 * s + (l >> 5)] for s in range(NS)) for l in range(64)]); sp = sp + sp[L64 ^ 32]
            dpp = np.array([sum(dof[l, s] * bv[2 * s + (l >> 5)] for s in range(NS)) for l in range(64)]); dpp = dpp + dpp[L64 ^ 32]
            e = np.exp(sp - lse)
            if thr16: kp = np.array([sum(kept(seed, bh, int(qrow[l]), kk, thr16) for kk in range(n, n + n_pad)) for l in range(64)]) * inv_keep
            else: kp = np.full(64, float(n_pad))
            dsp = e * (kp * dpp - n_pad * dl); wv = np.where(qok, e * kp, 0.0)
            for cb in range(NCB):
                for l in range(64):
                    for r in range(16):
                        c = cb * 32 + row_of(r, l >> 5)
                        dqT[cb, l, r] += dsp[l] * bk[c]
                        # every lane of a query adds its columns: both halves together cover each column once per query
                        dbv[c] += wv[l] * dO[g[l], c]
        for l in range(64):
            if qok[l]:
                for cb in range(NCB):
                    for r in range(16): dQ[qrow[l], cb * 32 + row_of(r, l >> 5)] = dqT[cb, l, r] * scale
    return dQ, DELTA, dbv

def dkv_kernel(Q, K, V, dO, LSE, DELTA, n, HD, p_drop, seed, bh):
    scale = 1 / np.sqrt(HD); NS = HD // 2; NCB = HD // 32
    n32 = (n + 31) // 32 * 32
    Qs = np.zeros((n32, HD)); Ds = np.zeros((n32, HD)); Qs[:n] = Q * scale; Ds[:n] = dO
    Ls = np.zeros(n32); Dl = np.zeros(n32); Ls[:n] = LSE; Dl[:n] = DELTA
    dK = np.zeros((n, HD)); dV = np.zeros((n, HD))
    thr16 = int(np.float32(p_drop) * np.float32(65536.0)); inv_keep = 1 / (1 - p_drop) if p_drop > 0 else 1.0
    for kt in range(n32 // 32):
        krow = kt * 32 + (L64 & 31); g = np.minimum(krow, n - 1)
        kf = np.array([[K[g[l], 2 * s + (l >> 5)] for s in range(NS)] for l in range(64)])
        vf = np.array([[V[g[l], 2 * s + (l >> 5)] for s in range(NS)] for l in range(64)])
        dkT = np.zeros((NCB, 64, 16)); dvT = np.zeros((NCB, 64, 16))
        for qt in range(kt, n32 // 32):
            sm = np.zeros((64, 16)); dpm = np.zeros((64, 16))
            for s in range(NS):
                sm = mfma(np.array([Qs[qt * 32 + (l & 31), 2 * s + (l >> 5)] for l in range(64)]), kf[:, s], sm)
                dpm = mfma(np.array([Ds[qt * 32 + (l & 31), 2 * s + (l >> 5)] for l in range(64)]), vf[:, s], dpm)
            pd = np.zeros((64, 16)); ds = np.zeros((64, 16))
            for l in range(64):
                for r in range(16):
                    qr = qt * 32 + row_of(r, l >> 5)
                    ok = krow[l] <= qr and qr < n and krow[l] < n
                    p = np.exp(sm[l, r] - Ls[qr]) if ok else 0.0
                    kf_ = 1.0
                    if thr16: kf_ = inv_keep if kept(seed, bh, qr, int(krow[l]), thr16) else 0.0
                    pd[l, r] = p * kf_; ds[l, r] = p * (dpm[l, r] * kf_ - Dl[qr])
            for r in range(16):
                for cb in range(NCB):
                    dvT[cb] = mfma(np.array([Ds[qt * 32 + row_of(r, l >> 5), cb * 32 + (l & 31)] for l in range(64)]), pd[:, r], dvT[cb])
                    dkT[cb] = mfma(np.array([Qs[qt * 32 + row_of(r, l >> 5), cb * 32 + (l & 31)] for l in range(64)]), ds[:, r], dkT[cb])
        for l in range(64):
            if krow[l] < n:
                for cb in range(NCB):
                    for r in range(16):
                        c = cb * 32 + row_of(r, l >> 5)
                        dK[krow[l], c] = dkT[cb, l, r]; dV[krow[l], c] = dvT[cb, l, r]
    return dK, dV

def dense_reference(Q, K, V, dO, bk, bv, n, window, HD, p_drop, seed, bh):
    scale = 1 / np.sqrt(HD); n_pad = max(window - n, 0) if bk is not None else 0
    thr16 = int(np.float32(p_drop) * np.float32(65536.0)); inv_keep = 1 / (1 - p_drop) if p_drop > 0 else 1.0
    Kall = np.vstack([K, np.tile(bk, (n_pad, 1))]) if n_pad else K      # real keys 0..n-1, pad keys n..n+n_pad-1 (the hash numbering)
    Vall = np.vstack([V, np.tile(bv, (n_pad, 1))]) if n_pad else V
    S = (Q @ Kall.T) * scale
    vis = np.zeros((n, n + n_pad), bool); vis[:, :n] = np.tril(np.ones((n, n), bool)); vis[:, n:] = True
    S = np.where(vis, S, -np.inf)
    lse = np.log(np.exp(S - S.max(1, keepdims=True)).sum(1)) + S.max(1)
    P = np.exp(S - lse[:, None])
    Mk = np.array([[ (kept(seed, bh, i, j, thr16) if thr16 else True) for j in range(n + n_pad)] for i in range(n)]) * inv_keep
    Pt = P * Mk
    O = Pt @ Vall
    dPt = dO @ Vall.T; dP = dPt * Mk
    delta = (dO * O).sum(1)
    dS = P * (dP - delta[:, None])
    dQ = scale * dS @ Kall; dKall = scale * dS.T @ Q; dVall = Pt.T @ dO
    return O, lse, dQ, dKall[:n], dVall[:n], (dVall[n:].sum(0) if n_pad else np.zeros(HD)), delta

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for HD, n, window, pads, p_drop in ((32, 37, 50, True, 0.25), (32, 33, 33, True, 0.0), (64, 40, 64, True, 0.2), (32, 20, 40, False, 0.3),
                                        (32, 64, 70, True, 0.1), (32, 1, 9, True, 0.5)):
        Q, K, V, dO = (rng.standard_normal((n, HD)) for _ in range(4))
        bk, bv = (rng.standard_normal(HD), rng.standard_normal(HD)) if pads else (None, None)
        seed, bh = 0x1234567890ABCDEF, 5
        rO, rlse, rdQ, rdK, rdV, rdbv, rdelta = dense_reference(Q, K, V, dO, bk, bv, n, window, HD, p_drop, seed, bh)
        O, LSE = fwd_kernel(Q, K, V, bk, bv, n, window, HD, p_drop, seed, bh)
        dQ, DELTA, dbv = dq_kernel(Q, K, V, O, dO, LSE, bk, bv, n, window, HD, p_drop, seed, bh)
        dK, dV = dkv_kernel(Q, K, V, dO, LSE, DELTA, n, HD, p_drop, seed, bh)
        err = lambda a, b: float(np.abs(a - b).max())   # noqa: E731
        print(f"HD{HD} n{n} window{window} pads{pads} p{p_drop}: O {err(O, rO):.1e} lse {err(LSE, rlse):.1e} dQ {err(dQ, rdQ):.1e} "
              f"dK {err(dK, rdK):.1e} dV {err(dV, rdV):.1e} dbv {err(dbv, rdbv):.1e} delta {err(DELTA, rdelta):.1e}")
