"""The error bounds the two-stage top-k proof rests on (include/rectools_hip.h K12c; rt_topk.hip `err_coef`, rt_hm_image.hip), pinned on the
CPU: a numpy restatement of the two coarse images — (h, m): two truncated bf16 planes; one plane: round-to-nearest bf16 — and of the
coarse score they produce, checked against the coefficients the exact pass uses in `e_k - eps > tau`.  A coefficient that is too small makes
the proof unsound (a user could be reported "proven" with a wrong top-k); rows built from the worst values of each rounding reach within a
few per cent of the bounds, so both directions are pinned.  No GPU, no library: the coefficients are restated from the source and the test
fails if the source text stops carrying them."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bf16_trunc(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def bf16_rne(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def hm_planes(x):
    """rt_to_hm_rows, normalize = 0: word = (h << 16) | m, h = trunc_bf16(x), m = trunc_bf16(x - h)."""
    h = bf16_trunc(x)
    m = bf16_trunc(x - h)
    return h, m


def coef_hm(d):       # rt_topk.hip: r.err_coef of the (h, m) coarse pass
    return 1.03 * (2.0 ** -14 + 5.0 * d * 2.0 ** -24)


def coef_one_plane(d):  # rt_topk.hip: r.err_coef with ts->h_only
    return 1.03 * (2.0 ** -7 + 2.0 ** -16 + 2.0 * d * 2.0 ** -24)


def test_the_source_carries_these_coefficients():
    src = open(os.path.join(ROOT, "rectools_amd", "csrc", "rt_topk.hip")).read()
    hm = re.search(r"r\.err_coef = 1\.03f \* \(([0-9.e+-]+)f \+ 5\.0f \* \(float\)d \* ([0-9.e+-]+)f\);", src)
    one = re.search(r"if \(ts->h_only\) r\.err_coef = 1\.03f \* \(([0-9.e+-]+)f \+ ([0-9.e+-]+)f \+ 2\.0f \* \(float\)d \* ([0-9.e+-]+)f\);", src)
    assert hm and one, "rt_topk.hip no longer sets err_coef in the form this test restates"
    assert float(hm.group(1)) == 2.0 ** -14 and float(hm.group(2)) == 2.0 ** -24
    assert float(one.group(1)) == 2.0 ** -7 and float(one.group(2)) == 2.0 ** -16 and float(one.group(3)) == 2.0 ** -24


def _coarse_f32(a_planes, b_planes, order):
    """fp32 accumulation of the bf16 products in the given order of k (the matrix instruction's own order is not specified; the bound must
    hold for any) — products of two bf16 values are exact in fp32."""
    acc = np.float32(0.0)
    for k in order:
        for a in a_planes:
            for b in b_planes:
                acc = np.float32(acc + np.float32(a[k]) * np.float32(b[k]))
    return float(acc)


def _exact_f32(u, v, order):
    acc = np.float32(0.0)
    for k in order:
        acc = np.float32(acc + np.float32(u[k] * v[k]))      # an fp32 multiply-add chain (the exact pass's arithmetic class)
    return float(acc)


def _worst_rows(kind, d, rng):
    """Rows that maximise the image's rounding error, all products of one sign (so that sum |u_k v_k| = |u| |v| when u is parallel to v)."""
    if kind == "halfway":          # 1 + 2^-8 scaled by powers of two: exactly halfway between two bf16 neighbours, ties-to-even rounds DOWN
        base = np.float32(1.0 + 2.0 ** -8)
    elif kind == "all_ones":       # every mantissa bit set: the largest truncation residue of both planes
        base = np.float32(2.0 - 2.0 ** -23)
    else:
        raise ValueError(kind)
    scale = np.float32(2.0) ** rng.integers(-3, 4, size=1).astype(np.float32)
    u = np.full(d, base, dtype=np.float32) * scale
    return u, u.copy()


@pytest.mark.parametrize("d", [64, 256, 512, 2048])
def test_one_plane_bound_holds_and_is_nearly_reached(d):
    rng = np.random.default_rng(d)
    worst = 0.0
    cases = [_worst_rows("halfway", d, rng), _worst_rows("all_ones", d, rng)]
    for _ in range(20):           # random rows, random scales, random signs
        u = (rng.normal(size=d) * 10.0 ** rng.uniform(-3, 3)).astype(np.float32)
        v = (rng.normal(size=d) * 10.0 ** rng.uniform(-3, 3)).astype(np.float32)
        cases.append((u, v))
    for _ in range(5):            # halfway mantissas with random signs and exponents: parallel rows, the per-product error at its maximum
        e = np.float32(2.0) ** rng.integers(-6, 7, size=d).astype(np.float32)
        u = (np.float32(1.0 + 2.0 ** -8) * e).astype(np.float32) * rng.choice(np.float32([-1, 1]), size=d)
        cases.append((u, u.copy()))
    for u, v in cases:
        uh, vh = bf16_rne(u), bf16_rne(v)
        order = rng.permutation(d)
        coarse = _coarse_f32([uh], [vh], order)
        exact = _exact_f32(u, v, order[::-1])
        scale = float(np.linalg.norm(u.astype(np.float64)) * np.linalg.norm(v.astype(np.float64)))
        rel = abs(coarse - exact) / scale
        assert rel <= coef_one_plane(d), (rel, coef_one_plane(d))
        worst = max(worst, rel)
    assert worst >= 0.9 * (2.0 ** -7)          # the halfway rows come within a few per cent of the bound: it cannot be halved
    assert worst > 1.03 * (2.0 ** -8 + 2.0 ** -18 + 2.0 * d * 2.0 ** -24) or d > 8192   # (what the first version of the kernel charged)


@pytest.mark.parametrize("d", [64, 256, 512, 2048])
def test_hm_bound_holds(d):
    rng = np.random.default_rng(d + 1)
    cases = [_worst_rows("all_ones", d, rng), _worst_rows("halfway", d, rng)]
    for _ in range(20):
        u = (rng.normal(size=d) * 10.0 ** rng.uniform(-3, 3)).astype(np.float32)
        v = (rng.normal(size=d) * 10.0 ** rng.uniform(-3, 3)).astype(np.float32)
        cases.append((u, v))
    worst = 0.0
    for u, v in cases:
        order = rng.permutation(d)
        coarse = _coarse_f32(list(hm_planes(u)), list(hm_planes(v)), order)
        exact = _exact_f32(u, v, order[::-1])
        scale = float(np.linalg.norm(u.astype(np.float64)) * np.linalg.norm(v.astype(np.float64)))
        rel = abs(coarse - exact) / scale
        assert rel <= coef_hm(d), (rel, coef_hm(d))
        worst = max(worst, rel)
    assert worst >= 2.0 ** -17                  # the all-ones rows lose ~2^-15 per product to the dropped third plane (fp32 rounding of the
    #                                             two chains takes part of it back)


def test_plane_residues():
    """x = h + m + l with |l| < 2^-15 |x| (two truncated planes); |x - rne(x)| <= 2^-8 |x| (one rounded plane)."""
    rng = np.random.default_rng(0)
    x = (rng.normal(size=200_000) * 10.0 ** rng.uniform(-6, 6, size=200_000)).astype(np.float32)
    x = np.concatenate([x, np.float32([1.0 + 2.0 ** -8, 2.0 - 2.0 ** -23, 1.0, -3.0, 1e-30, 1e30])])
    h, m = hm_planes(x)
    l = x.astype(np.float64) - h.astype(np.float64) - m.astype(np.float64)
    assert bool((np.abs(l) < 2.0 ** -15 * np.abs(x.astype(np.float64))).all())
    r = bf16_rne(x)
    assert bool((np.abs(x.astype(np.float64) - r.astype(np.float64)) <= 2.0 ** -8 * np.abs(x.astype(np.float64))).all())
    assert float(np.max(np.abs(x.astype(np.float64) - r.astype(np.float64)) / np.abs(x.astype(np.float64)))) > 0.99 * 2.0 ** -8
