"""Host-side replay of the tile / chunk bookkeeping of K6v2 (csrc/rt_attention_v2.hip: v2_hstu_fwd / bwd_dq / bwd_dkv kernels): which
(query, key) pairs each pass visits, in which chunk, and whether a chunk starts an owner row's accumulator from zero or continues what an
earlier chunk stored.  Pure integer logic — the arithmetic itself is checked on the GPU (tests/test_packed_hstu_gpu.py)."""
import numpy as np
import pytest

HCH, NW = 192, 8


def my_tiles(wave, n_tiles, heavy_last):
    """for_my_tiles<NW, HEAVY_LAST>: the o-th heaviest owner tile -> wave, zigzag of period 2 NW."""
    out = []
    for base in range(0, n_tiles, 2 * NW):
        for o in (base + wave, base + 2 * NW - 1 - wave):
            if o < n_tiles:
                out.append(n_tiles - 1 - o if heavy_last else o)
    return out


def elements(t):
    """the 8 partner rows (local to the chunk) of lane group g, register e, of 32-row tile t"""
    return {(g, e): t * 32 + 16 * (e >> 2) + 4 * g + (e & 3) for g in range(4) for e in range(8)}


def replay_query_owner(n):
    """forward / dQ: a lane owns a query, chunks of keys.  -> visits[q, k] and the accumulator protocol violations"""
    visits = np.zeros((n, n), dtype=np.int32)
    stored = np.zeros(n, dtype=bool)            # an earlier chunk stored this query's accumulator
    bad = []
    n_tiles = (n + 15) >> 4
    for c0 in range(0, n, HCH):
        ln = min(HCH, n - c0)
        touched = []
        for wave in range(NW):
            for qt in my_tiles(wave, n_tiles, True):
                if qt * 16 + 15 < c0:
                    continue
                for i in range(16):
                    qrow = qt * 16 + i
                    if qrow >= n:
                        continue
                    if c0 > 0 and not stored[qrow]:
                        bad.append(("load of a row never stored", qrow, c0))
                    if c0 == 0 and stored[qrow]:
                        bad.append(("zero start over a stored row", qrow, c0))
                    touched.append(qrow)
                    t_last = min((qt * 16 + 15 - c0) >> 5, (ln - 1) >> 5)
                    for t in range(t_last + 1):
                        for kl in elements(t).values():
                            key = c0 + kl
                            if kl < ln and key <= qrow:
                                visits[qrow, key] += 1
        stored[touched] = True
    return visits, bad


def replay_key_owner(n):
    """dK / dV: a lane owns a key, chunks of queries"""
    visits = np.zeros((n, n), dtype=np.int32)
    stored = np.zeros(n, dtype=bool)
    bad = []
    n_tiles = (n + 15) >> 4
    for c0 in range(0, n, HCH):
        ln = min(HCH, n - c0)
        touched = []
        for wave in range(NW):
            for kt in my_tiles(wave, n_tiles, False):
                if kt * 16 > c0 + ln - 1:
                    continue
                first = c0 <= kt * 16
                for i in range(16):
                    krow = kt * 16 + i
                    if krow >= n:
                        continue
                    if not first and not stored[krow]:
                        bad.append(("load of a row never stored", krow, c0))
                    if first and stored[krow]:
                        bad.append(("zero start over a stored row", krow, c0))
                    touched.append(krow)
                    t_first = max(kt * 16 - c0, 0) >> 5
                    for t in range(t_first, ((ln - 1) >> 5) + 1):
                        for ql in elements(t).values():
                            q = c0 + ql
                            if ql < ln and krow <= q:
                                visits[q, krow] += 1
        stored[touched] = True
    return visits, bad


@pytest.mark.parametrize("n", [1, 7, 16, 31, 32, 33, 191, 192, 193, 200, 383, 384, 385, 400, 511, 512])
def test_every_causal_pair_is_visited_exactly_once(n):
    want = np.tril(np.ones((n, n), dtype=np.int32))
    for replay in (replay_query_owner, replay_key_owner):
        visits, bad = replay(n)
        assert not bad, bad[:3]
        np.testing.assert_array_equal(visits, want)


def test_owner_tiles_are_dealt_once_each():
    for n_tiles in (1, 5, 13, 16, 17, 32):
        for heavy_last in (True, False):
            got = sorted(t for w in range(NW) for t in my_tiles(w, n_tiles, heavy_last))
            assert got == list(range(n_tiles))
