"""Pins `oracle/ranker_oracle.py` to the reference's own outputs (tests/golden/ranker_golden.npz).

The golden file holds the reference's known-answer inputs (tests/models/rank/test_rank.py:51-64 of the
reference) in every rank() mode it tests plus seeded random cases, with outputs produced by the
unmodified `TorchRanker` (rank_torch.py).  CPU-only.
"""
import numpy as np
import pytest

from conftest import load_ranker_golden
from oracle import ranker_oracle

CASES = load_ranker_golden()


@pytest.mark.parametrize("case", CASES, ids=[c["tag"] for c in CASES])
def test_oracle_matches_reference(case):
    su, it, sc = ranker_oracle.rank(
        case["users"], case["items"], case["sids"], k=case["k"], filter_pairs_csr=case["filt"],
        sorted_object_whitelist=case["wl"], distance=case["distance"],
    )
    np.testing.assert_array_equal(su, case["ref_subjects"])
    np.testing.assert_array_equal(it, case["ref_items"])
    # reference compares scores to 5 decimals (tests/models/rank/test_rank.py:27, EPS_DIGITS)
    np.testing.assert_allclose(sc, case["ref_scores"], rtol=2e-5, atol=1e-5)


def test_reference_kat_values():
    """The literal expected values of the reference's KAT (tests/models/rank/test_rank.py:66-77)."""
    users = np.array([[-4, 0, 3], [0, 1, 2]], dtype=np.float32)
    items = np.array([[-4, 0, 3], [0, 2, 4], [1, 10, 100]], dtype=np.float32)
    _, it, sc = ranker_oracle.rank(users, items, [0, 1], k=3, distance="dot")
    assert it.tolist() == [2, 0, 1, 2, 1, 0]
    assert sc.tolist() == [296, 25, 12, 210, 10, 6]
    _, it, sc = ranker_oracle.rank(users, items, [0, 1], k=3, distance="cosine")
    assert it.tolist() == [0, 2, 1, 1, 2, 0]
    np.testing.assert_almost_equal(sc, [1, 0.5890328, 0.5366563, 1, 0.9344414, 0.5366563], decimal=5)
    _, it, sc = ranker_oracle.rank(users, items, [0, 1], k=3, distance="euclidean")
    assert it.tolist() == [0, 1, 2, 1, 0, 2]
    np.testing.assert_almost_equal(
        sc, [0, 4.58257569, 97.64220399, 2.23606798, 4.24264069, 98.41747812], decimal=5
    )


def test_csr_row_mismatch_raises():
    from scipy import sparse

    users = np.zeros((2, 3), np.float32)
    items = np.zeros((3, 3), np.float32)
    with pytest.raises(ValueError):
        ranker_oracle.rank(users, items, [0, 1], k=1, filter_pairs_csr=sparse.csr_matrix(np.zeros((3, 3))))
