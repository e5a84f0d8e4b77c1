"""CPU oracle for the SASRec/BERT4Rec/HSTU fit()+recommend() hot path.

TEST INFRASTRUCTURE ONLY.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package; the product (`rectools_amd/`) never does, and fails loudly when its
HIP library is missing instead of falling back to anything here.

Parity status: PINNED.  `tests/golden/*.npz` were produced by running the *unmodified* reference
(`/root/reference`, RecTools v0.17.0) through `oracle/ref_shims.py` with `tests/golden/make_golden.py`;
`tests/test_oracle_golden.py` checks every function of this package against those vectors.
"""
