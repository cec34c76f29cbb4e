"""The oracle restatement against the committed golden vectors (generated from the compiled reference by
tests/golden/make_golden.py).  Runs without the reference and without a GPU."""
import numpy as np

import cases
from util import assert_bitexact


def test_oracle_matches_reference_vectors(oracle, oracle_backend):
    gold = cases.load_golden()
    got = cases.golden_outputs(oracle)
    assert set(got) == set(gold)
    for k in sorted(gold):
        if k.startswith("cgstat"):
            assert got[k][0] == gold[k][0], k                      # iteration count
            assert_bitexact(np.float32(got[k][1:]), np.float32(gold[k][1:]), k)
        else:
            assert_bitexact(got[k], gold[k], k)
