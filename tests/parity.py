"""Comparison helpers for top-k parity (test infrastructure).

fp32 dot products summed in a different order than BLAS differ by ~1e-7, so two
rows whose scores are closer than that may swap.  IDs are therefore required to
be bit-exact wherever the oracle's float64 gap to the neighbours exceeds
``tie_eps``; inside a tie band the sets must agree."""
import numpy as np


def assert_topk_matches(rows, scores, ref_sims64, k, tie_eps=4e-7, score_tol=1e-5):
    """rows/scores: candidate top-k (desc).  ref_sims64: float64 scores of ALL rows."""
    rows = np.asarray(rows)
    scores = np.asarray(scores, dtype=np.float64)
    n = len(ref_sims64)
    k = min(k, n)
    assert rows.shape == (k,) and len(set(rows.tolist())) == k, "k distinct rows expected"
    assert np.all(np.diff(scores) <= 0), "scores must be non-increasing"
    # scores agree with the float64 yardstick
    np.testing.assert_allclose(scores, ref_sims64[rows], atol=score_tol, rtol=0)
    order = np.lexsort((np.arange(n), -ref_sims64))
    ref_rows = order[:k]
    kth = ref_sims64[ref_rows[-1]]
    # membership: everything clearly above the k-th score is present, nothing clearly below
    must = set(order[ref_sims64[order] > kth + tie_eps].tolist())
    may = set(order[ref_sims64[order] >= kth - tie_eps].tolist())
    got = set(rows.tolist())
    assert must <= got, f"missing rows {sorted(must - got)[:5]}"
    assert got <= may, f"unexpected rows {sorted(got - may)[:5]}"
    # order: positions may differ only inside tie bands
    for i, (a, b) in enumerate(zip(rows, ref_rows)):
        if a != b:
            assert abs(ref_sims64[a] - ref_sims64[b]) <= tie_eps, \
                f"rank {i}: row {a} vs oracle {b}, gap {abs(ref_sims64[a] - ref_sims64[b]):.3e}"


def min_gap(ref_sims64, k):
    top = np.sort(ref_sims64)[::-1][:k + 1]
    return float(np.min(-np.diff(top))) if len(top) > 1 else float("inf")
