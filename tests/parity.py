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


def assert_ranking_matches(got_ids, want_ids, want_final, band, pool_ids=None, pool_final=None):
    """Ranked ids against the oracle's ranking, compared PER TIE BAND: consecutive oracle entries whose finals differ by at
    most ``band`` (twice the score tolerance: two scores each within tol may swap) form a band, inside which the order is
    free and outside which it is exact.  A band cut off by k may also hold other candidates of the oracle's pool
    (``pool_ids`` / ``pool_final``) whose final is within ``band`` of the band's."""
    got_ids, want_ids = list(got_ids), list(want_ids)
    wf = np.asarray(want_final, dtype=np.float64)
    assert len(got_ids) == len(want_ids) == len(wf)
    i, n = 0, len(want_ids)
    while i < n:
        j = i + 1
        while j < n and abs(wf[j - 1] - wf[j]) <= band:
            j += 1
        if j == n and pool_ids is not None:
            pf = np.asarray(pool_final, dtype=np.float64)
            allowed = {p for p, f in zip(pool_ids, pf) if wf[j - 1] - band <= f <= wf[i] + band}
            assert set(got_ids[i:j]) <= allowed | set(want_ids[i:j]), (i, j)
        else:
            assert set(got_ids[i:j]) == set(want_ids[i:j]), f"ranks {i}..{j - 1}: {got_ids[i:j]} vs {want_ids[i:j]}"
        i = j
