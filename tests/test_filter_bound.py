"""The error bound the filter scan's selection relies on (csrc/rr_dense_flt.hip, DESIGN K1a''), checked
in numpy: for bf16-rounded operands  |s~ - s| <= max||a - a~|| ||q~|| + max||a|| ||q - q~||  and the
containment argument built on it -- every row of the exact top-pool has s~ >= tau~ - 2 eps."""
import numpy as np
import pytest

from oracle.dense import round_to_bf16


def _eps(V, q):
    Vr, qr = round_to_bf16(V).astype(np.float64), round_to_bf16(q[None, :])[0].astype(np.float64)
    V64, q64 = V.astype(np.float64), q.astype(np.float64)
    row_norm = np.linalg.norm(V64, axis=1).max()
    row_delta = np.linalg.norm(V64 - Vr, axis=1).max()
    eps = row_delta * np.linalg.norm(qr) + row_norm * np.linalg.norm(q64 - qr)
    return Vr, qr, eps


@pytest.mark.parametrize("kind", ["unit", "scaled", "aligned"])
def test_bf16_filter_scores_stay_within_the_bound(kind):
    rng = np.random.default_rng(7)
    V = rng.standard_normal((20000, 384)).astype(np.float32)
    q = rng.standard_normal(384).astype(np.float32)
    if kind == "unit":
        V /= np.linalg.norm(V, axis=1, keepdims=True)
        q /= np.linalg.norm(q)
    elif kind == "scaled":
        V *= rng.uniform(0.01, 30.0, (len(V), 1)).astype(np.float32)
        q *= np.float32(7.5)
    else:   # rows proportional to the query, every product of one sign, mantissas that round badly
        V = (np.abs(q)[None, :] * rng.uniform(0.5, 1.5, (len(V), 1))).astype(np.float32) * np.float32(1 + 2 ** -9)
        q = np.abs(q)
    Vr, qr, eps = _eps(V, q)
    s = V.astype(np.float64) @ q.astype(np.float64)
    s_approx = Vr @ qr
    assert np.abs(s_approx - s).max() <= eps * (1 + 1e-12)
    # containment: tau~ from the approximate scores (here: the exact pool-th largest, the kernel uses a lower
    # bound of it), candidates = rows with s~ >= tau~ - 2 eps must hold the exact top-pool
    pool = 150
    tau = np.sort(s_approx)[-pool]
    cand = set(np.nonzero(s_approx >= tau - 2 * eps)[0].tolist())
    top = set(np.argsort(-s, kind="stable")[:pool].tolist())
    assert top <= cand
    assert (s[list(top)] >= tau - eps).all()          # the row filter after rescoring keeps them


def test_bound_is_not_vacuous_for_unit_vectors():
    rng = np.random.default_rng(3)
    V = rng.standard_normal((5000, 384)).astype(np.float32)
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    q = V[17].copy()
    _, _, eps = _eps(V, q)
    assert 1e-3 < eps < 8e-3          # ~2 x 2^-8 / sqrt(3): a few thousandths of the score range


def _fp32_chain(rows, q):
    """The per-row arithmetic of rr_rescore_chain on fp32 operands: lane j of 16 takes 16-byte units j, j + 16, ... and
    runs one fmaf chain over its 24 elements; the 16 partial sums are added pairwise (xor 8, 4, 2, 1).  float64 products
    rounded once to float32 stand in for fmaf (exact here: a float32 product has 48 significant bits, the sum of it
    and a float32 accumulator rounds once from float64 except in double-rounding ties that do not move the bound)."""
    n = rows.shape[0]
    lanes = np.zeros((n, 16), dtype=np.float32)
    r4 = rows.reshape(n, 6, 16, 4)
    q4 = q.reshape(6, 16, 4)
    for i in range(6):
        for e in range(4):
            lanes = (lanes.astype(np.float64) + r4[:, i, :, e].astype(np.float64) * q4[i, :, e].astype(np.float64)).astype(np.float32)
    for step in (8, 4, 2, 1):
        lanes = lanes + lanes[:, np.arange(16) ^ step]
    return lanes[:, 0]


def _plane_chain(rows, q):
    """The plane pre-scoring of rr_rescore_chain: lane j of 16 takes the 16-byte units j, j + 16, j + 32 of the bf16 plane
    row (8 elements each: dims 8 (j + 16 i) .. + 7), one fmaf chain over its 24 elements, pairwise sum of the 16 partials."""
    n = rows.shape[0]
    lanes = np.zeros((n, 16), dtype=np.float32)
    r8 = rows.reshape(n, 3, 16, 8)
    q8 = q.reshape(3, 16, 8)
    for i in range(3):
        for e in range(8):
            lanes = (lanes.astype(np.float64) + r8[:, i, :, e].astype(np.float64) * q8[i, :, e].astype(np.float64)).astype(np.float32)
    for step in (8, 4, 2, 1):
        lanes = lanes + lanes[:, np.arange(16) ^ step]
    return lanes[:, 0]


@pytest.mark.parametrize("kind", ["unit", "scaled"])
def test_plane_prescoring_of_rescored_rows_never_drops_a_pool_row(kind):
    """rr_rescore_chain scores a candidate row on its bf16 plane row first (sum a~_k q_k with the fp32 query) and runs
    the exact chain only if that reaches tau - 1.01 eps, tau = tau~ - 1.02 eps: a row of the exact top-pool (exact score
    >= tau~ - eps) always does, and at least `pool` rows do."""
    rng = np.random.default_rng(11)
    V = rng.standard_normal((20000, 384)).astype(np.float32)
    q = rng.standard_normal(384).astype(np.float32)
    if kind == "unit":
        V /= np.linalg.norm(V, axis=1, keepdims=True)
        q /= np.linalg.norm(q)
    else:
        V *= rng.uniform(0.01, 30.0, (len(V), 1)).astype(np.float32)
        q *= np.float32(7.5)
    Vr, qr, eps = _eps(V, q)
    pool = 150
    s_scan = Vr @ qr                                          # what the filter scan estimates
    tau_est = np.sort(s_scan)[-pool]
    exact = _fp32_chain(V, q)
    est = _plane_chain(Vr.astype(np.float32), q)              # the plane row against the fp32 query
    assert np.abs(est.astype(np.float64) - exact.astype(np.float64)).max() <= eps
    pre_thr = (tau_est - 1.02 * eps) - 1.01 * eps
    takes_chain = est >= pre_thr
    needed = exact >= tau_est - eps
    assert not (needed & ~takes_chain).any()
    assert takes_chain.sum() >= pool
    top = np.argsort(-exact.astype(np.float64), kind="stable")[:pool]
    assert takes_chain[top].all()
    assert takes_chain.mean() < 0.5                           # and it does skip rows: not a vacuous test
