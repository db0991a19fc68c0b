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
