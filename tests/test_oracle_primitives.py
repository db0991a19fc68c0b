"""Pins the oracle (oracle/primitives.py, oracle/dense.py) to the reference:
(1) golden vectors produced by the reference's utils.py (tests/golden/make_golden.py),
(2) the known answers of the reference's own tests/test_utils.py.
The product's host-side text functions are held to the same vectors."""
import math

import numpy as np
import pytest

from conftest import unarr
from oracle import dense as OD
from oracle import primitives as OP
from review_recommender_amd import synth, text


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.dtype == b.dtype and a.shape == b.shape
    assert np.array_equal(a, b, equal_nan=True)


def test_l2_normalize_golden(golden_primitives):
    for case in golden_primitives["l2_normalize"]:
        same(OP.l2_normalize(unarr(case["x"])), unarr(case["y"]))


def test_minmax_golden(golden_primitives):
    for case in golden_primitives["minmax_normalize"]:
        same(OP.minmax_normalize(unarr(case["x"])), unarr(case["y"]))


def test_minmax_cli_copy_returns_empty_unchanged():
    x = np.array([], dtype=np.float64)
    assert OP.minmax_normalize(x, empty_passthrough=True).dtype == np.float64   # app/test.py:115
    assert OP.minmax_normalize(x).dtype == np.float32                          # utils.py:48-49


@pytest.mark.parametrize("mod", [OP, text])
def test_tokenize_and_groups_golden(golden_primitives, mod):
    for case in golden_primitives["tokenize_query"]:
        assert mod.tokenize_query(case["q"]) == case["tokens"]
    for case in golden_primitives["build_gate_groups"]:
        assert [sorted(g) for g in mod.build_gate_groups(case["q"])] == case["groups"]


@pytest.mark.parametrize("mod", [OP, text])
def test_gate_factor_golden(golden_primitives, mod):
    for c in golden_primitives["calculate_gate_factor"]:
        groups = mod.build_gate_groups(c["q"])
        f, hits, total = mod.calculate_gate_factor(c["text"], groups, c["penalty"])
        assert (f, hits, total) == (c["factor"], c["hits"], c["total"])


def test_bayesian_prior_golden(golden_primitives):
    for c in golden_primitives["bayesian_prior"]:
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                y = OP.bayesian_prior(unarr(c["avg"]), unarr(c["n"]), c["C"], c["gmean"])
        same(y, unarr(c["y"]))


def test_trust_golden(golden_primitives):
    for c in golden_primitives["trust_score_from_reviews"]:
        same(OP.trust_score_from_reviews(unarr(c["n"]), c["min_reviews"], c["saturation"]), unarr(c["y"]))


def test_dense_golden_is_reproduced_by_the_oracle(golden_dense):
    n, dim, seed_v, seed_q, k = golden_dense["recipe"].tolist()
    V = synth.unit_rows(n, dim, seed_v)
    Q = synth.unit_rows(golden_dense["rows"].shape[0], dim, seed_q)
    for i, q in enumerate(Q):
        rows, sims = OD.cosine_similarity_search(q, V, k)
        assert np.array_equal(rows, golden_dense["rows"][i])
        assert np.array_equal(sims, golden_dense["sims"][i])
        assert rows.dtype == np.int64 and sims.dtype == np.float32


# ---- the reference's own known answers (tests/test_utils.py), as data ----
def test_reference_kats():
    # test_utils.py:44-49, 53-57, 61-64
    y = OP.minmax_normalize(np.array([1.0, 2.0, 3.0, 4.0, 5.0]))
    assert y[0] == 0.0 and y[-1] == 1.0
    assert np.array_equal(OP.minmax_normalize(np.array([3.0, 3.0, 3.0, 3.0])), np.zeros(4, np.float32))
    assert len(OP.minmax_normalize(np.array([]))) == 0
    # test_utils.py:22-40
    z = OP.l2_normalize(np.array([[0.0, 0.0], [3.0, 4.0]]))
    assert z[0, 0] == 0.0 and z[0, 1] == 0.0 and abs(np.linalg.norm(z[1]) - 1.0) < 1e-5
    # test_utils.py:121-147
    groups = [{"yellow", "mustard", "gold"}, {"cat", "cats", "kitten"}, {"sock", "socks"}]
    assert OP.calculate_gate_factor("yellow cat socks soft comfortable", groups, 0.5) == (1.0, 3, 3)
    assert OP.calculate_gate_factor("yellow comfortable shoes", groups, 0.5) == (0.25, 1, 3)
    # test_utils.py:181-208
    emb = np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]], dtype=np.float32)
    rows, sims = OD.cosine_similarity_search(np.array([0.0, 1.0], dtype=np.float32), emb, 2)
    assert len(rows) == 2 and rows[0] == 1 and sims[0] == 1.0
    rows, sims = OD.cosine_similarity_search(np.array([0.0, 1.0], dtype=np.float32), emb[:2], 10)
    assert len(rows) == 2 and len(sims) == 2
    # test_utils.py:71-90 (membership)
    toks = OP.tokenize_query("best wireless headphones for music")
    assert {"best", "wireless", "headphones", "music"} <= set(toks) and "for" not in toks
    # SURVEY 8a9 probed values of trust (min=8, sat=80)
    t = OP.trust_score_from_reviews(np.array([0, 5, 10, 50, 100]), 8, 80)
    np.testing.assert_allclose(t, [0, 0.538093, 0.81826586, 0.9578902, 1.0], rtol=1e-6)
    # top_k == 0 returns two empty arrays
    rows, sims = OD.cosine_similarity_search(np.array([0.0, 1.0], dtype=np.float32), emb, 0)
    assert len(rows) == 0 and len(sims) == 0


def test_index_time_tokenizer_differs_from_query_time():
    # nlp/12_product_prep.py:75-78: bigger stop list and len > 1
    assert text.tokenize_document("I was at a USB c hub") == ["usb", "hub"]
    assert text.tokenize_query("I was at a USB c hub") == ["i", "was", "at", "usb", "c", "hub"]
    assert len(text.tokenize_document("x1 " * 6000)) == 5000


def test_nan_rating_zeroes_the_whole_rating_prior():
    # SURVEY section 7 "NaN semantics": one NaN star -> minmax sees non-finite -> zeros
    n = np.array([10, 20, 30])
    r = np.array([4.0, np.nan, 3.0])
    pr = OP.bayesian_prior(r, n, 20.0)
    assert math.isnan(pr[1])
    assert np.array_equal(OP.minmax_normalize(pr), np.zeros(3, np.float32))
