"""CPU oracle for the hybrid-retrieval hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain numpy / pure-Python restatement of the reference's
algorithm for the path SURVEY.md section 8 names (dense cosine top-k -> BM25 at the
candidate pool -> priors / trust / gate -> weighted blend -> top-k).  Each
function cites the reference file:line it follows.

Rules (enforced by tests/test_no_oracle_in_product.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
    ``bench.py`` may import anything from here;
  * the product package (``review-recommender_amd/``) never imports it and has
    no CPU fallback: without the HIP library it raises.

Pinning status
  * ``oracle.primitives`` (a1-a4, a7, a9-a11 of SURVEY section 8a) and
    ``oracle.dense.cosine_similarity_search`` are PINNED: checked in this
    container against the importable reference ``utils.py`` and against the
    known-answer values in the reference's ``tests/test_utils.py``; the
    resulting vectors are committed under ``tests/golden/`` together with
    the generating script ``tests/golden/make_golden.py``.
  * ``oracle.bm25`` restates the public algorithm of the third-party package
    ``rank_bm25`` (BM25Okapi, 0.2.x defaults k1=1.5, b=0.75, epsilon=0.25).
    The reference neither vendors nor pins that package and holds no golden
    value for it  ->  **BM25: parity unpinned** (checked only against
    hand-computed known answers on the reference's own 3-doc fixture corpus,
    tests/conftest.py:90-100).
  * ``oracle.pipeline`` restates ``run_search`` (app/app_product_search.py:
    245-317) and the CLI ``search`` (app/test.py:228-342).  Those modules
    cannot be imported here (streamlit / hub fetches), and the reference's
    tests pin no fused score  ->  **pipeline: parity unpinned beyond its
    pinned primitives**; it is a line-by-line restatement using the same
    numpy expressions so numpy's own dtype rules decide every rounding.
"""
