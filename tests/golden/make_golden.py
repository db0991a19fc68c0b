"""Generates tests/golden/*.json|npz by running the REFERENCE's own utils.py
(importable in the build container from /root/reference) on
seeded inputs.  The reference never travels to the GPU box: only these vectors do.

    python tests/golden/make_golden.py          # needs /root/reference

Files
  primitives.json   I/O of l2_normalize, minmax_normalize, tokenize_query,
                    build_gate_groups, calculate_gate_factor, bayesian_prior,
                    trust_score_from_reviews (reference utils.py outputs)
  dense_10k.npz     cosine_similarity_search of the reference on the 10k x 384
                    seed recipe (config 1): top-150 rows + scores for 4 queries
"""
import json
import pathlib
import sys

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
REF = pathlib.Path("/root/reference")
sys.path.insert(0, str(REF))
sys.path.insert(0, str(HERE.parent.parent))
import utils as ref  # noqa: E402  (the reference's utils.py)

from review_recommender_amd import synth  # noqa: E402


def arr(a):
    a = np.asarray(a)
    return {"dtype": str(a.dtype), "shape": list(a.shape),
            "data": [None if (isinstance(v, float) and v != v) else v
                     for v in a.astype(object).ravel().tolist()]}


def main():
    rng = np.random.default_rng(20251024)
    out = {"generator": "tests/golden/make_golden.py", "source": "reference utils.py",
           "numpy": np.__version__}

    # l2_normalize
    x = rng.standard_normal((5, 8)).astype(np.float32)
    x[2] = 0.0
    out["l2_normalize"] = [{"x": arr(x), "y": arr(ref.l2_normalize(x))},
                           {"x": arr(x.astype(np.float64)), "y": arr(ref.l2_normalize(x.astype(np.float64)))}]

    # minmax_normalize
    mm_inputs = [
        np.array([1.0, 2.0, 3.0, 4.0, 5.0]),
        np.array([3.0, 3.0, 3.0, 3.0]),
        np.array([], dtype=np.float64),
        rng.standard_normal(17).astype(np.float32),
        rng.standard_normal(17),
        np.array([0.1, np.nan, 0.7], dtype=np.float32),
        np.array([0.1, np.inf, 0.7]),
        (rng.random(150) * 1e-3 + 0.2).astype(np.float32),
        np.array([0.25, 0.25 + 5e-13]),
    ]
    out["minmax_normalize"] = [{"x": arr(v), "y": arr(ref.minmax_normalize(v))} for v in mm_inputs]

    # tokenizer / gate groups
    queries = ["best wireless headphones for music", "noise-cancelling headphones, really good!",
               "yellow cat socks", "random unique product", "it's the kid's toy", "The AND of",
               "reduced price bluetooth earbuds", "USB-C cable 2m", "", "golden retriever dog leash navy",
               "socks sock SOCKS keyboard keyboards wireless design theme dog cat noise anc"]
    out["tokenize_query"] = [{"q": q, "tokens": ref.tokenize_query(q)} for q in queries]
    out["build_gate_groups"] = [{"q": q, "groups": [sorted(g) for g in ref.build_gate_groups(q)]}
                                for q in queries]

    # gate factor
    texts = ["yellow cat socks soft comfortable", "yellow comfortable shoes", "Category: KITTENS and more",
             "", "noise canceling over-ear headset, mustard colour"]
    gf = []
    for q in queries:
        groups = ref.build_gate_groups(q)
        for t in texts:
            for pen in (0.0, 0.3, 0.5, 1.0):
                f, hits, total = ref.calculate_gate_factor(t, groups, pen)
                gf.append({"q": q, "text": t, "penalty": pen, "factor": f, "hits": hits, "total": total})
    out["calculate_gate_factor"] = gf

    # bayesian prior (pool mean, NaN ratings) and trust
    n = np.array([0, 1, 5, 10, 50, 100, 2500], dtype=np.int64)
    r = np.array([4.0, np.nan, 3.2, 5.0, 4.4, 1.5, 4.9])
    out["bayesian_prior"] = [
        {"avg": arr(r), "n": arr(n), "C": 20.0, "gmean": None, "y": arr(ref.bayesian_prior(r, n, 20.0))},
        {"avg": arr(r), "n": arr(n), "C": 5.0, "gmean": 4.0, "y": arr(ref.bayesian_prior(r, n, 5.0, 4.0))},
        {"avg": arr(np.array([np.nan, np.nan])), "n": arr(np.array([3, 4])), "C": 20.0, "gmean": None,
         "y": arr(ref.bayesian_prior(np.array([np.nan, np.nan]), np.array([3, 4]), 20.0))},
    ]
    out["trust_score_from_reviews"] = [
        {"n": arr(n), "min_reviews": mr, "saturation": sat,
         "y": arr(ref.trust_score_from_reviews(n, mr, sat))}
        for mr, sat in ((8, 50), (8, 80), (1, 80), (0, 0), (5, 80))]
    (HERE / "primitives.json").write_text(json.dumps(out, indent=1))

    # dense: config 1 recipe (10k x 384 fp32 unit rows, seed 1234), reference's own search
    V = synth.unit_rows(10_000, 384, 1234)
    Q = synth.unit_rows(4, 384, 4321)
    rows, sims = [], []
    for q in Q:
        i, s = ref.cosine_similarity_search(q, V, 150)
        rows.append(i)
        sims.append(s)
    np.savez_compressed(HERE / "dense_10k.npz", rows=np.stack(rows), sims=np.stack(sims),
                        recipe=np.array([10_000, 384, 1234, 4321, 150]))
    print("wrote", HERE / "primitives.json", HERE / "dense_10k.npz")


if __name__ == "__main__":
    main()
