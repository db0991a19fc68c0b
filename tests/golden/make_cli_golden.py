#!/usr/bin/env python3
"""Generates tests/golden/cli_helpers.json, cli_search.json and index_tokenizer.json by RUNNING THE REFERENCE'S OWN
CLI module (/root/reference/app/test.py) and index builder (nlp/12_product_prep.py) in the build container.

    python tests/golden/make_cli_golden.py          # needs /root/reference; the reference never travels: only these vectors do

app/test.py imports with numpy + pandas only (its heavy imports sit inside _load_st_encoder / _load_cross_encoder /
_load_rankbm25, app/test.py:91-104).  Those three loader hooks are replaced with local stubs -- the injection the
reference's own integration test uses for the same objects (tests/test_integration.py:41-48):

  _load_st_encoder     -> an object whose encode([q], normalize_embeddings=True) returns the case's seeded unit vector
                          (bge-small is not available offline)
  _load_cross_encoder  -> tests/cli_worlds.FakeCrossEncoder (a deterministic function of the pair's text)
  _load_rankbm25       -> oracle.bm25.BM25OkapiOracle: rank_bm25 is absent from the reference tree and from this
                          image, so the BM25 ARITHMETIC inside these fixtures is the build's restatement (parity
                          unpinned, oracle/bm25.py); everything around it -- ensure_same_order, bm25_scores, minmax,
                          priors, rerank placement, snippets, gate, blend, sort, rounding -- is the reference's code.

Nothing else of the module is touched: search(args) (app/test.py:228-342) reads data/processed/* written here into a
temporary directory, runs, and writes its --json-out file, which is committed verbatim per case (`results`).  A
second run of every case with the module-global name `round` shadowed by the identity records the same rows before
the 4-dp rounding (`results_full`): same code path, unrounded values, so the tests can hold the 1e-5 bar.

  cli_helpers.json      I/O of minmax, bayesian_prior, cosine_search, ensure_same_order, bm25_scores (stub get_scores),
                        tokenize_query, _build_gate_groups, _gate_factor, l2_normalize on seeded inputs
  cli_search.json       search(args) on the worlds / configs of tests/cli_worlds.py
  index_tokenizer.json  nlp/12_product_prep.py:75-78 tokenize() on seeded strings
"""
import argparse
import contextlib
import importlib.util
import io
import json
import os
import pathlib
import pickle
import sys
import tempfile
import warnings

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REF = pathlib.Path("/root/reference")
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import cli_worlds as W  # noqa: E402
from oracle.bm25 import BM25OkapiOracle  # noqa: E402


def load_reference_module(rel, name):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def arr(a):
    a = np.asarray(a)
    return {"dtype": str(a.dtype), "shape": list(a.shape),
            "data": [None if (isinstance(v, float) and v != v) else v for v in a.astype(object).ravel().tolist()]}


class StubScores:
    """bm25_scores only needs get_scores (app/test.py:168-173)."""
    def __init__(self, scores):
        self.scores = scores

    def get_scores(self, tokens):
        return self.scores * float(len(tokens))


def helper_fixtures(ref):
    rng = np.random.default_rng(20261005)
    out = {"generator": "tests/golden/make_cli_golden.py", "source": "reference app/test.py", "numpy": np.__version__}
    mm = [np.array([1.0, 2.0, 3.0, 4.0, 5.0]), np.array([3.0, 3.0, 3.0]), np.array([], dtype=np.float64),
          np.array([], dtype=np.float32), rng.standard_normal(23).astype(np.float32), rng.standard_normal(23),
          np.array([0.1, np.nan, 0.7], dtype=np.float32), np.array([0.1, np.inf, 0.7]),
          (rng.random(100) * 1e-3 + 0.2).astype(np.float32), np.array([0.25, 0.25 + 5e-13])]
    out["minmax"] = [{"x": arr(v), "y": arr(ref.minmax(v))} for v in mm]            # app/test.py:114-119

    avg = np.round(np.clip(rng.normal(4.1, 0.6, 40), 1, 5), 3)
    n = np.floor(rng.lognormal(2.5, 1.2, 40))
    avg_nan = avg.copy()
    avg_nan[[3, 17]] = np.nan
    bp = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for a, c, g in ((avg, 20.0, None), (avg, 5.0, None), (avg_nan, 20.0, None), (avg, 20.0, 3.9),
                        (np.full(6, np.nan), 20.0, None)):
            nn = n[:len(a)]
            bp.append({"avg": arr(a), "n": arr(nn), "C": c, "global_mean": g,
                       "y": arr(ref.bayesian_prior(a, nn, C=c, global_mean=g))})   # app/test.py:121-123
    out["bayesian_prior"] = bp

    M = ref.l2_normalize(rng.standard_normal((500, 16)).astype(np.float32))        # app/test.py:109-112
    q = ref.l2_normalize(rng.standard_normal((1, 16)).astype(np.float32))[0]
    cs = []
    for k in (1, 10, 499, 500, 800):
        idx, s = ref.cosine_search(q, M, k)                                        # app/test.py:125-132
        cs.append({"k": k, "idx": arr(idx), "sims": arr(s)})
    out["cosine_search"] = {"seed": 20261005, "M": arr(M), "q": arr(q), "cases": cs}
    x = rng.standard_normal((4, 6)).astype(np.float32)
    x[1] = 0
    out["l2_normalize"] = [{"x": arr(x), "y": arr(ref.l2_normalize(x))}]

    import pandas as pd
    meta = pd.DataFrame({"sku": ["b", "a", "c", "d"]})
    eso = []
    for skus in (["a", "b", "c", "d"], ["d", "c", "b", "a", "x"], ["a", "b", "c"], ["a", "b", "c", "d", "a"]):
        eso.append({"meta_skus": meta["sku"].tolist(), "bm25_skus": skus,
                    "order": ref.ensure_same_order(meta, skus)})                   # app/test.py:159-166
    out["ensure_same_order"] = eso
    sc = rng.random(8)
    bs = []
    for order, top in ((None, [0, 3, 7]), ([7, 6, 5, 4, 3, 2, 1, 0], [0, 3, 7]), ([2, 2, 0, 1, 3, 4, 5, 6], [1, 0])):
        y = ref.bm25_scores(StubScores(sc), ["t1", "t2"], order, np.array(top))    # app/test.py:168-173
        bs.append({"scores_all": arr(sc * 2.0), "order": order, "top_idx": top, "y": arr(y)})
    out["bm25_scores"] = bs

    queries = ["wireless yellow cat socks", "Reduced price noise cancelling HEADPHONES", "it's a kid's toy for the dog",
               "blue navy cobalt azure olive emerald ivory rose violet amber tan slate keyboard design", "", "a an the",
               "golden retriever puppy leash", "USB-C charger 65w"]
    out["tokenize_query"] = [{"q": s, "y": ref.tokenize_query(s)} for s in queries]   # app/test.py:175-179
    out["build_gate_groups"] = [{"q": s, "y": [sorted(g) for g in ref._build_gate_groups(s)]} for s in queries]
    texts = ["Soft yellow socks with a cat print", "Category: bluetooth earbuds", "", "Mustard kitten SOCK, wireless"]
    gf = []
    for s in queries[:4]:
        groups = ref._build_gate_groups(s)                                         # app/test.py:62-78
        for t in texts:
            for pen in (0.0, 0.3, 0.5, 1.0):
                gf.append({"q": s, "text": t, "penalty": pen, "y": ref._gate_factor(t, groups, penalty=pen)})  # :80-88
    out["gate_factor"] = gf
    return out


class StubEncoder:
    qvec = None

    def encode(self, texts, normalize_embeddings=True):
        assert len(texts) == 1 and normalize_embeddings
        return np.asarray(StubEncoder.qvec, dtype=np.float32)[None, :].copy()


def write_world(d, world):
    """data/processed/* in the reference's layout (nlp/11_build_product_embeddings.py:82-90, nlp/12_product_prep.py:85-89)."""
    p = pathlib.Path(d) / "data" / "processed"
    p.mkdir(parents=True)
    np.save(p / "product_emb.npy", world["emb"])
    world["meta"].to_parquet(p / "product_emb_meta.parquet", index=False)
    if world["blob"] is not None:
        with open(p / "product_bm25.pkl", "wb") as f:
            pickle.dump(world["blob"], f, protocol=4)
    if world["reviews"] is not None:
        frame, E = world["reviews"]
        out = frame.copy()
        out["embedding"] = [np.asarray(e, dtype=np.float32) for e in E]
        out.to_parquet(p / "reviews_with_embeddings.parquet", index=False)


def run_case(ref, args_dict, full):
    """search(args) of the reference, its stdout swallowed; returns the JSON it wrote."""
    ns = argparse.Namespace(**args_dict, json_out="out/result.json")
    if full:
        ref.round = lambda x, nd=None: x          # module-global shadow of the builtin: rows before the 4-dp rounding
    try:
        with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
            warnings.simplefilter("ignore")       # nanmean of an all-NaN pool warns
            ref.search(ns)
    finally:
        if full:
            del ref.round
    return json.loads(pathlib.Path("out/result.json").read_text())


def search_fixtures(ref):
    ref._load_st_encoder = lambda: StubEncoder()
    ref._load_cross_encoder = lambda: W.FakeCrossEncoder()
    ref._load_rankbm25 = lambda: BM25OkapiOracle
    cases = []
    here = os.getcwd()
    current = None
    tmp = None
    try:
        for wname, cname, qi, over in W.plan():
            if wname != current:
                if tmp is not None:
                    os.chdir(here)
                    tmp.cleanup()
                tmp = tempfile.TemporaryDirectory()
                world = W.make_world(wname)
                write_world(tmp.name, world)
                os.chdir(tmp.name)
                current = wname
            cfg = dict(W.CONFIGS[cname])
            args = dict(query=W.QUERIES[qi], no_snippets=False, max_reviews_scan=1_000_000, **cfg)
            args.update(over)
            case = {"world": wname, "config": cname, "query": W.QUERIES[qi], "qvec_seed": 500 + qi,
                    "qvec_is_row7": wname == "ties", "args": {k: v for k, v in args.items() if k != "query"}}
            StubEncoder.qvec = W.qvec_of(case, world["emb"])
            case["results"] = run_case(ref, args, full=False)["results"]
            case["results_full"] = run_case(ref, args, full=True)["results"]
            assert [r["sku"] for r in case["results"]] == [r["sku"] for r in case["results_full"]]
            cases.append(case)
            print(f"  {len(cases):3d} {wname:9s} {cname:17s} q{qi} -> {len(case['results'])} rows, top {case['results'][0]['sku']}"
                  f" score {case['results'][0]['score']}", flush=True)
    finally:
        os.chdir(here)
        if tmp is not None:
            tmp.cleanup()
    return {"generator": "tests/golden/make_cli_golden.py", "source": "reference app/test.py search(args), loader hooks stubbed",
            "bm25_arithmetic": "oracle.bm25.BM25OkapiOracle (rank_bm25 absent: unpinned)", "numpy": np.__version__,
            "cases": cases}


def tokenizer_fixtures():
    prep = load_reference_module("nlp/12_product_prep.py", "ref_product_prep")     # module level: re, numpy, pandas
    rng = np.random.default_rng(7)
    words = ("I'm won't can't its it's a an THE Wireless USB-C 65w x y zz kid's o'clock rock'n'roll 3.5mm "
             "won't cannot their café naïve 100% q w-e r_t").split()
    texts = ["", "A", "I you he she we they", "Won't you BE my neighbour? It's 5 o'clock.", "x y z aa bb",
             "USB-C charger (65W), fast-charging; 2m cable!"] + \
            [" ".join(rng.choice(words, size=int(m))) for m in rng.integers(1, 30, 20)] + [" ".join(["tok%d" % i for i in range(5200)])]
    return {"generator": "tests/golden/make_cli_golden.py", "source": "reference nlp/12_product_prep.py tokenize (:75-78)",
            "cases": [{"text": t, "tokens": prep.tokenize(t)} for t in texts]}


def main():
    ref = load_reference_module("app/test.py", "ref_cli")
    (HERE / "cli_helpers.json").write_text(json.dumps(helper_fixtures(ref)))
    print("cli_helpers.json written")
    (HERE / "index_tokenizer.json").write_text(json.dumps(tokenizer_fixtures()))
    print("index_tokenizer.json written")
    fx = search_fixtures(ref)
    (HERE / "cli_search.json").write_text(json.dumps(fx))
    print("cli_search.json:", len(fx["cases"]), "cases")


if __name__ == "__main__":
    main()
