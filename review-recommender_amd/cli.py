"""Command-line search, the reference's ``app/test.py`` on the GPU path.

Same flags (app/test.py:345-361), same printed lines and JSON schema (app/test.py:312-342),
CLI flavour of the pipeline (pool floor 100, no trust factor, BM25 aligned by sku permutation).

    python -m review_recommender_amd.cli -q "wireless headphones" -k 10 --data-dir data/processed \
        [--qvec-npy query.npy] [--json-out out.json]

The query encoder and the cross-encoder run on the GPU (csrc/rr_ce.hip) from local Hugging Face
model directories (--emb-model-dir / --rerank-model-dir); without an encoder pass the query
embedding with --qvec-npy.  Reranking degrades to zeros like the reference does when the model
cannot be loaded (app/test.py:217-222).  Loading the two models by hub NAME through
sentence-transformers (a torch backend, what app/test.py:91-98 does) is not part of the product
path: it happens only with the explicit --allow-hub switch.
"""
from __future__ import annotations

import argparse
import json
import os
import pathlib
import sys

import numpy as np

EMB_MODEL = os.environ.get("EMB_MODEL", "BAAI/bge-small-en-v1.5")           # app/test.py:28
RERANK_MODEL = os.environ.get("RERANK_MODEL", "cross-encoder/ms-marco-MiniLM-L-6-v2")   # app/test.py:29


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="Search products with dense + (optional) BM25 + reranker + priors")
    ap.add_argument("-q", "--query", required=True, help="User query")
    ap.add_argument("-k", "--k", type=int, default=10, help="How many results to show")
    ap.add_argument("--rerank_k", type=int, default=50, help="Cross-encoder rerank pool size (0 to disable)")
    ap.add_argument("--no-snippets", action="store_true", help="Disable review snippets (faster)")
    ap.add_argument("--max-reviews-scan", type=int, default=1_000_000, help="Max reviews to load for snippets")
    ap.add_argument("--w-dense", type=float, default=0.55)
    ap.add_argument("--w-bm25", type=float, default=0.15)
    ap.add_argument("--w-rerank", type=float, default=0.15)
    ap.add_argument("--w-prior", type=float, default=0.10)
    ap.add_argument("--w-best", type=float, default=0.05)
    ap.add_argument("--prior-C", type=float, default=20.0, help="Bayesian prior strength")
    ap.add_argument("--gate-penalty", type=float, default=0.5, help="Penalty per missing attribute group (0.1-1.0)")
    ap.add_argument("--json-out", type=str, default="", help="Optional path to save results JSON")
    # additions of this build
    ap.add_argument("--data-dir", type=str, default="data/processed", help="directory with the three artefacts")
    ap.add_argument("--qvec-npy", type=str, default="", help=".npy with the query embedding (offline use)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--emb-model-dir", type=str, default=os.environ.get("EMB_MODEL_DIR", ""),
                    help="local Hugging Face directory of the query encoder (model.safetensors | pytorch_model.bin, vocab.txt): "
                         "runs on the GPU (csrc/rr_ce.hip); nothing is fetched")
    ap.add_argument("--rerank-model-dir", type=str, default=os.environ.get("RERANK_MODEL_DIR", ""),
                    help="local Hugging Face directory of the cross-encoder reranker, likewise")
    ap.add_argument("--ce-precision", choices=["fp32", "bf16"], default="fp32",
                    help="arithmetic of the two GPU encoders: fp32 = the reference's (logits / embeddings within 1e-5 of "
                         "transformers; ~6x the time of bf16), bf16 = the fast path (2.5e-2 on logits)")
    ap.add_argument("--allow-hub", action="store_true",
                    help="opt in to sentence-transformers by model name (EMB_MODEL / RERANK_MODEL, app/test.py:28-29) for "
                         "whatever model no local directory was given; off by default: the product path is the GPU one")
    return ap.parse_args(argv)


def _load_encoders(args=None):
    """Query encoder and reranker from local model directories, on the GPU (--emb-model-dir / --rerank-model-dir).
    Only with --allow-hub: sentence-transformers by model name as the reference does (app/test.py:91-104) for a model
    no directory was given for.  A reranker that cannot be loaded degrades to zeros with the reference's warning."""
    enc = ce = None
    if args is not None and (args.emb_model_dir or args.rerank_model_dir):
        from .cross_encoder import CrossEncoder, QueryEncoder
        if args.emb_model_dir:
            enc = QueryEncoder.from_pretrained_dir(args.emb_model_dir, device=args.device, precision=args.ce_precision)
        if args.rerank_model_dir:
            try:
                ce = CrossEncoder.from_pretrained_dir(args.rerank_model_dir, device=args.device, precision=args.ce_precision)
            except Exception as e:   # app/test.py:220-222
                print(f"[warn] cross-encoder load failed: {e}; skipping reranker.", flush=True)
    if args is None or not args.allow_hub:
        return enc, ce
    try:
        from sentence_transformers import CrossEncoder, SentenceTransformer
        enc = enc or SentenceTransformer(EMB_MODEL)
        try:
            ce = ce or CrossEncoder(RERANK_MODEL)
        except Exception as e:   # app/test.py:220-222
            print(f"[warn] cross-encoder load failed: {e}; skipping reranker.", flush=True)
    except Exception:
        pass
    return enc, ce


def main(argv=None) -> int:
    args = parse_args(argv)
    from .artifacts import ArtifactError
    from .engine import SearchEngine, cli_rows
    qvec = np.load(args.qvec_npy).astype(np.float32).reshape(-1) if args.qvec_npy else None
    enc, ce = _load_encoders(args) if (qvec is None or args.rerank_model_dir) else (None, None)
    if qvec is None and enc is None:
        raise SystemExit(f"[ERR] loading/encoding with {EMB_MODEL} failed: no query encoder; pass --emb-model-dir or "
                         "--qvec-npy (or opt in to the hub loader with --allow-hub)")          # app/test.py:234-235
    try:
        engine = SearchEngine.from_artifacts(args.data_dir, encoder=enc, cross_encoder=ce, flavour="cli",
                                             device=args.device)
    except ArtifactError as e:
        raise SystemExit(f"[ERR] {e}")
    # snippets are scored whenever the review file exists and --no-snippets is absent (app/test.py:271-276)
    frame, snips, _ = engine.run_search(args.query, args.k, args.rerank_k, args.w_dense, args.w_bm25, args.w_rerank,
                                        args.w_prior, args.w_best, args.prior_C, not args.no_snippets,
                                        args.max_reviews_scan, 8, args.gate_penalty, qvec=qvec)
    rows = cli_rows(frame, snips)
    print("\nTop results:")
    for i, r in enumerate(rows, 1):
        print(f"[{i}] {r['sku']}  score={r['score']}  (dense={r['dense']} bm25={r['bm25']} rerank={r['rerank']} "
              f"prior={r['prior']} best={r['bestrev']})  reviews={r['n_reviews']} avg={r['avg_stars']}")
        if r["snippet"]:
            print("    ", r["snippet"])                        # app/test.py:335-336
    if args.json_out:
        pathlib.Path(args.json_out).parent.mkdir(parents=True, exist_ok=True)
        with open(args.json_out, "w") as f:
            json.dump({"query": args.query, "results": rows}, f, ensure_ascii=False, indent=2)
        print(f"\n[ok] wrote {args.json_out}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
