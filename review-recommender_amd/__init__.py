"""review-recommender_amd: MI355X-native hybrid retrieval hot path
(dense cosine top-k + BM25 at the pool + fused priors / blend / top-k), a drop-in
for the search path of Ntropy86/review-recommender.  Hand-written HIP (gfx950)
behind a C ABI (include/rr_hip.h); see DESIGN.md and INTEGRATION.md.

Importing the package does not load the HIP library; the first call does, and
raises if ``librr_hip.so`` has not been built (there is no CPU fallback).
"""
from .text import (build_gate_groups, calculate_gate_factor, tokenize_document,  # noqa: F401
                   tokenize_query)

__all__ = ["tokenize_query", "tokenize_document", "build_gate_groups", "calculate_gate_factor",
           "ProductIndex", "BM25Corpus", "BM25Index", "BM25Okapi", "SearchEngine",
           "HybridSearcher", "FusionWeights", "cosine_similarity_search"]

_LAZY = {
    "ProductIndex": "index", "BM25Corpus": "bm25", "BM25Index": "bm25", "BM25Okapi": "bm25",
    "SearchEngine": "engine", "HybridSearcher": "engine", "FusionWeights": "engine",
    "cosine_similarity_search": "engine", "ShardedSearcher": "sharded",
}


def __getattr__(name):
    mod = _LAZY.get(name)
    if mod is None:
        raise AttributeError(name)
    import importlib
    return getattr(importlib.import_module(f"{__name__}.{mod}"), name)
