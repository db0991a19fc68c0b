"""Oracle: dense cosine top-k (numpy restatement of utils.py:111-124).

TEST INFRASTRUCTURE -- never imported by the product path.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def cosine_similarity_search(query_vector: np.ndarray, embeddings_matrix: np.ndarray,
                             top_k: int) -> Tuple[np.ndarray, np.ndarray]:
    """utils.py:111-124 (= _cosine_pool app/app_product_search.py:192-195,
    cosine_search app/test.py:125-132): BLAS matvec, argpartition, argsort."""
    sims = embeddings_matrix @ query_vector
    if top_k >= len(sims):
        top_k = len(sims)
    part = np.argpartition(-sims, top_k - 1)[:top_k]
    order = part[np.argsort(-sims[part])]
    return order, sims[order]


def topk_reference_order(sims: np.ndarray, top_k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Deterministic form of the same selection: (score desc, row asc).

    The reference's tie order is unspecified (np.argsort quicksort,
    utils.py:122); the build defines it as ascending row index.  Used by the
    parity helpers to compare tie groups.
    """
    top_k = min(top_k, len(sims))
    order = np.lexsort((np.arange(len(sims)), -sims.astype(np.float64)))[:top_k]
    return order.astype(np.int64), sims[order]


def sims_float64(embeddings_matrix: np.ndarray, query_vector: np.ndarray) -> np.ndarray:
    """Float64 dot products: the rounding-free yardstick for tolerance checks."""
    q64 = query_vector.astype(np.float64)
    step = 1_000_000                 # (rows are independent: chunking bounds the float64 copy at 10M rows, same values)
    if embeddings_matrix.shape[0] <= step:
        return embeddings_matrix.astype(np.float64) @ q64
    return np.concatenate([embeddings_matrix[s:s + step].astype(np.float64) @ q64
                           for s in range(0, embeddings_matrix.shape[0], step)])


def round_to_bf16(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest-even bfloat16, returned widened back to float32 (SURVEY section 8d:
    the bf16 configs' oracle is the fp32 matvec over the matrix rounded ONCE to bf16)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)) << np.uint32(16)
    return r.view(np.float32)
