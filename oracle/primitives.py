"""Oracle: scoring primitives (numpy restatement of the reference's utils.py).

TEST INFRASTRUCTURE -- never imported by the product path.

Every function names the reference lines it follows.  The arithmetic is kept
as the same numpy expression the reference evaluates, so dtype promotion and
rounding are decided by numpy exactly as they are for the reference.
"""
from __future__ import annotations

import math
import re
from typing import List, Sequence, Set, Tuple

import numpy as np

# utils.py:11-12 (same pattern / list at app/app_product_search.py:152-153)
QUERY_TOKEN = re.compile(r"[a-z0-9]+(?:'[a-z0-9]+)?")
QUERY_STOP = frozenset(
    "a an the and or of for to in on with is are it this that".split()
)

# utils.py:15-24
SYNONYM_GROUPS = {
    "sock": {"sock", "socks"},
    "headphone": {"headphone", "headphones", "earphone", "earphones",
                  "earbud", "earbuds", "headset"},
    "keyboard": {"keyboard", "keyboards"},
    "wireless": {"wireless", "bluetooth"},
    "noise": {"noise cancelling", "noise-canceling", "noise canceling", "anc"},
    "cat": {"cat", "cats", "kitten", "kittens", "kitty"},
    "dog": {"dog", "dogs", "puppy", "puppies"},
    "design": {"design", "pattern", "print", "graphic", "artwork", "motif",
               "theme"},
}

# utils.py:26-38 (insertion order matters: groups are emitted in this order)
COLOR_GROUPS = {
    "yellow": {"yellow", "mustard", "lemon", "gold", "golden"},
    "red": {"red", "scarlet", "crimson", "maroon"},
    "blue": {"blue", "navy", "cobalt", "azure"},
    "green": {"green", "emerald", "olive"},
    "black": {"black"},
    "white": {"white", "ivory"},
    "pink": {"pink", "rose"},
    "purple": {"purple", "violet", "lavender"},
    "orange": {"orange", "amber"},
    "brown": {"brown", "tan", "beige", "khaki"},
    "gray": {"gray", "grey", "charcoal", "slate"},
}


def l2_normalize(x: np.ndarray, axis: int = 1, eps: float = 1e-12) -> np.ndarray:
    """utils.py:40-44: x / max(||x||_2, eps) along ``axis`` (dtype follows x)."""
    norms = np.maximum(np.linalg.norm(x, axis=axis, keepdims=True), eps)
    return x / norms


def minmax_normalize(x: np.ndarray, empty_passthrough: bool = False) -> np.ndarray:
    """utils.py:46-55 / app/app_product_search.py:182-187.

    ``empty_passthrough=True`` gives the CLI copy (app/test.py:114-119), which
    returns an empty input unchanged instead of casting it to float32.
    """
    if x.size == 0:
        return x if empty_passthrough else x.astype(np.float32)
    lo = float(np.min(x))
    hi = float(np.max(x))
    if (not math.isfinite(lo)) or (not math.isfinite(hi)) or hi - lo < 1e-12:
        return np.zeros_like(x, dtype=np.float32)
    # array - pyfloat and array / pyfloat keep the array dtype (numpy 2 weak
    # scalars): float32 input is scaled in float32, float64 input in float64.
    return ((x - lo) / (hi - lo + 1e-12)).astype(np.float32)


def tokenize_query(query: str) -> List[str]:
    """utils.py:57-60: lowercase, regex tokens, minus the 16 stop words."""
    return [t for t in QUERY_TOKEN.findall(query.lower()) if t not in QUERY_STOP]


def build_gate_groups(query: str) -> List[Set[str]]:
    """utils.py:62-86: colour groups by substring, then per-token groups; cap 6."""
    lowered = query.lower()
    found: List[Set[str]] = []
    for members in COLOR_GROUPS.values():
        if any(m in lowered for m in members):
            found.append(members)
    for tok in tokenize_query(query):
        if tok in SYNONYM_GROUPS:
            found.append(SYNONYM_GROUPS[tok])
        elif len(tok) >= 4:
            found.append({tok})
    dedup: List[Set[str]] = []
    for g in found:
        if g not in dedup:
            dedup.append(g)
    return dedup[:6]


def calculate_gate_factor(text: str, groups: Sequence[Set[str]],
                          penalty: float = 0.5) -> Tuple[float, int, int]:
    """utils.py:88-101: penalty ** (#groups with no member as substring of text)."""
    lowered = text.lower()
    hits = 0
    factor = 1.0
    for g in groups:
        if any(m in lowered for m in g):
            hits += 1
        else:
            factor *= penalty
    return factor, hits, len(groups)


def bayesian_prior(avg_ratings: np.ndarray, review_counts: np.ndarray,
                   prior_strength: float = 20.0,
                   global_mean: float | None = None) -> np.ndarray:
    """utils.py:103-109 / app/app_product_search.py:197-199."""
    g = float(np.nanmean(avg_ratings)) if global_mean is None else float(global_mean)
    return ((avg_ratings * review_counts) + (g * prior_strength)) / (
        review_counts + prior_strength + 1e-9)


def trust_score_from_reviews(review_counts: np.ndarray, min_reviews: int = 8,
                             saturation: int = 50) -> np.ndarray:
    """utils.py:126-133 / app/app_product_search.py:238-242 (app calls sat=80)."""
    ramp = np.clip(review_counts / max(min_reviews, 1), 0, 1)
    sat = np.minimum(1.0, np.log1p(review_counts) / np.log1p(max(saturation, 1)))
    return (0.6 * ramp + 0.4 * sat).astype(np.float32)
