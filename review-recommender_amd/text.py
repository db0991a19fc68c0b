"""Host-side query text handling of the hot path (a4, a10, a11).

These are string operations on one query and <= pool short texts; they stay on
the host (the reference does them in Python too) and feed the HIP kernels with
token ids and one gate factor per candidate.

  tokenize_query        utils.py:57-60   (= _tokenize app/app_product_search.py:189-190)
  tokenize_document     nlp/12_product_prep.py:75-78  (index-time tokenizer)
  build_gate_groups     utils.py:62-86   (= _build_gate_groups app/...:211-227)
  calculate_gate_factor utils.py:88-101  (= _gate_factor app/...:228-236; CLI returns the float)
"""
from __future__ import annotations

import re
from typing import List, Sequence, Set, Tuple

_TOKEN = re.compile(r"[a-z0-9]+(?:'[a-z0-9]+)?")

# query-time stop words (utils.py:12)
QUERY_STOP_WORDS = frozenset(
    ["a", "an", "the", "and", "or", "of", "for", "to", "in", "on", "with", "is",
     "are", "it", "this", "that"])

# index-time stop words (nlp/12_product_prep.py:43-49) -- a superset, plus len(t) > 1
INDEX_STOP_WORDS = frozenset(
    ["a", "an", "and", "the", "is", "are", "am", "be", "been", "to", "for", "of", "in",
     "on", "at", "by", "it", "its", "this", "that", "with", "from", "as", "or", "if",
     "but", "than", "then", "so", "i", "you", "he", "she", "we", "they", "my", "your",
     "our", "their", "me", "him", "her", "us", "them", "was", "were", "will", "would",
     "should", "could", "may", "might", "can", "cannot", "cant", "won't"])
INDEX_TOKEN_CAP = 5000

# attribute groups used by the gate (utils.py:15-38); dict order is emission order
SYNONYMS = {
    "sock": {"sock", "socks"},
    "headphone": {"headphone", "headphones", "earphone", "earphones", "earbud",
                  "earbuds", "headset"},
    "keyboard": {"keyboard", "keyboards"},
    "wireless": {"wireless", "bluetooth"},
    "noise": {"noise cancelling", "noise-canceling", "noise canceling", "anc"},
    "cat": {"cat", "cats", "kitten", "kittens", "kitty"},
    "dog": {"dog", "dogs", "puppy", "puppies"},
    "design": {"design", "pattern", "print", "graphic", "artwork", "motif", "theme"},
}
COLORS = {
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
MAX_GATE_GROUPS = 6
MIN_KEYWORD_LEN = 4


def tokenize_query(query: str) -> List[str]:
    return [tok for tok in _TOKEN.findall(query.lower()) if tok not in QUERY_STOP_WORDS]


def tokenize_document(text: str) -> List[str]:
    toks = [t for t in _TOKEN.findall(text.lower())
            if t not in INDEX_STOP_WORDS and len(t) > 1]
    return toks[:INDEX_TOKEN_CAP]


def build_gate_groups(query: str) -> List[Set[str]]:
    q = query.lower()
    groups: List[Set[str]] = [syn for syn in COLORS.values() if any(w in q for w in syn)]
    for tok in tokenize_query(query):
        if tok in SYNONYMS:
            groups.append(SYNONYMS[tok])
        elif len(tok) >= MIN_KEYWORD_LEN:
            groups.append({tok})
    unique: List[Set[str]] = []
    for g in groups:
        if g not in unique:
            unique.append(g)
    return unique[:MAX_GATE_GROUPS]


def calculate_gate_factor(text: str, groups: Sequence[Set[str]],
                          penalty: float = 0.5) -> Tuple[float, int, int]:
    body = text.lower()
    matched = sum(1 for g in groups if any(s in body for s in g))
    factor = 1.0
    for _ in range(len(groups) - matched):
        factor *= penalty   # same left-to-right products as the reference's loop
    return factor, matched, len(groups)
