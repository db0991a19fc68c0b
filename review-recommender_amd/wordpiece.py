"""BERT WordPiece tokenisation on the host (text -> token ids for csrc/rr_ce.hip).

What `CrossEncoder.predict` / `SentenceTransformer.encode` do before the model runs
(app/app_product_search.py:250-251,277-278; app/test.py:223-225,232): the models' `BertTokenizer`
(lower-casing "uncased" vocabularies for both ms-marco-MiniLM-L-6-v2 and bge-small-en-v1.5) turns a text
pair into `[CLS] a [SEP] b [SEP]` with token types 0 / 1, truncated `longest_first` to 512 tokens.
Written from the published algorithm (BasicTokenizer + greedy longest-match WordPiece); checked against
`transformers.BertTokenizer` on a synthetic vocabulary (tests/golden/k5_tokenizer.json).  The vocabulary is a
local `vocab.txt`; nothing is fetched.
"""
from __future__ import annotations

import unicodedata
from typing import Dict, List, Optional, Tuple

import numpy as np


def _is_whitespace(ch: str) -> bool:
    if ch in (" ", "\t", "\n", "\r"):
        return True
    return unicodedata.category(ch) == "Zs"


def _is_control(ch: str) -> bool:
    if ch in ("\t", "\n", "\r"):
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch: str) -> bool:
    cp = ord(ch)
    if (33 <= cp <= 47) or (58 <= cp <= 64) or (91 <= cp <= 96) or (123 <= cp <= 126):
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return ((0x4E00 <= cp <= 0x9FFF) or (0x3400 <= cp <= 0x4DBF) or (0x20000 <= cp <= 0x2A6DF)
            or (0x2A700 <= cp <= 0x2B73F) or (0x2B740 <= cp <= 0x2B81F) or (0x2B820 <= cp <= 0x2CEAF)
            or (0xF900 <= cp <= 0xFAFF) or (0x2F800 <= cp <= 0x2FA1F))


def basic_tokenize(text: str, do_lower_case: bool = True) -> List[str]:
    out = []
    for ch in text:                                    # clean + isolate CJK characters
        cp = ord(ch)
        if cp == 0 or cp == 0xFFFD or _is_control(ch):
            continue
        if _is_whitespace(ch):
            out.append(" ")
        elif _is_cjk(cp):
            out.extend((" ", ch, " "))
        else:
            out.append(ch)
    text = unicodedata.normalize("NFC", "".join(out))
    tokens: List[str] = []
    for tok in text.split():
        if do_lower_case:
            tok = tok.lower()
            tok = "".join(c for c in unicodedata.normalize("NFD", tok) if unicodedata.category(c) != "Mn")
        cur: List[str] = []
        for ch in tok:                                 # every punctuation character is its own token
            if _is_punctuation(ch):
                if cur:
                    tokens.append("".join(cur))
                    cur = []
                tokens.append(ch)
            else:
                cur.append(ch)
        if cur:
            tokens.append("".join(cur))
    return tokens


class WordPieceTokenizer:
    def __init__(self, vocab: Dict[str, int], do_lower_case: bool = True, unk: str = "[UNK]", cls: str = "[CLS]",
                 sep: str = "[SEP]", pad: str = "[PAD]", max_chars_per_word: int = 100, max_length: int = 512):
        self.vocab = vocab
        self.do_lower_case = do_lower_case
        for t in (unk, cls, sep):
            if t not in vocab:
                raise ValueError(f"vocabulary lacks the special token {t}")
        self.unk_id, self.cls_id, self.sep_id = vocab[unk], vocab[cls], vocab[sep]
        self.pad_id = vocab.get(pad, 0)
        self.max_chars_per_word, self.max_length = max_chars_per_word, max_length
        self._cache: Dict[str, List[int]] = {}

    @classmethod
    def from_vocab_file(cls, path, **kw) -> "WordPieceTokenizer":
        vocab: Dict[str, int] = {}
        with open(path, encoding="utf-8") as f:
            for i, line in enumerate(f):
                vocab[line.rstrip("\n")] = i
        return cls(vocab, **kw)

    def _word(self, word: str) -> List[int]:
        hit = self._cache.get(word)
        if hit is not None:
            return hit
        if len(word) > self.max_chars_per_word:
            ids = [self.unk_id]
        else:
            ids, start, n = [], 0, len(word)
            while start < n:
                end, cur = n, None
                while start < end:
                    piece = word[start:end] if start == 0 else "##" + word[start:end]
                    cur = self.vocab.get(piece)
                    if cur is not None:
                        break
                    end -= 1
                if cur is None:
                    ids = [self.unk_id]
                    break
                ids.append(cur)
                start = end
        if len(self._cache) < 1_000_000:
            self._cache[word] = ids
        return ids

    def text_ids(self, text: str) -> List[int]:
        out: List[int] = []
        for w in basic_tokenize(text, self.do_lower_case):
            out.extend(self._word(w))
        return out

    def encode_pair(self, a: str, b: Optional[str] = None, max_length: Optional[int] = None
                    ) -> Tuple[np.ndarray, np.ndarray]:
        """([CLS] a [SEP] (b [SEP])) token ids and type ids, truncated `longest_first` to max_length."""
        L = max_length or self.max_length
        ia = self.text_ids(a)
        if b is None:
            ia = ia[:max(L - 2, 0)]
            ids = [self.cls_id] + ia + [self.sep_id]
            return np.asarray(ids, dtype=np.int32), np.zeros(len(ids), dtype=np.int32)
        ib = self.text_ids(b)
        avail = max(L - 3, 0)
        if len(ia) + len(ib) > avail:
            # `longest_first` as the Rust `tokenizers` library (what AutoTokenizer gives sentence-transformers)
            # implements it: only the longer side is cut while the shorter fits in what is left; when both must
            # be cut the shorter side keeps avail // 2 tokens and the longer the rest (b on a tie).
            n1, n2, swap = len(ia), len(ib), False
            if n1 > n2:
                n1, n2, swap = n2, n1, True
            n2 = n1 if n1 > avail else max(n1, avail - n1)
            if n1 + n2 > avail:
                n1 = avail // 2
                n2 = n1 + avail % 2
            if swap:
                n1, n2 = n2, n1
            ia, ib = ia[:n1], ib[:n2]
        ids = [self.cls_id] + ia + [self.sep_id] + ib + [self.sep_id]
        typ = [0] * (len(ia) + 2) + [1] * (len(ib) + 1)
        return np.asarray(ids, dtype=np.int32), np.asarray(typ, dtype=np.int32)
