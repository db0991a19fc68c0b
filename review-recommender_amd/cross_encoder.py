"""Cross-encoder reranker and query encoder on the GPU (K5).

Drop-ins for the two sentence-transformers objects the reference holds:

  CrossEncoder(RERANK_MODEL).predict(pairs, batch_size=64, show_progress_bar=False) -> (len(pairs),) float32
      app/app_product_search.py:71-86,277-278; app/test.py:96-104,223-225
  SentenceTransformer(EMB_MODEL).encode([query], normalize_embeddings=True) -> (n, 384) float32
      app/app_product_search.py:53-69,250-251; app/test.py:91-94,232

Both run csrc/rr_ce.hip through the C ABI (rr_ce_create / rr_ce_forward_dev), in one of two precisions: "fp32" (default:
the reference's arithmetic: every fp32 operand as two fp16 numbers, three matrix-core products per fp32 product, within 1e-5
of `transformers`) or "bf16"
(the fast path: bf16 MFMA GEMMs + attention, fp32 residual stream).  Weights come from a local state dict (Hugging Face BERT names: a
`model.safetensors` / `pytorch_model.bin` directory, or a dict of arrays); text is tokenised by
wordpiece.WordPieceTokenizer from a local vocab.txt.  Without a vocabulary on disk the pre-tokenised entry
points (`predict_ids`, `encode_ids`) take token-id sequences directly.  Nothing is ever fetched.

The reference leaves one thing undeterminable offline: whether predict applies Identity or Sigmoid to the
single logit (it depends on the hub model's config).  `activation` is therefore a parameter,
default raw logits; min-max follows in run_search either way (app/app_product_search.py:279).
"""
from __future__ import annotations

import ctypes as C
import json
import pathlib
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .wordpiece import WordPieceTokenizer

OUT_LOGITS, OUT_CLS, OUT_HIDDEN = 0, 1, 2       # RR_CE_OUT_*
PRECISIONS = {"bf16": 0, "fp32": 1}              # RR_CE_PRECISION_*
HIDDEN, HEADS, FFN = 384, 12, 1536

_LAYER_KEYS = ("attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight",
               "attention.self.key.bias", "attention.self.value.weight", "attention.self.value.bias",
               "attention.output.dense.weight", "attention.output.dense.bias", "attention.output.LayerNorm.weight",
               "attention.output.LayerNorm.bias", "intermediate.dense.weight", "intermediate.dense.bias",
               "output.dense.weight", "output.dense.bias", "output.LayerNorm.weight", "output.LayerNorm.bias")


def _np(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().float().numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


def _find_prefix(sd: Dict) -> str:
    for p in ("bert.", "model.", ""):
        if p + "embeddings.word_embeddings.weight" in sd:
            return p
    raise ValueError("state dict holds no '[bert.]embeddings.word_embeddings.weight': not a BERT checkpoint")


class BertEncoderGPU:
    """A BERT encoder (+ optional sequence-classification head) resident on one GPU."""

    def __init__(self, state_dict: Dict, *, device: int = 0, ln_eps: float = 1e-12, with_head: bool = True,
                 max_tokens_per_call: int = 131_072, precision: str = "fp32"):
        """``precision``: "fp32" = the reference's arithmetic (fp32 weights and activations, products on the fp16 matrix
        cores as exact as an fp32 multiply-add chain: logits / embeddings within 1e-5 of `transformers`; a value beyond
        fp16's range switches the handle to the wide-range kernels, see `out_of_range`), "bf16" = the fast path (bf16
        operands, fp32 accumulation: 2.5e-2 on O(1) logits, ~2.7x the throughput).  include/rr_hip.h: RR_CE_PRECISION_*."""
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
        self.precision = precision
        sd = state_dict
        p = _find_prefix(sd)
        word = _np(sd[p + "embeddings.word_embeddings.weight"])
        pos = _np(sd[p + "embeddings.position_embeddings.weight"])
        typ = _np(sd[p + "embeddings.token_type_embeddings.weight"])
        n_layers = 0
        while f"{p}encoder.layer.{n_layers}.attention.self.query.weight" in sd:
            n_layers += 1
        if n_layers == 0:
            raise ValueError("state dict holds no encoder layers")
        has_head = with_head and "classifier.weight" in sd and (p + "pooler.dense.weight") in sd
        tensors: List[np.ndarray] = [word, pos, typ, _np(sd[p + "embeddings.LayerNorm.weight"]),
                                     _np(sd[p + "embeddings.LayerNorm.bias"])]
        for l in range(n_layers):
            tensors += [_np(sd[f"{p}encoder.layer.{l}.{k}"]) for k in _LAYER_KEYS]
        n_labels = 0
        if has_head:
            cw = _np(sd["classifier.weight"])
            n_labels = cw.shape[0]
            tensors += [_np(sd[p + "pooler.dense.weight"]), _np(sd[p + "pooler.dense.bias"]), cw,
                        _np(sd["classifier.bias"])]
        if word.shape[1] != HIDDEN or tensors[5 + 10].shape != (FFN, HIDDEN):
            raise ValueError(f"the kernels are built for hidden {HIDDEN} / FFN {FFN} (MiniLM-L6, bge-small); got hidden "
                             f"{word.shape[1]}, FFN {tensors[5 + 10].shape[0]}")
        self.n_layers, self.n_labels, self.vocab, self.max_pos = n_layers, n_labels, word.shape[0], pos.shape[0]
        self.type_vocab, self.device = typ.shape[0], device
        self.max_tokens_per_call = int(max_tokens_per_call)
        cfg = _lib.CEConfig(HIDDEN, n_layers, HEADS, FFN, self.vocab, self.max_pos, self.type_vocab, n_labels, ln_eps,
                            PRECISIONS[precision])
        ptrs = (C.c_void_p * len(tensors))(*[t.ctypes.data for t in tensors])
        h = C.c_void_p()
        _lib.check(_lib.load().rr_ce_create(device, C.byref(cfg), ptrs, len(tensors), C.byref(h)), "rr_ce_create")
        self._h = h
        import torch
        if not torch.cuda.is_available():
            raise _lib.HipLibraryError("no GPU visible: the encoder runs on the device only")
        self._torch = torch
        self._dev = torch.device("cuda", device)

    @property
    def handle(self):
        return self._h

    # ------------------------------------------------------------------ forward over token ids
    def _check(self, seqs):
        for ids, typ in seqs:
            if len(ids) < 1 or len(ids) > self.max_pos:
                raise ValueError(f"a sequence has {len(ids)} tokens; the model takes 1..{self.max_pos}")
            if len(typ) != len(ids):
                raise ValueError("token ids and token type ids differ in length")

    def forward_ids(self, seqs: Sequence[Tuple[Sequence[int], Sequence[int]]], mode: int = OUT_LOGITS) -> np.ndarray:
        """seqs: (token_ids, type_ids) per sequence (unpadded).  Returns logits (n, n_labels), CLS states (n, 384)
        or the hidden states of all tokens (sum of lengths, 384) by `mode`.  Sequences are packed back to back and
        processed in chunks of at most `max_tokens_per_call` tokens; inside a chunk they are sorted by length only for
        the launch bound (the result is independent of batch composition: there is no padding to attend to)."""
        torch = self._torch
        self._check(seqs)
        n = len(seqs)
        width = {OUT_LOGITS: self.n_labels, OUT_CLS: HIDDEN, OUT_HIDDEN: HIDDEN}[mode]
        lens = np.array([len(s[0]) for s in seqs], dtype=np.int64)
        if mode == OUT_LOGITS and self.n_labels == 0:
            raise ValueError("this encoder was loaded without a classification head")
        out = np.empty((int(lens.sum()) if mode == OUT_HIDDEN else n, width), dtype=np.float32)
        lib = _lib.load()
        start = 0
        tok_done = 0
        while start < n:
            end, tot = start, 0
            while end < n and (end == start or tot + lens[end] <= self.max_tokens_per_call):
                tot += int(lens[end])
                end += 1
            part = seqs[start:end]
            cu = np.zeros(len(part) + 1, dtype=np.int32)
            np.cumsum(lens[start:end], out=cu[1:])
            ids = np.concatenate([np.asarray(s[0], dtype=np.int32) for s in part])
            typ = np.concatenate([np.asarray(s[1], dtype=np.int32) for s in part])
            if ids.min() < 0 or ids.max() >= self.vocab or typ.min() < 0 or typ.max() >= self.type_vocab:
                raise ValueError("token id or token type id outside the embedding tables")
            pos = (np.arange(tot, dtype=np.int32) - np.repeat(cu[:-1], lens[start:end])).astype(np.int32)
            packed = np.concatenate([ids, typ, pos, cu]).astype(np.int32)
            with torch.cuda.device(self._dev):
                d = torch.from_numpy(packed).to(self._dev)
                rows = tot if mode == OUT_HIDDEN else len(part)
                d_out = torch.empty((rows, width), dtype=torch.float32, device=self._dev)
                base = d.data_ptr()
                _lib.check(lib.rr_ce_forward_dev(
                    self._h, C.c_void_p(base), C.c_void_p(base + 4 * tot), C.c_void_p(base + 8 * tot),
                    C.c_void_p(base + 12 * tot), len(part), tot, int(lens[start:end].max()), mode,
                    C.c_void_p(d_out.data_ptr()), C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)),
                    "rr_ce_forward_dev")
                res = d_out.cpu().numpy()
                if self.precision == "fp32" and self.out_of_range():
                    # an activation beyond fp16's 65504 (csrc/rr_ce_h2.hip): this handle continues on the wide-range kernels
                    self.set_wide_range(True)
                    _lib.check(lib.rr_ce_forward_dev(
                        self._h, C.c_void_p(base), C.c_void_p(base + 4 * tot), C.c_void_p(base + 8 * tot),
                        C.c_void_p(base + 12 * tot), len(part), tot, int(lens[start:end].max()), mode,
                        C.c_void_p(d_out.data_ptr()), C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)),
                        "rr_ce_forward_dev")
                    res = d_out.cpu().numpy()
            if mode == OUT_HIDDEN:
                out[tok_done:tok_done + tot] = res
            else:
                out[start:end] = res
            tok_done += tot
            start = end
        return out

    def forward_packed_dev(self, tok, typ, pos, cu, n_seqs: int, max_len: int, mode: int = OUT_LOGITS):
        """Forward over sequences that are already packed ON THE DEVICE (int32 torch tensors: token ids, type ids,
        position ids per token; cu_seqlens per sequence): returns a device tensor, no host hop.  Asynchronous on
        torch's current stream.  Used where the pairs are assembled on the GPU (bench.py's rerank mode)."""
        torch = self._torch
        n_tokens = int(tok.numel())
        width = {OUT_LOGITS: self.n_labels, OUT_CLS: HIDDEN, OUT_HIDDEN: HIDDEN}[mode]
        out = torch.empty((n_tokens if mode == OUT_HIDDEN else n_seqs, width), dtype=torch.float32, device=self._dev)
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(_lib.load().rr_ce_forward_dev(
            self._h, p(tok), p(typ), p(pos), p(cu), int(n_seqs), n_tokens, int(max_len), mode, p(out),
            C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)), "rr_ce_forward_dev")
        return out

    def out_of_range(self) -> bool:
        """True when the LAST forward pass (waited for here) met a value beyond fp16's range: its logits / CLS rows are NaN
        (include/rr_hip.h: rr_ce_range_status).  `forward_ids` checks and reruns by itself; a caller of
        `forward_packed_dev` checks after it has synchronised, calls `set_wide_range(True)` and submits the batch again."""
        flag = C.c_int32()
        _lib.check(_lib.load().rr_ce_range_status(self._h, C.byref(flag)), "rr_ce_range_status")
        return bool(flag.value)

    def set_wide_range(self, on: bool = True) -> None:
        """fp32 precision only: three bf16 terms per operand and six products (any fp32 range) instead of two fp16 terms
        and three."""
        if on and not getattr(self, "_warned_wide", False):
            import warnings
            warnings.warn("encoder activations exceed the fp16 range: this handle switches to the bf16 three-term kernels")
            self._warned_wide = True
        _lib.check(_lib.load().rr_ce_set_wide_range(self._h, 1 if on else 0), "rr_ce_set_wide_range")

    def last_forward_ms(self) -> float:
        ms = C.c_float()
        _lib.check(_lib.load().rr_ce_last_forward_ms(self._h, C.byref(ms)), "rr_ce_last_forward_ms")
        return ms.value

    def close(self) -> None:
        if getattr(self, "_h", None):
            _lib.load().rr_ce_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_state_dict_dir(path) -> Tuple[Dict, Optional[pathlib.Path], Dict]:
    """(state dict, vocab.txt path or None, config dict) from a local Hugging Face model directory
    (model.safetensors or pytorch_model.bin; sentence-transformers keeps the transformer in the directory root
    or in `0_Transformer/`)."""
    d = pathlib.Path(path)
    for sub in ("", "0_Transformer"):
        dd = d / sub
        if (dd / "model.safetensors").exists():
            from safetensors.numpy import load_file
            sd = load_file(str(dd / "model.safetensors"))
            break
        if (dd / "pytorch_model.bin").exists():
            import torch
            sd = torch.load(str(dd / "pytorch_model.bin"), map_location="cpu", weights_only=True)
            break
    else:
        raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {d}")
    vocab = dd / "vocab.txt"
    cfg = json.loads((dd / "config.json").read_text()) if (dd / "config.json").exists() else {}
    return sd, (vocab if vocab.exists() else None), cfg


class CrossEncoder:
    """`sentence_transformers.CrossEncoder` on the calls the reference makes: `predict(pairs, batch_size=64,
    show_progress_bar=False)` -> float32 (len(pairs),) for a single-label model."""

    def __init__(self, state_dict: Dict, tokenizer: Optional[WordPieceTokenizer] = None, *, device: int = 0,
                 max_length: int = 512, activation: Optional[str] = None, ln_eps: float = 1e-12, precision: str = "fp32"):
        """``precision`` "fp32" (default: the reference reranks in fp32, app/app_product_search.py:277-278) or "bf16"
        (fast path, logits within 2.5e-2): see BertEncoderGPU."""
        self.model = BertEncoderGPU(state_dict, device=device, ln_eps=ln_eps, with_head=True, precision=precision)
        if self.model.n_labels < 1:
            raise ValueError("the state dict has no pooler / classifier: not a sequence-classification checkpoint")
        self.tokenizer = tokenizer
        self.max_length = min(int(max_length), self.model.max_pos)
        if activation not in (None, "identity", "sigmoid"):
            raise ValueError("activation must be None / 'identity' / 'sigmoid'")
        self.activation = activation

    @classmethod
    def from_pretrained_dir(cls, path, **kw) -> "CrossEncoder":
        sd, vocab, cfg = load_state_dict_dir(path)
        tok = WordPieceTokenizer.from_vocab_file(vocab) if vocab else None
        kw.setdefault("ln_eps", float(cfg.get("layer_norm_eps", 1e-12)))
        return cls(sd, tok, **kw)

    def _post(self, logits: np.ndarray) -> np.ndarray:
        x = logits[:, 0] if logits.shape[1] == 1 else logits
        if self.activation == "sigmoid":
            x = (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)
        return x

    def predict_ids(self, seqs: Sequence[Tuple[Sequence[int], Sequence[int]]]) -> np.ndarray:
        """Scores of pre-tokenised `[CLS] query [SEP] text [SEP]` sequences (token ids, type ids)."""
        if len(seqs) == 0:
            return np.zeros(0, dtype=np.float32)
        return self._post(self.model.forward_ids(seqs, OUT_LOGITS))

    def tokenize_pairs(self, pairs: Sequence[Tuple[str, str]]):
        if self.tokenizer is None:
            raise ValueError("no vocabulary was loaded: pass pre-tokenised ids to predict_ids, or construct with a "
                             "WordPieceTokenizer (vocab.txt)")
        return [self.tokenizer.encode_pair(str(a), str(b), self.max_length) for a, b in pairs]

    def predict(self, pairs, batch_size: int = 64, show_progress_bar: bool = False, **_ignored) -> np.ndarray:
        """(query, text) string pairs -> scores.  `batch_size` only bounds host-side tokenisation here: the packed
        forward has no padding, so the scores do not depend on how pairs are batched (in the reference they do not
        either, up to fp32 rounding: padding positions are masked)."""
        if len(pairs) == 0:
            return np.zeros(0, dtype=np.float32)
        return self.predict_ids(self.tokenize_pairs(pairs))


class QueryEncoder:
    """`SentenceTransformer` on the call the reference makes: `encode([query], normalize_embeddings=True)` ->
    (n, 384) float32; CLS pooling (bge-small-en-v1.5's pooling config)."""

    def __init__(self, state_dict: Dict, tokenizer: Optional[WordPieceTokenizer] = None, *, device: int = 0,
                 max_length: int = 512, ln_eps: float = 1e-12, precision: str = "fp32"):
        """``precision`` "fp32" by default: the query vector feeds K1, whose answers are held to a 4e-7 tie band -- a bf16
        encoder would move dense scores by ~2e-3 (one short sequence per query: the fp32 forward costs microseconds)."""
        self.model = BertEncoderGPU(state_dict, device=device, ln_eps=ln_eps, with_head=False, precision=precision)
        self.tokenizer = tokenizer
        self.max_length = min(int(max_length), self.model.max_pos)

    @classmethod
    def from_pretrained_dir(cls, path, **kw) -> "QueryEncoder":
        sd, vocab, cfg = load_state_dict_dir(path)
        tok = WordPieceTokenizer.from_vocab_file(vocab) if vocab else None
        kw.setdefault("ln_eps", float(cfg.get("layer_norm_eps", 1e-12)))
        return cls(sd, tok, **kw)

    def encode_ids(self, seqs, normalize_embeddings: bool = False) -> np.ndarray:
        e = self.model.forward_ids(seqs, OUT_CLS)
        if normalize_embeddings:       # torch.nn.functional.normalize(p=2, dim=1, eps=1e-12)
            e = (e / np.maximum(np.linalg.norm(e, axis=1, keepdims=True), 1e-12)).astype(np.float32)
        return e

    def encode(self, sentences, normalize_embeddings: bool = False, **_ignored) -> np.ndarray:
        if self.tokenizer is None:
            raise ValueError("no vocabulary was loaded: pass pre-tokenised ids to encode_ids")
        single = isinstance(sentences, str)
        seqs = [self.tokenizer.encode_pair(s, None, self.max_length) for s in ([sentences] if single else sentences)]
        e = self.encode_ids(seqs, normalize_embeddings)
        return e[0] if single else e
