"""Oracle: the cross-encoder / query-encoder forward (numpy fp32 restatement of Hugging Face BERT).

TEST INFRASTRUCTURE -- never imported by the product path.

The reference reranks with ``CrossEncoder(RERANK_MODEL).predict(pairs, batch_size=64,
show_progress_bar=False)`` (app/app_product_search.py:71-86,271-282; app/test.py:96-104,217-225) and encodes
queries with ``SentenceTransformer(EMB_MODEL).encode([q], normalize_embeddings=True)``
(app/app_product_search.py:53-69,250-251; app/test.py:91-94,232).  Both are third-party
(`sentence-transformers` 5.1.0, pinned in requirements.txt:60, over `transformers`): sentence-transformers is
not installed here and the model weights / vocabularies cannot be fetched, so this file restates the
published architecture the two model names resolve to --

  cross-encoder/ms-marco-MiniLM-L-6-v2  ->  BertForSequenceClassification(hidden 384, 6 layers, 12 heads,
                                            intermediate 1536, 512 positions, num_labels 1)
  BAAI/bge-small-en-v1.5                ->  BertModel(hidden 384, 12 layers, 12 heads, intermediate 1536),
                                            CLS pooling, l2 normalisation

-- following transformers' modeling_bert.py (BertEmbeddings, BertSelfAttention, BertSelfOutput,
BertIntermediate (gelu = erf form), BertOutput, BertPooler, classifier).  **Pinned** by logits and hidden
states that `transformers` 5.15 itself produced in the build container on seeded random weights
(tests/golden/k5_*.npz, generator tests/golden/make_k5_golden.py).  Unpinned (recorded, not fixable offline):
whether CrossEncoder.predict applies Identity or Sigmoid to the single logit (depends on the hub model's
config); raw logits are returned, the activation is the caller's parameter.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
from scipy.special import erf

F32 = np.float32


def _ln(x: np.ndarray, g: np.ndarray, b: np.ndarray, eps: float) -> np.ndarray:
    mean = x.mean(axis=-1, keepdims=True, dtype=F32)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True, dtype=F32)
    return ((x - mean) / np.sqrt(var + F32(eps)) * g + b).astype(F32)


def _lin(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    return (x @ w.T + b).astype(F32)


def bert_hidden(sd: Dict[str, np.ndarray], ids: np.ndarray, type_ids: np.ndarray, n_layers: int,
                n_heads: int = 12, eps: float = 1e-12, prefix: str = "bert.") -> np.ndarray:
    """last_hidden_state (S, H) of ONE unpadded sequence."""
    S = len(ids)
    e = prefix + "embeddings."
    x = (sd[e + "word_embeddings.weight"][ids] + sd[e + "token_type_embeddings.weight"][type_ids]
         + sd[e + "position_embeddings.weight"][np.arange(S)]).astype(F32)
    x = _ln(x, sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"], eps)
    H = x.shape[1]
    d = H // n_heads
    for l in range(n_layers):
        p = f"{prefix}encoder.layer.{l}."
        q = _lin(x, sd[p + "attention.self.query.weight"], sd[p + "attention.self.query.bias"])
        k = _lin(x, sd[p + "attention.self.key.weight"], sd[p + "attention.self.key.bias"])
        v = _lin(x, sd[p + "attention.self.value.weight"], sd[p + "attention.self.value.bias"])
        qh, kh, vh = (t.reshape(S, n_heads, d).transpose(1, 0, 2) for t in (q, k, v))
        sc = (qh @ kh.transpose(0, 2, 1) / F32(np.sqrt(d))).astype(F32)
        sc = sc - sc.max(axis=-1, keepdims=True)
        pr = np.exp(sc)
        pr = (pr / pr.sum(axis=-1, keepdims=True)).astype(F32)
        ctx = (pr @ vh).transpose(1, 0, 2).reshape(S, H).astype(F32)
        a = _lin(ctx, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
        x = _ln(a + x, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], eps)
        h = _lin(x, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"])
        h = (h * F32(0.5) * (F32(1.0) + erf(h / F32(np.sqrt(2.0))))).astype(F32)      # gelu, erf form
        o = _lin(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
        x = _ln(o + x, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
    return x


def cross_encoder_logits(sd: Dict[str, np.ndarray], seqs: Sequence[Tuple[np.ndarray, np.ndarray]], n_layers: int,
                         prefix: str = "bert.") -> np.ndarray:
    """BertForSequenceClassification logits (n_seqs, n_labels): classifier(tanh(pooler(hidden[0])))."""
    out = []
    for ids, typ in seqs:
        h = bert_hidden(sd, np.asarray(ids), np.asarray(typ), n_layers, prefix=prefix)
        pooled = np.tanh(_lin(h[0], sd[prefix + "pooler.dense.weight"], sd[prefix + "pooler.dense.bias"]))
        out.append(_lin(pooled.astype(F32), sd["classifier.weight"], sd["classifier.bias"]))
    return np.stack(out).astype(F32)


def predict_oracle(sd, seqs, n_layers: int = 6) -> np.ndarray:
    """CrossEncoder.predict for num_labels == 1: one float32 score per pair (raw logit)."""
    return cross_encoder_logits(sd, seqs, n_layers)[:, 0]


def encode_oracle(sd, seqs, n_layers: int = 12, normalize: bool = True, prefix: str = "") -> np.ndarray:
    """SentenceTransformer.encode with CLS pooling (bge-small): hidden[0], optionally l2-normalised
    (torch.nn.functional.normalize: x / max(||x||, 1e-12))."""
    out = []
    for ids, typ in seqs:
        c = bert_hidden(sd, np.asarray(ids), np.asarray(typ), n_layers, prefix=prefix)[0]
        if normalize:
            c = c / max(float(np.linalg.norm(c)), 1e-12)
        out.append(c.astype(F32))
    return np.stack(out)
