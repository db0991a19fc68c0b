"""K5 oracle pinned (CPU): oracle/cross_encoder.py against outputs of Hugging Face `transformers` BERT on the
seeded weights (tests/golden/k5_*.npz, generated in the build container by tests/golden/make_k5_golden.py),
and the WordPiece tokenizer against transformers.BertTokenizer's output (tests/golden/k5_tokenizer.json)."""
import json

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import cross_encoder as OC
from review_recommender_amd import synth
from review_recommender_amd.wordpiece import WordPieceTokenizer, basic_tokenize


def split(fx):
    cu = fx["cu_seqlens"]
    return [(fx["token_ids"][cu[i]:cu[i + 1]], fx["type_ids"][cu[i]:cu[i + 1]]) for i in range(len(cu) - 1)]


def test_oracle_cross_encoder_matches_transformers_fixture():
    fx = np.load(GOLDEN / "k5_cross_encoder.npz")
    sd = synth.bert_state_dict(int(fx["seed"]), n_layers=int(fx["n_layers"]), n_labels=1)
    seqs = split(fx)
    pick = [i for i, (ids, _) in enumerate(seqs) if len(ids) <= 140][:14] + \
           [i for i, (ids, _) in enumerate(seqs) if len(ids) in (257, 512)]
    assert len(pick) >= 12
    got = OC.predict_oracle(sd, [seqs[i] for i in pick], n_layers=6)
    np.testing.assert_allclose(got, fx["logits"][pick], atol=3e-5, rtol=0)
    for i in pick[:6]:
        h = OC.bert_hidden(sd, *seqs[i], n_layers=6)
        rows = fx["hidden_rows"][i]
        for j, r in enumerate(rows):
            if r >= 0:
                np.testing.assert_allclose(h[r], fx["hidden_vals"][i, j], atol=5e-5, rtol=0)


def test_oracle_query_encoder_matches_transformers_fixture():
    fx = np.load(GOLDEN / "k5_query_encoder.npz")
    sd = synth.bert_state_dict(int(fx["seed"]), n_layers=12, n_labels=0, prefix="")
    emb = OC.encode_oracle(sd, split(fx), n_layers=12, normalize=True)
    np.testing.assert_allclose(emb, fx["embeddings"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-6)


def test_wordpiece_matches_transformers_bert_tokenizer():
    fx = json.loads((GOLDEN / "k5_tokenizer.json").read_text())
    tok = WordPieceTokenizer({w: i for i, w in enumerate(fx["vocab"])})
    for text, want in zip(fx["texts"], fx["single_max32"]):
        ids, typ = tok.encode_pair(text, None, 32)
        assert ids.tolist() == want and not typ.any(), text
    for p in fx["pairs"]:
        ids, typ = tok.encode_pair(p["a"], p["b"], p["max_length"])
        assert ids.tolist() == p["input_ids"], (p["a"][:30], p["b"][:30], p["max_length"])
        assert typ.tolist() == p["token_type_ids"]
        assert len(ids) <= p["max_length"]


def test_basic_tokenizer_rules():
    assert basic_tokenize("Kid's  café, naïve!") == ["kid", "'", "s", "cafe", ",", "naive", "!"]
    assert basic_tokenize("a\x00b�c\td") == ["abc", "d"]
    assert basic_tokenize("中文ab") == ["中", "文", "ab"]
    assert basic_tokenize("") == []


def test_seeded_weights_are_deterministic_and_hf_named():
    a = synth.bert_state_dict(7, n_layers=2)
    b = synth.bert_state_dict(7, n_layers=2)
    assert list(a) == list(b) and all(np.array_equal(a[k], b[k]) for k in a)
    assert len(a) == 5 + 16 * 2 + 4 and "bert.encoder.layer.1.output.LayerNorm.bias" in a
    assert a["classifier.weight"].shape == (1, 384) and a["bert.encoder.layer.0.intermediate.dense.weight"].shape == (1536, 384)
