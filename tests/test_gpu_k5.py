"""K5 on the GPU (csrc/rr_ce.hip through the C ABI): the cross-encoder / query-encoder forward against
 (1) the committed outputs of Hugging Face `transformers` on the seeded weights (tests/golden/k5_*.npz),
 (2) the live numpy oracle on fresh inputs, and (3) properties: batch-composition invariance, chunking.

Two precisions (include/rr_hip.h RR_CE_PRECISION_*), two bars:
  * "fp32" (the default: the reference's arithmetic, fp32 operands end to end) is held to the north star's bar:
    logits, embeddings and fused scores within 1e-5 of `transformers` / the oracle, returned skus identical;
  * "bf16" (the fast path: bf16 MFMA operands -- weights rounded once, activations at every producer --, fp32
    accumulation and residual stream): on the seeded O(1) weights ~3e-3 on logits of magnitude ~0.5 and ~1e-2 on
    hidden states of magnitude ~1; bars 2.5e-2 / 6e-2 / 8e-3 absolute -- an indexing or layout bug shows up as O(1)."""
import numpy as np
import pandas as pd
import pytest

from conftest import GOLDEN
from oracle import cross_encoder as OC
from oracle.bm25 import BM25OkapiOracle
from oracle.pipeline import run_search_oracle
from parity import assert_ranking_matches
from review_recommender_amd import synth
from review_recommender_amd.cross_encoder import OUT_HIDDEN, BertEncoderGPU, CrossEncoder, QueryEncoder
from review_recommender_amd.engine import SearchEngine
from review_recommender_amd.wordpiece import WordPieceTokenizer

pytestmark = pytest.mark.gpu
LOGIT_TOL, HIDDEN_TOL, EMB_TOL = 2.5e-2, 6e-2, 8e-3          # "bf16"
F32_LOGIT_TOL, F32_HIDDEN_TOL, F32_EMB_TOL = 1e-5, 5e-5, 1e-5      # "fp32" (hidden states of magnitude ~3: 5e-5 is ~2 ulp)


def split(fx):
    cu = fx["cu_seqlens"]
    return [(fx["token_ids"][cu[i]:cu[i + 1]], fx["type_ids"][cu[i]:cu[i + 1]]) for i in range(len(cu) - 1)]


@pytest.fixture(scope="module")
def ce_world():
    fx = np.load(GOLDEN / "k5_cross_encoder.npz")
    sd = synth.bert_state_dict(int(fx["seed"]), n_layers=6, n_labels=1)
    return fx, sd, CrossEncoder(sd, precision="bf16")


@pytest.fixture(scope="module")
def ce_world_f32():
    fx = np.load(GOLDEN / "k5_cross_encoder.npz")
    sd = synth.bert_state_dict(int(fx["seed"]), n_layers=6, n_labels=1)
    return fx, sd, CrossEncoder(sd)                              # default precision: fp32


def test_fp32_mode_matches_the_transformers_fixture_to_1e_5(ce_world_f32):
    """Reference precision (RR_CE_PRECISION_F32): logits within 1e-5 of what `transformers` produced in fp32, the
    ranking of the fixture's pairs identical wherever float32 separates them, hidden states to a few ulp."""
    fx, sd, ce = ce_world_f32
    assert ce.model.precision == "fp32"
    seqs = split(fx)
    got = ce.predict_ids(seqs)
    err = np.abs(got - fx["logits"])
    print("fp32 mode: max |logit error|", err.max())
    assert err.max() < F32_LOGIT_TOL
    order, want_order = np.argsort(-got, kind="stable"), np.argsort(-fx["logits"], kind="stable")
    gaps = -np.diff(fx["logits"][want_order])
    for i in range(len(order)):
        if (i == 0 or gaps[i - 1] > 2 * F32_LOGIT_TOL) and (i == len(order) - 1 or gaps[i] > 2 * F32_LOGIT_TOL):
            assert order[i] == want_order[i]
    hid = ce.model.forward_ids(seqs, OUT_HIDDEN)
    cu = fx["cu_seqlens"]
    worst = 0.0
    for i in range(len(seqs)):
        for j, r in enumerate(fx["hidden_rows"][i]):
            if r >= 0:
                worst = max(worst, float(np.abs(hid[cu[i] + r] - fx["hidden_vals"][i, j]).max()))
    print("fp32 mode: max |hidden error|", worst)
    assert worst < F32_HIDDEN_TOL


def test_fp32_mode_every_length_and_batch_invariance(ce_world_f32):
    fx, sd, ce = ce_world_f32
    rng = np.random.default_rng(3)
    seqs = []
    for n in list(range(1, 40)) + [63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512]:
        ids = rng.integers(0, 30522, n).astype(np.int32)
        ids[0] = 101
        seqs.append((ids, (np.arange(n) > n // 3).astype(np.int32)))
    got = ce.predict_ids(seqs)
    want = OC.predict_oracle(sd, seqs, n_layers=6)
    print("fp32 mode vs the numpy oracle:", np.abs(got - want).max())
    assert np.abs(got - want).max() < 2e-5                        # (the numpy oracle itself is fp32 in another summation order)
    alone = np.concatenate([ce.predict_ids([s]) for s in seqs[:6]])
    assert np.array_equal(got[:6], alone)                         # packed sequences: no dependence on the batch
    assert np.array_equal(got, ce.predict_ids(seqs[::-1])[::-1])


def test_cross_encoder_logits_match_the_transformers_fixture(ce_world):
    fx, sd, ce = ce_world
    seqs = split(fx)
    got = ce.predict_ids(seqs)
    assert got.shape == (len(seqs),) and got.dtype == np.float32
    err = np.abs(got - fx["logits"])
    print("max |logit error|", err.max(), "of logits spanning", fx["logits"].min(), fx["logits"].max())
    assert err.max() < LOGIT_TOL
    # ranking agreement where the fp32 logits are separated by more than the tolerance
    order = np.argsort(-fx["logits"])
    gaps = -np.diff(fx["logits"][order])
    for a, b, gap in zip(order[:-1], order[1:], gaps):
        if gap > 2 * LOGIT_TOL:
            assert got[a] > got[b]


def test_hidden_states_match_the_transformers_fixture(ce_world):
    fx, sd, ce = ce_world
    seqs = split(fx)
    hid = ce.model.forward_ids(seqs, OUT_HIDDEN)
    cu = fx["cu_seqlens"]
    assert hid.shape == (int(cu[-1]), 384)
    worst = 0.0
    for i in range(len(seqs)):
        for j, r in enumerate(fx["hidden_rows"][i]):
            if r >= 0:
                worst = max(worst, float(np.abs(hid[cu[i] + r] - fx["hidden_vals"][i, j]).max()))
    print("max |hidden error|", worst)
    assert worst < HIDDEN_TOL


def test_every_length_from_1_to_70_and_the_tile_edges_against_the_live_oracle(ce_world):
    fx, sd, ce = ce_world
    rng = np.random.default_rng(3)
    seqs = []
    for n in list(range(1, 71)) + [95, 96, 97, 127, 128, 129, 160, 255, 256, 257, 384, 511, 512]:
        ids = rng.integers(0, 30522, n).astype(np.int32)
        ids[0] = 101
        typ = (np.arange(n) > n // 3).astype(np.int32)
        seqs.append((ids, typ))
    got = ce.predict_ids(seqs)
    want = OC.predict_oracle(sd, seqs, n_layers=6)
    assert np.abs(got - want).max() < LOGIT_TOL


def test_scores_do_not_depend_on_batch_composition_or_chunking(ce_world):
    """Packed sequences attend only to themselves and every output element is one fixed-order dot product: a pair
    scores bitwise the same alone, in any batch, at any position, and across activation-scratch chunks."""
    fx, sd, ce = ce_world
    seqs = split(fx)[:24]
    whole = ce.predict_ids(seqs)
    alone = np.concatenate([ce.predict_ids([s]) for s in seqs[:8]])
    assert np.array_equal(whole[:8], alone)
    rev = ce.predict_ids(seqs[::-1])[::-1]
    assert np.array_equal(whole, rev)
    small = BertEncoderGPU(sd, max_tokens_per_call=700, precision="bf16")          # forces several chunks
    chunked = small.forward_ids(seqs, 0)[:, 0]
    assert np.array_equal(whole, chunked)
    small.close()


def vocab_tokenizer():
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list(synth.WORDS) + ["##s", "##ing", "for", "the", "of", "and"]
    return WordPieceTokenizer({w: i for i, w in enumerate(words)})


def test_predict_on_text_pairs_is_the_reference_call_shape():
    """predict(pairs, batch_size=64, show_progress_bar=False) (app/app_product_search.py:278): tokenise + forward;
    equals the pre-tokenised entry point, empty input gives an empty array, Sigmoid is a parameter."""
    tok = vocab_tokenizer()
    sd = synth.bert_state_dict(11, n_layers=6, n_labels=1, vocab=len(tok.vocab))
    ce = CrossEncoder(sd, tok, precision="bf16")
    texts = synth.text_corpus(50, 4, mean_len=60)
    pairs = [("wireless cat socks for running", t[:2000]) for t in texts]
    got = ce.predict(pairs, batch_size=64, show_progress_bar=False)
    seqs = [tok.encode_pair(a, b, 512) for a, b in pairs]
    assert np.array_equal(got, ce.predict_ids(seqs)) and got.dtype == np.float32 and got.shape == (50,)
    assert np.abs(got - OC.predict_oracle(sd, seqs, 6)).max() < LOGIT_TOL
    assert ce.predict([]).shape == (0,)
    sig = CrossEncoder(sd, tok, activation="sigmoid", precision="bf16").predict(pairs[:5])
    np.testing.assert_allclose(sig, 1 / (1 + np.exp(-got[:5].astype(np.float64))), atol=1e-6)
    with pytest.raises(ValueError):
        CrossEncoder(sd).predict(pairs[:1])                   # no vocabulary: ids only
    with pytest.raises(ValueError):
        ce.predict_ids([(np.arange(600), np.zeros(600, int))])  # longer than the position table


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_query_encoder_matches_the_transformers_fixture(precision):
    fx = np.load(GOLDEN / "k5_query_encoder.npz")
    sd = synth.bert_state_dict(int(fx["seed"]), n_layers=12, n_labels=0, prefix="")
    enc = QueryEncoder(sd, precision=precision)
    emb = enc.encode_ids(split(fx), normalize_embeddings=True)
    assert emb.shape == fx["embeddings"].shape
    print(precision, "max |embedding error|", np.abs(emb - fx["embeddings"]).max())
    assert np.abs(emb - fx["embeddings"]).max() < (F32_EMB_TOL if precision == "fp32" else EMB_TOL)
    np.testing.assert_allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-5)
    cos = (emb * fx["embeddings"]).sum(axis=1)
    assert cos.min() > (1 - 1e-6 if precision == "fp32" else 0.9995)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_run_search_reranks_with_the_real_kernel_end_to_end(precision):
    """a12: run_search with rerank_k > 0 through the HIP cross-encoder (not a stand-in) against the oracle pipeline
    fed the numpy oracle's scores; the query encoder runs on the GPU too (f4) and feeds K1 without a host model.
    "fp32" (reference precision): the returned skus are the oracle's, `_final` within 1e-5 (BASELINE config 5's bar);
    "bf16": the fast path's stated tolerance."""
    n = 3000
    tok = vocab_tokenizer()
    V = synth.unit_rows(n, 384, 21)
    n_rev, stars = synth.metadata(n, 22)
    texts = synth.text_corpus(n, 23, mean_len=40)
    meta = pd.DataFrame({"sku": synth.skus(n), "n_reviews": n_rev, "avg_stars": stars, "agg_text": texts})
    corpus = [t.split() for t in texts]
    blob = {"skus": meta["sku"].tolist(), "corpus": corpus}
    sd = synth.bert_state_dict(31, n_layers=6, n_labels=1, vocab=len(tok.vocab))
    sd_q = synth.bert_state_dict(32, n_layers=12, n_labels=0, prefix="", vocab=len(tok.vocab))
    ce = CrossEncoder(sd, tok, precision=precision)
    qe = QueryEncoder(sd_q, tok, precision=precision)
    engine = SearchEngine(meta, V, blob, encoder=qe, cross_encoder=ce, normalize=False)
    cfg = dict(k=20, rerank_k=40, w_dense=0.4, w_bm25=0.2, w_rerank=0.3, w_prior=0.1, w_best=0.0, prior_C=20.0,
               min_reviews=5, gate_penalty=0.5)
    query = "blue insulated coffee mug"
    got, _, dbg = engine.run_search(query, cfg["k"], cfg["rerank_k"], cfg["w_dense"], cfg["w_bm25"], cfg["w_rerank"],
                                    cfg["w_prior"], cfg["w_best"], cfg["prior_C"], False, 0, cfg["min_reviews"],
                                    cfg["gate_penalty"])
    qvec = engine.encode(query)                                   # GPU query encoder, l2-normalised
    assert abs(np.linalg.norm(qvec) - 1) < 1e-5
    q_or = OC.encode_oracle(sd_q, [tok.encode_pair(query, None, 512)], n_layers=12, normalize=True)[0]
    assert np.abs(qvec - q_or).max() < (F32_EMB_TOL if precision == "fp32" else EMB_TOL)
    rr = lambda pairs: OC.predict_oracle(sd, [tok.encode_pair(a, b, 512) for a, b in pairs], 6)
    want, _, _, cand = run_search_oracle(query=query, qvec=qvec, meta=meta, V=V, bm25=BM25OkapiOracle(corpus),
                                         bm25_skus=blob["skus"], rerank_fn=rr, **cfg)
    assert got["_rerank"].dtype == np.float32
    g = got.set_index("sku")
    c = cand.set_index("sku")
    # min-max of the reranker scores stretches their error by 1 / (max - min) of ~40 logits
    span = float(np.ptp(rr([(query, t[:2000]) for t in cand["agg_text"].tolist()[:cfg["rerank_k"]]])))
    if precision == "fp32":
        # the north star's bar: ids bit-exact, fused scores within 1e-5 (min-max stretches the 1e-5 of the logits by 1 / span)
        # order exact outside bands of finals closer than twice the score bar; inside a band the sets agree
        assert_ranking_matches(got["sku"].tolist(), want["sku"].tolist(), want["_final"].values, 2e-5,
                               cand["sku"].tolist(), cand["_final"].values)
        assert np.abs(g["_rerank"].values - c.loc[g.index, "_rerank"].values).max() < 2 * 2e-5 / span + 1e-6
        assert np.abs(g["_final"].values - c.loc[g.index, "_final"].values).max() < 1e-5
        return
    tol = LOGIT_TOL / span * 2
    assert np.abs(g["_rerank"].values - c.loc[g.index, "_rerank"].values).max() < tol
    assert np.abs(g["_final"].values - c.loc[g.index, "_final"].values).max() < cfg["w_rerank"] * tol + 1e-5
    top_want = set(want["sku"][:10])
    assert len(top_want & set(got["sku"])) >= 8


def write_model_dir(path, sd, vocab_words, with_config=True):
    """A local Hugging Face style model directory: model.safetensors + vocab.txt (+ config.json)."""
    import json
    from safetensors.numpy import save_file
    path.mkdir(parents=True, exist_ok=True)
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, str(path / "model.safetensors"))
    (path / "vocab.txt").write_text("\n".join(vocab_words) + "\n", encoding="utf-8")
    if with_config:
        (path / "config.json").write_text(json.dumps({"layer_norm_eps": 1e-12, "hidden_size": 384}))


def test_cli_runs_offline_from_local_model_directories(tmp_path, capsys):
    """app/test.py end to end with BOTH models on the GPU, loaded from local directories (nothing fetched): the query is
    encoded by QueryEncoder, the top rerank_k candidates rescored by CrossEncoder, JSON rows against the oracle fed the
    numpy oracle's embedding and scores."""
    import json
    from oracle.pipeline import cli_rows
    from review_recommender_amd import artifacts
    from review_recommender_amd.cli import main
    tok = vocab_tokenizer()
    words = sorted(tok.vocab, key=tok.vocab.get)
    sd_ce = synth.bert_state_dict(41, n_layers=6, n_labels=1, vocab=len(words))
    sd_q = synth.bert_state_dict(42, n_layers=12, n_labels=0, prefix="", vocab=len(words))
    write_model_dir(tmp_path / "ce", sd_ce, words)
    write_model_dir(tmp_path / "enc", sd_q, words, with_config=False)
    n = 2500
    V = synth.unit_rows(n, 384, 51)
    n_rev, stars = synth.metadata(n, 52)
    meta = pd.DataFrame({"sku": synth.skus(n), "n_reviews": n_rev, "avg_stars": stars, "last_ts": np.arange(n),
                         "agg_text": synth.text_corpus(n, 53, mean_len=30)})
    blob = artifacts.build_bm25_blob(meta)
    artifacts.save_artifacts(tmp_path / "data", meta, V, blob)
    out = tmp_path / "out.json"
    query = "soft cotton shirt"
    rc = main(["-q", query, "-k", "8", "--rerank_k", "30", "--data-dir", str(tmp_path / "data"), "--json-out", str(out),
               "--emb-model-dir", str(tmp_path / "enc"), "--rerank-model-dir", str(tmp_path / "ce"), "--gate-penalty", "1.0"])
    assert rc == 0 and "Top results:" in capsys.readouterr().out
    got = json.loads(out.read_text())["results"]
    qvec = OC.encode_oracle(sd_q, [tok.encode_pair(query, None, 512)], n_layers=12, normalize=True)[0]
    qvec_gpu = QueryEncoder.from_pretrained_dir(tmp_path / "enc").encode([query], normalize_embeddings=True)[0]
    assert np.abs(qvec - qvec_gpu).max() < F32_EMB_TOL            # (the CLI loads both models in the default precision: fp32)
    rr = lambda pairs: OC.predict_oracle(sd_ce, [tok.encode_pair(a, b, 512) for a, b in pairs], 6)
    from oracle.primitives import l2_normalize
    want, _, _, cand = run_search_oracle(query=query, qvec=qvec_gpu, meta=meta, V=l2_normalize(V),
                                         bm25=BM25OkapiOracle(blob["corpus"]), bm25_skus=blob["skus"], k=8, rerank_k=30,
                                         w_dense=0.55, w_bm25=0.15, w_rerank=0.15, w_prior=0.10, w_best=0.05, prior_C=20.0,
                                         gate_penalty=1.0, flavour="cli", rerank_fn=rr)
    exp = {r["sku"]: r for r in cli_rows(cand)}                     # every pool row, by sku
    span = float(np.ptp(rr([(query, t[:2000]) for t in cand["agg_text"].tolist()[:30]])))
    tol = 0.15 * (2 * 2e-5 / span) + 1.01e-4                      # (the JSON rows carry 4 decimals)
    assert len(got) == 8 and set(r["sku"] for r in got) == set(want["sku"])
    for g in got:
        e = exp[g["sku"]]
        assert abs(g["rerank"] - e["rerank"]) <= 2 * 2e-5 / span + 1.01e-4
        assert abs(g["score"] - e["score"]) <= tol
        for key in ("dense", "bm25", "prior"):
            assert abs(g[key] - e[key]) <= 1.01e-4


K5_SWITCH_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
from review_recommender_amd import synth
from review_recommender_amd.cross_encoder import CrossEncoder
fx = np.load(%r)
cu = fx["cu_seqlens"]
seqs = [(fx["token_ids"][cu[i]:cu[i + 1]], fx["type_ids"][cu[i]:cu[i + 1]]) for i in range(len(cu) - 1)]
ce = CrossEncoder(synth.bert_state_dict(int(fx["seed"]), n_layers=6, n_labels=1), precision="bf16")
print("LOGITS " + " ".join(repr(float(v)) for v in ce.predict_ids(seqs)))
"""


@pytest.mark.parametrize("switch", ["RR_CE_QKV_TILED", "RR_CE_OPROJ_APART", "RR_CE_UNFUSED"])
def test_bf16_path_switches_stay_inside_the_bf16_bar(ce_world, switch):
    """The A/B forms of the fast path (the QKV projection as the tiled GEMM; the attention output projection + LayerNorm as
    its own launch; the FFN as two GEMM launches with the polynomial GELU) are read once per process: each runs the fixture in a child and must meet the same bar as the
    default form, and agree with it to bf16 rounding."""
    import os
    import subprocess
    import sys
    fx, sd, ce = ce_world
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = K5_SWITCH_CHILD % (root, str(GOLDEN / "k5_cross_encoder.npz"))
    p = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, **{switch: "1"}), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    got = np.array([float(v) for v in [l for l in p.stdout.splitlines() if l.startswith("LOGITS ")][-1].split()[1:]], dtype=np.float32)
    assert np.abs(got - fx["logits"]).max() < LOGIT_TOL
    assert np.abs(got - ce.predict_ids(split(fx))).max() < LOGIT_TOL


def test_fp32_mode_on_the_fp32_input_matrix_instruction_meets_the_same_bar(ce_world_f32):
    """The fp32 mode's GEMMs run by operand splitting on the bf16 matrix cores (three bf16 terms per operand, six products:
    exact to the rounding of one fp32 multiply); RR_CE_F32_MFMA=1 runs them on v_mfma_f32_32x32x2_f32 instead.  Both forms
    are held to the 1e-5 bar and agree with each other inside it."""
    import os
    import subprocess
    import sys
    fx, sd, ce = ce_world_f32
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = (K5_SWITCH_CHILD % (root, str(GOLDEN / "k5_cross_encoder.npz"))).replace('precision="bf16"', 'precision="fp32"')
    p = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, RR_CE_F32_MFMA="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    got = np.array([float(v) for v in [l for l in p.stdout.splitlines() if l.startswith("LOGITS ")][-1].split()[1:]], dtype=np.float32)
    assert np.abs(got - fx["logits"]).max() < F32_LOGIT_TOL
    assert np.abs(got - ce.predict_ids(split(fx))).max() < F32_LOGIT_TOL


def test_fp32_mode_two_splits_agree_and_a_value_beyond_fp16_switches_the_handle(ce_world_f32):
    """RR_CE_PRECISION_F32 multiplies fp16 pairs (hi + lo / 2048: three MFMA products, csrc/rr_ce_h2.hip); the bf16 three-term
    kernels (six products, any fp32 range) stay behind `set_wide_range`.  (1) On the seeded model the two agree to fp32
    rounding on logits and hidden states.  (2) A model whose FFN output exceeds 65504 cannot be split into fp16: the
    device raises its flag, the logits of that pass are NaN -- never a wrong finite number --, `forward_ids` notices,
    switches the handle and returns what a wide-range handle returns, bit for bit."""
    fx, sd, ce = ce_world_f32
    seqs = split(fx)[:12]
    wide = CrossEncoder(sd)
    wide.model.set_wide_range(True)
    a, b = ce.predict_ids(seqs), wide.predict_ids(seqs)
    assert not ce.model.out_of_range() and not wide.model.out_of_range()
    print("fp16-pair vs bf16-triple logits: max |diff|", np.abs(a - b).max())
    assert np.abs(a - b).max() < 4e-6
    ha, hb = ce.model.forward_ids(seqs, OUT_HIDDEN), wide.model.forward_ids(seqs, OUT_HIDDEN)
    assert np.abs(ha - hb).max() < 2e-5
    # (2) blow one layer's FFN up: |intermediate| reaches ~1e5
    big = dict(sd)
    key = [k for k in big if k.endswith("encoder.layer.2.intermediate.dense.weight")][0]
    big[key] = np.asarray(big[key], dtype=np.float32) * np.float32(4.0e4)
    ref = CrossEncoder(big)
    ref.model.set_wide_range(True)
    want = ref.predict_ids(seqs)
    assert np.isfinite(want).all()
    auto = CrossEncoder(big)
    ids = np.concatenate([np.asarray(s[0], dtype=np.int32) for s in seqs])
    import torch
    lens = np.array([len(s[0]) for s in seqs])
    cu = np.zeros(len(seqs) + 1, dtype=np.int32)
    np.cumsum(lens, out=cu[1:])
    dev = torch.device("cuda", 0)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.int32)).to(dev)
    pos = (np.arange(cu[-1]) - np.repeat(cu[:-1], lens)).astype(np.int32)
    typ = np.concatenate([np.asarray(s[1], dtype=np.int32) for s in seqs])
    raw = auto.model.forward_packed_dev(t(ids), t(typ), t(pos), t(cu), len(seqs), int(lens.max()), 0)
    torch.cuda.synchronize()
    assert auto.model.out_of_range() and bool(torch.isnan(raw).all())
    with pytest.warns(UserWarning, match="fp16 range"):
        got = auto.predict_ids(seqs)
    assert np.array_equal(got, want)
    assert not auto.model.out_of_range()                     # (the handle now runs the wide-range kernels)


H2_GEMM_CHILD = r"""
import os, sys
os.environ["RR_DEBUG_HARNESS"] = "1"
sys.path.insert(0, %r)
import ctypes as C
import numpy as np, torch
from scipy.special import erf
from review_recommender_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(21)
EPI_F32, EPI_H2, EPI_GELU = 0, 1, 2
def run(epi, M, N, K, x, w, b, qcols=0):
    dx, dw, db = (torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev) for a in (x, w, b))
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
    flag = C.c_int32(-1)
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.rr_debug_ce_h2_gemm(epi, M, N, K, p(dx), p(dw), p(db), qcols, p(out), C.byref(flag)), "rr_debug_ce_h2_gemm")
    return out.cpu().numpy().astype(np.float64), flag.value
cases = [("out-proj shape, ragged M", EPI_F32, 1000, 384, 384), ("FFN-2 shape", EPI_F32, 517, 384, 1536), ("one token", EPI_F32, 1, 128, 32),
         ("QKV shape, h2 out, Q scaled", EPI_H2, 700, 1152, 384), ("FFN-1 shape, GELU", EPI_GELU, 300, 1536, 384)]
for name, epi, M, N, K in cases:
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    # a tenth of the entries around and below fp16's subnormal range, some exact zeros
    tiny = rng.random((M, K)) < 0.1
    x[tiny] = (10.0 ** rng.uniform(-9, -4, tiny.sum()) * rng.choice([-1, 1], tiny.sum())).astype(np.float32)
    x[rng.random((M, K)) < 0.01] = 0.0
    b = rng.standard_normal(N).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    mag = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(b)
    qcols = 384 if epi == EPI_H2 else 0
    if epi == EPI_H2:
        ref[:, :qcols] *= 0.5; mag[:, :qcols] *= 0.5
    if epi == EPI_GELU:
        ref = 0.5 * ref * (1.0 + erf(ref / np.sqrt(2.0)))
    got, flag = run(epi, M, N, K, x, w, b, qcols)
    f32 = (torch.from_numpy(x).to(dev) @ torch.from_numpy(w).to(dev).T).cpu().numpy().astype(np.float64) + b
    e_f32 = np.abs(f32 - (x.astype(np.float64) @ w.astype(np.float64).T + b)).max() if epi == EPI_F32 else float("nan")
    print("CASE|%%s|%%d|%%.3e|%%.3e|%%d" %% (name, flag, float((np.abs(got - ref) / mag).max()), float(e_f32 / mag.max()), int(np.isfinite(got).all())))
# a value beyond fp16's range in an h2 result raises the flag; an fp32 result of the same product does not split anything
x = np.full((4, 32), 300.0, np.float32); w = np.full((128, 32), 10.0, np.float32); b = np.zeros(128, np.float32)
_, flag_h2 = run(EPI_H2, 4, 128, 32, x, w, b)
got, flag_f32 = run(EPI_F32, 4, 128, 32, x, w, b)
print("RANGE|%%d|%%d|%%.1f" %% (flag_h2, flag_f32, got[0, 0]))
"""


def test_h2_gemm_kernel_against_float64_products():
    """ce_gemm_h2 by itself (through the harness library, in a child process): packed operands -> three fp16 products ->
    the three epilogues, on the model's four GEMM shapes with ragged token counts, a tenth of the activations in and below
    fp16's SUBNORMAL range (1e-9 .. 1e-4) and exact zeros.  Every output within 3e-7 of sum |x w| of the float64 product
    (an fp32 GEMM of the same data: ~1e-7; the h2 epilogues add one fp16-pair rounding, 2^-22); GELU against the float64 erf
    form within 6e-7; the range flag up exactly when a value that is SPLIT leaves fp16's range."""
    from review_recommender_amd.build import DEBUG_LIB_PATH
    import pathlib
    import subprocess
    import sys
    if not DEBUG_LIB_PATH.exists():
        pytest.skip("librr_hip_dbg.so not built (python review-recommender_amd/build.py --debug)")
    root = str(pathlib.Path(__file__).resolve().parent.parent)
    p = subprocess.run([sys.executable, "-c", H2_GEMM_CHILD % root], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    cases = [l.split("|") for l in p.stdout.splitlines() if l.startswith("CASE|")]
    assert len(cases) == 5, p.stdout
    for _, name, flag, err, err_f32, finite in cases:
        print(name, "error / sum|xw|:", err, " fp32 GEMM:", err_f32)
        assert int(flag) == 0 and int(finite) == 1, name
        assert float(err) < (6e-7 if "GELU" in name else 3e-7), (name, err)
    rng_line = [l.split("|") for l in p.stdout.splitlines() if l.startswith("RANGE|")][0]
    assert int(rng_line[1]) == 1 and int(rng_line[2]) == 0 and float(rng_line[3]) == 96000.0
