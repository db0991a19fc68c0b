#!/usr/bin/env python3
"""K5 timing: cross-encoder forward over packed pairs (BASELINE config 5 shape: pairs of <= 512 tokens).
Prints one JSON line with pairs/s, tokens/s and the bf16 MFMA roofline fraction (algorithmic FLOPs of the
GEMMs + attention / HIP-event time of the forward / 2.5 PFLOP/s)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--len", type=int, default=512, help="tokens per pair (fixed), or 0 for lengths uniform in [64, 512]")
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16",
                    help="bf16 = the fast path (roofline: bf16 MFMA); fp32 = reference precision: every fp32 product as three fp16 MFMA "
                         "products (roofline: 3 x the executed FLOPs against the fp16 MFMA peak; RR_CE_F32_SPLIT=bf16x3: six bf16 "
                         "products, RR_CE_F32_MFMA=1: the fp32-input instruction)")
    a = ap.parse_args()
    from review_recommender_amd import synth
    from review_recommender_amd.cross_encoder import CrossEncoder
    sd = synth.bert_state_dict(1, n_layers=a.layers, n_labels=1)
    ce = CrossEncoder(sd, precision=a.precision)
    if a.len:
        seqs = synth.token_pairs(a.pairs, 2, min_len=a.len, max_len=a.len)
    else:
        seqs = synth.token_pairs(a.pairs, 2, min_len=64, max_len=512)
    lens = np.array([len(s[0]) for s in seqs], dtype=np.float64)
    T = lens.sum()
    per_tok = 2.0 * 384 * (1152 + 384 + 2 * 1536)
    model_flops = a.layers * (T * per_tok + (4.0 * 384 * lens * lens).sum())      # what the reference's forward computes
    # executed: the last layer projects QKV for every token but attends / projects / normalises only the [CLS] row
    flops = (a.layers - 1) * (T * per_tok + (4.0 * 384 * lens * lens).sum()) + T * 2.0 * 384 * 1152 \
        + (4.0 * 384 * 16 * lens).sum() + len(seqs) * 2.0 * 384 * (384 + 2 * 1536)
    ce.predict_ids(seqs)
    ms = []
    for _ in range(a.reps):
        ce.predict_ids(seqs)
        ms.append(ce.model.last_forward_ms())
    t = float(np.median(ms)) * 1e-3
    # fp32 precision: every fp32 product is THREE fp16 MFMA products (csrc/rr_ce_h2.hip: operands as hi + lo / 2048), so the
    # matrix cores execute 3 x the model's FLOPs; the roofline below prices those against the dense fp16 peak (= the bf16 one).
    terms = 1 if a.precision == "bf16" else 3
    peak = 2500.0
    # attention is NOT on the MFMA roof: head dim 32 means one v_exp_f32 (a quarter-rate instruction: 4 lanes per cycle and
    # SIMD) per 64 MACs of QK^T + 64 of PV, and the fp32 mode adds the two-term split of every probability: its share of the
    # forward is vector-bound.  Stated here so the whole-forward MFMA fraction below is not read as the attention kernels' own roof.
    att_flops = a.layers * (4.0 * 384 * lens * lens).sum()
    exps = (a.layers - 1) * 12 * (lens * lens).sum() + 12 * (16 * lens).sum()
    bounds = {"gemm_kernels": "mfma (%s)" % ("bf16 matrix cores" if a.precision == "bf16" else "fp16 matrix cores: three fp16 term products per fp32 product"),
              "attention_kernels": {"bound": "vector (exp + probability split): one v_exp_f32 per score at 4 lanes per cycle and SIMD",
                                    "share_of_model_flops": round(att_flops / model_flops, 4),
                                    "exp_floor_ms_at_2.1GHz": round(exps / (256 * 4 * 4 * 2.1e9) * 1e3, 3)}}
    print(json.dumps({"precision": a.precision, "pairs": a.pairs, "tokens": int(T), "layers": a.layers, "forward_ms": round(t * 1e3, 3),
                      "pairs_per_s": round(a.pairs / t, 1), "tokens_per_s": round(T / t, 1),
                      "model_tflop": round(model_flops / 1e12, 4), "executed_tflop": round(flops / 1e12, 4),
                      "achieved_tflops": round(flops / t / 1e12, 2), "mfma_products_per_fp32_product": terms,
                      "roofline": {"bound": "mfma", "achieved": round(terms * flops / t / 1e12, 2), "peak": peak,
                                   "unit": "TFLOP/s", "frac": round(terms * flops / t / (peak * 1e12), 4)},
                      "bounds_by_kernel_family": bounds}))


if __name__ == "__main__":
    main()
