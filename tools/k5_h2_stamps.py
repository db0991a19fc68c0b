#!/usr/bin/env python3
"""Where a ce_gemm_h2 workgroup (fp32-mode FFN1: K = 384, N = 1536, GELU + h2 epilogue) spends its cycles: in-kernel phase
clocks (librr_hip_dbg.so) of waves 0 and 4 of workgroup 2048 in a 256 x 512-token forward.
    python tools/k5_h2_stamps.py            (RR_CE_H2_STAGE=regs | RR_CE_H2_NO_STAGGER=1 for the variants)"""
import ctypes as C
import os
import sys

os.environ["RR_DEBUG_HARNESS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from review_recommender_amd import _lib, synth
    from review_recommender_amd.cross_encoder import CrossEncoder
    lib = _lib.load()
    ce = CrossEncoder(synth.bert_state_dict(1, n_layers=2, n_labels=1), precision="fp32")   # layer 0 runs the full-size FFN
    seqs = synth.token_pairs(256, 2, min_len=512, max_len=512)
    for _ in range(3):
        ce.model.forward_ids(seqs, 2)                      # hidden states: every layer at full size
    print(f"forward (2 layers, all tokens) {ce.model.last_forward_ms():.3f} ms")
    out = (C.c_uint64 * 16)()
    _lib.check(lib.rr_debug_ce_h2_stamps(out), "rr_debug_ce_h2_stamps")
    names = ["prologue", "wait + barrier", "data movement", "reads -> first MFMA", "MFMA issue", "epilogue", "steps", "wall (10 ns)"]
    for w in range(2):
        v = [out[8 * w + i] for i in range(8)]
        steps = max(v[6], 1)
        cyc = v[0] + v[1] + v[2] + v[3] + v[4] + v[5]
        print(f"wave {4 * w}: prologue {v[0]}, per step ({steps}): wait+barrier {v[1] / steps:.0f}, move {v[2] / steps:.0f}, reads {v[3] / steps:.0f}, "
              f"mfma {v[4] / steps:.0f}; epilogue {v[5]}; {cyc} stamped shader cycles in {v[7] / 100.0:.2f} us")


if __name__ == "__main__":
    main()
