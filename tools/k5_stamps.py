#!/usr/bin/env python3
"""Where a ce_ffn_fused workgroup spends its cycles: in-kernel phase clocks (librr_hip_dbg.so) of the fused-FFN launch
of a 256 x 512-token bf16 forward.  Prints, for workgroups 0 and 600, shader cycles per 32-feature chunk and phase."""
import ctypes as C
import os
import sys

os.environ["RR_DEBUG_HARNESS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from review_recommender_amd import _lib, synth
    from review_recommender_amd.cross_encoder import CrossEncoder
    lib = _lib.load()
    ce = CrossEncoder(synth.bert_state_dict(1, n_layers=2, n_labels=1), precision="bf16")   # layer 0 runs the full-size FFN
    seqs = synth.token_pairs(256, 2, min_len=512, max_len=512)
    for _ in range(3):
        ce.predict_ids(seqs)
    print(f"forward (2 layers) {ce.model.last_forward_ms():.3f} ms")
    out = (C.c_uint64 * 20)()
    _lib.check(lib.rr_debug_ce_ffn_stamps(out), "rr_debug_ce_ffn_stamps")
    names = ["prologue", "iteration top", "slots 0-23 (second product)", "slots 24-47 (first product)", "barrier", "epilogue", "-", "-", "-"]
    for w in range(2):
        v = [out[10 * w + i] for i in range(10)]
        tot = sum(v[:9])
        us = v[9] / 100.0
        print(f"workgroup {(0, 600)[w]}: {tot} cycles in {us:.1f} us = {tot / us / 1e3:.2f} GHz; per iteration (49): "
              + ", ".join(f"{n} {v[i] / 49:.0f}" for i, n in enumerate(names) if 1 <= i <= 4)
              + f"; prologue {v[0]}, epilogue {v[5]}")


if __name__ == "__main__":
    main()
