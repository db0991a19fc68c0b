"""Times the BM25 kernels alone (dev tool): python tools/bm25_time.py <docs> [vocab]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from review_recommender_amd import synth
from review_recommender_amd.bm25 import BM25Corpus
n = int(sys.argv[1]); vocab = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
ip, terms, tf, dl = synth.bm25_forward_csr(n, vocab, 40, 17)
host = BM25Corpus(ip, terms, tf, dl, vocab)
dev = host.to_device()
df = np.bincount(terms, minlength=vocab)
qs = synth.query_terms(8, vocab, 5, df)
for q in qs[:4]:
    postings = int(df[q].sum())
    dev.get_scores_ids(q)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): dev.get_scores_ids(q)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    # algorithmic bytes: postings x (4 doc + 4 tf + 4 doc_len gather) + 8 B per document written
    alg = postings * 12 + n * 8
    print(f"docs {n} terms {len(q)} postings {postings}: get_scores {dt*1e3:.3f} ms incl. D2H of {n*8/1e6:.0f} MB "
          f"(algorithmic {alg/1e6:.1f} MB)", flush=True)
rows = np.random.default_rng(1).integers(0, n, size=(8, 150)).astype(np.int64)
for mode in ("forward", "postings"):
    dev.scores_at_ids(qs, rows, mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dev.scores_at_ids(qs, rows, mode)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"scores_at 8 x 150 ({mode}): {dt*1e3:.3f} ms per call incl. host copies", flush=True)
