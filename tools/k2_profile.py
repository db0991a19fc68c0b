"""K2 under the profiler: the BM25 kernels on bench.py's own 10M-document corpus (327 M postings).
  * rr_bm25_slices  -- BM25Okapi.get_scores(tokens): all N documents of a 4-term query (app/app_product_search.py:206)
  * rr_bm25_at<0|1> -- the hybrid path: BM25 at 256 x 150 candidate rows (forward lists | postings lists)
Run plain to print the per-call algorithmic bytes (sum of df x 12 B + 8 B x N for get_scores); run under
`rocprofv3 --kernel-trace --stats` / `--pmc FETCH_SIZE` for the kernel times / memory-side bytes (tools/k2_summary.py).
    python tools/k2_profile.py [docs] [out.json]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import json
import numpy as np, torch
from review_recommender_amd import _lib, synth
from review_recommender_amd.device_corpus import build_device_shard

docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
out_path = sys.argv[2] if len(sys.argv) > 2 else None
VOCAB, B, POOL, REPS = 200_000, 256, 150, 6
dev = torch.device("cuda", 0)
sh = build_device_shard(torch, None, docs=docs, rank=0, world=1, dev=dev, vocab=VOCAB, doc_len=40)
df = sh.stats["df"]
lib = _lib.load()
bm = sh.sharded.s.bm25
rng = np.random.default_rng(5)
queries = synth.query_terms(64, VOCAB, 99, df)
four = [q[:4] for q in queries if len(q) >= 4][:REPS]
scores = np.empty(docs, dtype=np.float64)
calls = []
for q in four:
    ids = np.ascontiguousarray(q, dtype=np.int32)
    _lib.check(lib.rr_bm25_get_scores(bm.handle, _lib.ptr(ids), len(ids), _lib.ptr(scores)), "rr_bm25_get_scores")
    sdf = int(df[ids].sum())
    calls.append({"terms": ids.tolist(), "sum_df": sdf, "algorithmic_bytes": sdf * 12 + 8 * docs,
                  "nonzero_scores": int(np.count_nonzero(scores))})
terms = synth.query_terms(B, VOCAB, 7, df)
rows = torch.from_numpy(rng.integers(0, docs, (B, POOL)).astype(np.int64)).cuda()
s = sh.sharded.s
n_tok = int(sum(len(t) for t in terms))
for mode in ("forward", "postings"):
    for _ in range(REPS):
        s.bm25_at(terms, rows, mode)
torch.cuda.synchronize()
res = {"docs": docs, "postings": int(sh.stats["nnz"]), "get_scores_calls": calls,
       "scores_at": {"queries": B, "pool": POOL, "query_tokens": n_tok,
                     # candidate-side: (candidate, token) pairs x ~log2(40) probes of 4 B in the candidate's forward list
                     "algorithmic_bytes_forward": int(B * POOL * (n_tok / B) * np.log2(40) * 4)}}
print(json.dumps(res))
if out_path:
    json.dump(res, open(out_path, "w"))
