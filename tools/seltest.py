import sys, time; sys.path.insert(0,'.')
import numpy as np
from review_recommender_amd import synth
from review_recommender_amd.index import ProductIndex
for n in (100_000, 1_000_000, 3_000_000):
    V = synth.unit_rows(n, 384, 1); q = synth.unit_rows(1,384,2)
    ix = ProductIndex(V)
    for pool in (150, 10):
        ix.dense_topk(q, pool); print(n, pool, ix.select_trace(), flush=True)
    ix.close()
