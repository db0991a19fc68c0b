"""Synthetic corpus of SURVEY section 8d built ON the GPU (10M rows x 384 fp32 = 15 GB never touch the
host): seeded blocks of unit rows + metadata + a BM25 corpus (Poisson(doc_len) tokens per document from
Zipf(1.07) over `vocab` terms), turned into the device-resident arrays `rr_index_adopt_device` and
`rr_bm25_create_dev` adopt.  bench.py times searches over exactly this build; tests/test_gpu_bench_config.py
checks the same build against the oracle."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

DIM = 384
SEED = 1234
N_BLOCKS = 8      # the corpus is generated in 8 seeded blocks so any world size in {1,2,4,8} sees the same data


def zipf_cdf(vocab: int, s: float = 1.07) -> np.ndarray:
    w = 1.0 / np.power(np.arange(1, vocab + 1, dtype=np.float64), s)
    return np.cumsum(w / w.sum())


def gen_block(torch, dev, block: int, rows: int, vocab: int, doc_len: int, cdf_dev, want_bm25: bool):
    """One seeded corpus block on the device: rows, metadata, forward BM25 entries."""
    g = torch.Generator(device=dev)
    g.manual_seed(SEED * 1000 + block)
    x = torch.randn((rows, DIM), generator=g, device=dev, dtype=torch.float32)
    n_rev = torch.clamp(torch.floor(torch.exp(torch.randn(rows, generator=g, device=dev, dtype=torch.float64)
                                              * 1.2 + 2.5)), 1, 5000)
    stars = torch.round(torch.clamp(torch.randn(rows, generator=g, device=dev, dtype=torch.float64) * 0.6 + 4.1,
                                    1.0, 5.0) * 1000) / 1000
    out = {"x": x, "n": n_rev, "stars": stars}
    if want_bm25:
        dl = torch.clamp(torch.poisson(torch.full((rows,), float(doc_len), device=dev), generator=g), min=1).long()
        total = int(dl.sum().item())
        u = torch.rand(total, generator=g, device=dev, dtype=torch.float64)
        tok = torch.clamp(torch.searchsorted(cdf_dev, u), max=vocab - 1)
        doc = torch.repeat_interleave(torch.arange(rows, device=dev), dl)
        key, _ = torch.sort(doc * vocab + tok)
        uniq, cnt = torch.unique_consecutive(key, return_counts=True)
        out.update(doc_len=dl.int(), e_doc=(uniq // vocab), e_term=(uniq % vocab).int(), e_tf=cnt.int())
    return out


@dataclass
class DeviceShard:
    sharded: object                 # ShardedSearcher over this rank's rows
    index: object                   # ProductIndex (adopted device matrix)
    matrix: object                  # torch tensor (n_local, 384) fp32 or bf16
    n_local: int
    row_lo: int
    bm25_arrays: Optional[dict] = None   # device tensors adopted by rr_bm25_create_dev (+ avgdl, idf)
    stats: dict = field(default_factory=dict)
    keep: List = field(default_factory=list)


def build_device_shard(torch, dist, *, docs: int, rank: int, world: int, dev, vocab: int = 200_000,
                       doc_len: int = 40, want_bm25: bool = True, dtype: str = "f32",
                       force_payload: bool = False) -> DeviceShard:
    from . import _lib
    from .bm25 import idf_with_floor
    from .engine import HybridSearcher
    from .index import ProductIndex
    from .sharded import ShardedSearcher, shard_bounds

    assert docs % N_BLOCKS == 0 and N_BLOCKS % world == 0, "docs %% 8 == 0 and gpus in {1,2,4,8}"
    per_block = docs // N_BLOCKS
    blocks = range(rank * N_BLOCKS // world, (rank + 1) * N_BLOCKS // world)
    lo, hi = shard_bounds(docs, world, rank)
    n_local = hi - lo
    cdf_dev = torch.from_numpy(zipf_cdf(vocab)).to(dev) if want_bm25 else None

    mat = torch.empty((n_local, DIM), device=dev, dtype=torch.float32)
    n_rev = torch.empty(n_local, device=dev, dtype=torch.float64)
    stars = torch.empty(n_local, device=dev, dtype=torch.float64)
    parts = []
    for j, b in enumerate(blocks):
        blk = gen_block(torch, dev, b, per_block, vocab, doc_len, cdf_dev, want_bm25)
        s = j * per_block
        mat[s:s + per_block] = blk["x"]
        n_rev[s:s + per_block] = blk["n"]
        stars[s:s + per_block] = blk["stars"]
        if want_bm25:
            parts.append((blk["doc_len"], blk["e_doc"] + s, blk["e_term"], blk["e_tf"]))
        del blk
    if dtype == "bf16":
        # normalise in fp32, round once to bf16 (nearest even): SURVEY section 8d
        mat = (mat / torch.clamp(mat.norm(dim=1, keepdim=True), min=1e-12)).to(torch.bfloat16).contiguous()
        index = ProductIndex(None, n_rows=n_local, dim=DIM, device=dev.index, row_offset=lo,
                             device_ptr=mat.data_ptr(), keepalive=mat, dtype="bf16")
    else:
        index = ProductIndex(None, n_rows=n_local, dim=DIM, device=dev.index, row_offset=lo,
                             device_ptr=mat.data_ptr(), keepalive=mat)
        index.l2_normalize()                          # utils.py:40-44 on the device
    index.set_meta(n_rev.cpu().numpy(), stars.cpu().numpy())

    bm25 = None
    arrays = None
    keep = [mat]
    stats = dict(nnz=0, avgdl=0.0, df=None)
    if want_bm25:
        dl = torch.cat([p[0] for p in parts])
        e_doc = torch.cat([p[1] for p in parts])
        e_term = torch.cat([p[2] for p in parts]).contiguous()
        e_tf = torch.cat([p[3] for p in parts]).contiguous()
        del parts
        nnz = int(e_doc.numel())
        doc_indptr = torch.zeros(n_local + 1, dtype=torch.int64, device=dev)
        doc_indptr[1:] = torch.cumsum(torch.bincount(e_doc, minlength=n_local), 0)
        # postings: entries sorted by (term, doc)
        pkey, perm = torch.sort(e_term.long() * n_local + e_doc)
        post_docs = (pkey % n_local).int()
        post_tf = e_tf[perm].contiguous()
        df_local = torch.bincount(e_term.long(), minlength=vocab)
        post_indptr = torch.zeros(vocab + 1, dtype=torch.int64, device=dev)
        post_indptr[1:] = torch.cumsum(df_local, 0)
        del pkey, perm, e_doc
        # corpus-wide statistics (setup-time collectives; not on the query path)
        df = df_local.clone()
        tot_len = dl.long().sum().reshape(1)
        if world > 1:
            if dist.get_backend() == "nccl":
                dist.all_reduce(df)
                dist.all_reduce(tot_len)
            else:                                   # gloo rehearsal: reduce on the host
                df_h, tl_h = df.cpu(), tot_len.cpu()
                dist.all_reduce(df_h)
                dist.all_reduce(tl_h)
                df, tot_len = df_h.to(dev), tl_h.to(dev)
        avgdl = int(tot_len.item()) / docs
        idf = torch.from_numpy(idf_with_floor(df.cpu().numpy(), docs)).to(dev)
        h = C.c_void_p()
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(_lib.load().rr_bm25_create_dev(
            dev.index, n_local, vocab, nnz, p(post_indptr), p(post_docs), p(post_tf), p(doc_indptr),
            p(e_term), p(e_tf), p(dl), p(idf), avgdl, 1.5, 0.75, lo, C.byref(h)), "rr_bm25_create_dev")

        class _Adopted:                                # minimal BM25Index look-alike over adopted arrays
            handle = h
        bm25 = _Adopted()
        arrays = dict(post_indptr=post_indptr, post_docs=post_docs, post_tf=post_tf, doc_indptr=doc_indptr,
                      doc_terms=e_term, doc_tf=e_tf, doc_len=dl, idf=idf, avgdl=avgdl)
        keep += [post_indptr, post_docs, post_tf, doc_indptr, e_term, e_tf, dl, idf]
        stats = dict(nnz=nnz, avgdl=avgdl, df=df.cpu().numpy())
    searcher = HybridSearcher(index, bm25)
    sharded = ShardedSearcher(searcher, docs, rank, world)
    sharded.force_payload = force_payload
    return DeviceShard(sharded, index, mat, n_local, lo, arrays, stats, keep)
