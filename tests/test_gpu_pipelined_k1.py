"""The pipelined K1 (include/rr_hip.h: rr_dense_scan_slot_dev / rr_dense_select_part_dev, CU-masked streams): a batch
scanned into a slot on one stream and selected from it on another -- in one piece or in its three parts on alternating
streams, with other batches' scans in between -- returns bit for bit what rr_dense_topk_dev returns for it."""
import ctypes as C

import numpy as np
import pytest
import torch

from review_recommender_amd import _lib, synth
from review_recommender_amd.engine import HybridSearcher
from review_recommender_amd.index import ProductIndex

pytestmark = pytest.mark.gpu
N, POOL = 640_000, 150


@pytest.fixture(scope="module")
def world():
    V = synth.unit_rows(N, 384, 71)
    V[5000] = V[600_000]                                  # an exact tie
    ix = ProductIndex(V)
    n_rev, stars = synth.metadata(N, 72)
    ix.set_meta(n_rev.astype(np.float64), stars)
    s = HybridSearcher(ix)
    Q = [torch.from_numpy(synth.unit_rows(b, 384, 80 + i)).cuda() for i, b in enumerate((256, 256, 200, 37, 256, 128))]
    Q[0][0] = torch.from_numpy(V[5000]).cuda()
    want = [tuple(t.clone() for t in s.dense_pool(q, POOL)) for q in Q]
    torch.cuda.synchronize()
    yield s, Q, want
    ix.close()


def masked_stream(lib, first, n):
    h = C.c_void_p()
    _lib.check(lib.rr_stream_create_cu_range(0, first, n, C.byref(h)), "rr_stream_create_cu_range")
    return torch.cuda.ExternalStream(h.value, device=torch.device("cuda", 0)), h


@pytest.mark.parametrize("scan_cus", [0, 224, 128])
@pytest.mark.parametrize("split", [False, True])
def test_scan_on_one_stream_selection_on_another_is_bitwise_the_straight_call(world, scan_cus, split):
    s, Q, want = world
    lib = s.lib
    total = torch.cuda.get_device_properties(0).multi_processor_count
    if scan_cus:
        A, hA = masked_stream(lib, 0, scan_cus)
        T, hT = masked_stream(lib, scan_cus, total - scan_cus)
    else:
        A, T, hA, hT = torch.cuda.Stream(), torch.cuda.Stream(), None, None
    _lib.check(lib.rr_index_set_scan_cus(s.index.handle, scan_cus))
    try:
        got = {}
        n = len(Q)
        with torch.cuda.stream(A):
            assert s.dense_scan_slot(0, Q[0], POOL) is True
        for i in range(n):
            if i + 1 < n:
                with torch.cuda.stream(A):                 # the next batch's scan goes first: it runs beside this batch's selection
                    assert s.dense_scan_slot((i + 1) % 3, Q[i + 1], POOL) is True
            B = Q[i].shape[0]
            if split:
                with torch.cuda.stream(T):
                    assert s.dense_select_slot(i % 3, B, POOL, parts=s.SELECT_LIST) is None
                with torch.cuda.stream(A):
                    assert s.dense_select_slot(i % 3, B, POOL, parts=s.SELECT_RESCORE) is None
                with torch.cuda.stream(T):
                    got[i] = s.dense_select_slot(i % 3, B, POOL, parts=s.SELECT_ORDER)
            else:
                with torch.cuda.stream(T):
                    got[i] = s.dense_select_slot(i % 3, B, POOL)
        torch.cuda.synchronize()
        for i in range(n):
            assert torch.equal(got[i][0], want[i][0]) and torch.equal(got[i][1], want[i][1]), i
        # a plain call afterwards (slot 0, any stream) still answers the same, and voids nothing it should not
        rows, dense = s.dense_pool(Q[3], POOL)
        assert torch.equal(rows, want[3][0]) and torch.equal(dense, want[3][1])
    finally:
        _lib.check(lib.rr_index_set_scan_cus(s.index.handle, 0))
        torch.cuda.synchronize()
        for h in (hA, hT):
            if h is not None:
                _lib.check(lib.rr_stream_destroy(h))


def test_whole_k1_in_a_slot_of_its_own_leaves_the_scans_parked_in_the_others(world):
    """rr_dense_topk_slot_dev: a batch the pipeline cannot split (here 3 queries) runs its whole K1 in slot 2 while slots 0
    and 1 hold parked scans -- both are still selectable afterwards, bit for bit."""
    s, Q, want = world
    assert s.dense_scan_slot(0, Q[0], POOL) is True and s.dense_scan_slot(1, Q[1], POOL) is True
    small = Q[3][:3].contiguous()
    rows3, dense3 = s.dense_pool(small, POOL, slot=2)
    ref3 = s.dense_pool(Q[3], POOL, slot=2)                 # (also in slot 2: the full batch of 37 for the reference rows)
    assert torch.equal(rows3, ref3[0][:3]) and torch.equal(dense3, ref3[1][:3])
    for slot in (1, 0):
        rows, dense = s.dense_select_slot(slot, Q[slot].shape[0], POOL)
        assert torch.equal(rows, want[slot][0]) and torch.equal(dense, want[slot][1]), slot
    # ... whereas the same call in slot 0 voids what was parked THERE (and only there)
    assert s.dense_scan_slot(0, Q[0], POOL) is True and s.dense_scan_slot(1, Q[1], POOL) is True
    s.dense_pool(small, POOL)
    with pytest.raises(ValueError, match="no scan of these"):
        s.dense_select_slot(0, 256, POOL)
    rows, dense = s.dense_select_slot(1, 256, POOL)
    assert torch.equal(rows, want[1][0])


def test_a_selection_without_its_scan_is_refused_and_small_calls_are_declined(world):
    s, Q, _ = world
    with pytest.raises(ValueError, match="no scan of these"):
        s.dense_select_slot(2, 256, POOL)                  # nothing is parked in slot 2
    assert s.dense_scan_slot(1, Q[3][:3], POOL) is None    # 3 queries: the VALU scans serve it (rr_dense_topk_dev)
    with pytest.raises(ValueError, match="scan slot"):
        s.dense_scan_slot(3, Q[0], POOL)
    with pytest.raises(ValueError, match="CUs"):
        _lib.check(s.lib.rr_stream_create_cu_range(0, 250, 16, C.byref(C.c_void_p())), "rr_stream_create_cu_range")
    # a scan parked in a slot is voided by a wrong-sized selection request, not silently served
    assert s.dense_scan_slot(1, Q[3], POOL) is True
    with pytest.raises(ValueError, match="no scan of these"):
        s.dense_select_slot(1, Q[3].shape[0] + 1, POOL)
