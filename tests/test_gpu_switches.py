"""The A/B switches of the batched dense path are read once per process: every one of them
must give the same exact answer.  Each switch runs the same seeded search in a child process; the parent compares the
answers with its own (default switches).  2.2M rows: above the store prefilter's 2M-row floor."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_ROWS, BATCH, POOL = 2_200_000, 256, 150

CHILD = r"""
import sys, json, hashlib
import numpy as np, torch
sys.path.insert(0, %r)
from review_recommender_amd.index import ProductIndex
n, b, pool = %d, %d, %d
g = torch.Generator(device="cuda"); g.manual_seed(11)
m = torch.randn((n, 384), device="cuda", generator=g); m /= m.norm(dim=1, keepdim=True)
ix = ProductIndex(None, n_rows=n, dim=384, device_ptr=m.data_ptr(), keepalive=m)
q = np.random.default_rng(12).standard_normal((b, 384)).astype(np.float32)
q /= np.linalg.norm(q, axis=1, keepdims=True)
out = {}
for nq in (b, 200, 130, 64):
    rows, sims = ix.dense_topk(q[:nq], pool)
    out[str(nq)] = [hashlib.sha256(np.ascontiguousarray(rows).tobytes()).hexdigest(),
                    hashlib.sha256(np.ascontiguousarray(sims).tobytes()).hexdigest()]
print("RESULT " + json.dumps(out))
""" % (ROOT, N_ROWS, BATCH, POOL)


def run_child(extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.fixture(scope="module")
def default_answer():
    return run_child({})


@pytest.mark.parametrize("switch", ["RR_FLTQ_NOASM", "RR_NO_FLTQ", "RR_NO_DUAL", "RR_NO_COUPLE", "RR_DUAL_PREFILTER", "RR_NO_PAIR",
                                    "RR_NO_PREFILTER", "RR_NO_SHADOW", "RR_NO_RESCORE_PLANE"])
def test_switch_gives_the_default_answer(default_answer, switch):
    """rows AND scores bitwise: every filter path rescores its candidates with the single-query chain.  (RR_SCAN_EXACT
    serves with split-operand arithmetic, equal to fp32 rounding: test_exact_scan_switch_differs_only_inside_the_tie_band.)"""
    env = {switch: "1"}
    if switch in ("RR_NO_COUPLE", "RR_DUAL_PREFILTER"):
        env["RR_NO_FLTQ"] = "1"                       # these two act on the rr_scan_flt16 form of the two-set launch
    got = run_child(env)
    for nq, (rows_h, sims_h) in default_answer.items():
        assert got[nq][0] == rows_h and got[nq][1] == sims_h, (switch, nq)


def test_hand_scheduled_loop_writes_the_tile_words_of_the_cpp_bodies():
    """rr_scan_fltq's steady-state loop is generated assembly (csrc/gen_fltq_loop.py); the C++ bodies it replaces stay in
    the kernel for the ends of a run.  Both forms must leave the SAME tile words and group maxima, bit for bit, for both
    query sets -- checked word by word through the ablation harness's library (librr_hip_dbg.so), in a child process."""
    from review_recommender_amd.build import DEBUG_LIB_PATH
    if not DEBUG_LIB_PATH.exists():
        pytest.skip("librr_hip_dbg.so not built (python review-recommender_amd/build.py --debug)")
    child = r"""
import os, sys
os.environ["RR_DEBUG_HARNESS"] = "1"
sys.path.insert(0, %r)
import ctypes as C
import numpy as np, torch
from review_recommender_amd import _lib
from review_recommender_amd.index import ProductIndex
lib = _lib.load()
for n in (2_200_000, 1_000_003):
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    m = torch.randn((n, 384), device="cuda", generator=g); m /= m.norm(dim=1, keepdim=True)
    ix = ProductIndex(None, n_rows=n, dim=384, device_ptr=m.data_ptr(), keepalive=m)
    ix.dense_topk(np.random.default_rng(6).standard_normal((256, 384)).astype(np.float32), 150)
    out = (C.c_int64 * 8)()
    _lib.check(lib.rr_debug_fltq_compare(ix.handle, out), "rr_debug_fltq_compare")
    print("COMPARE", n, out[0], out[1], out[2], out[3])
""" % ROOT
    p = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l.split() for l in p.stdout.splitlines() if l.startswith("COMPARE ")]
    assert len(lines) == 2
    for _, n, dw, dg, words, groups in lines:
        assert int(words) > 0 and int(groups) > 0
        assert int(dw) == 0 and int(dg) == 0, (n, dw, dg)


TIE_CHILD = r"""
import sys
import numpy as np, torch
sys.path.insert(0, %r)
from review_recommender_amd.index import ProductIndex
n, b, pool = %d, %d, %d
g = torch.Generator(device="cuda"); g.manual_seed(11)
m = torch.randn((n, 384), device="cuda", generator=g); m /= m.norm(dim=1, keepdim=True)
ix = ProductIndex(None, n_rows=n, dim=384, device_ptr=m.data_ptr(), keepalive=m)
q = np.random.default_rng(12).standard_normal((b, 384)).astype(np.float32)
q /= np.linalg.norm(q, axis=1, keepdims=True)
rows, sims = ix.dense_topk(q, pool)
# float64 scores of the returned rows (the matrix is regenerated from the seed in both children: same bits)
r = torch.from_numpy(rows).cuda()
s64 = torch.einsum("bpd,bd->bp", m[r].double(), torch.from_numpy(q).cuda().double()).cpu().numpy()
np.savez(sys.argv[1], rows=rows, sims=sims, s64=s64)
""" % (ROOT, N_ROWS, BATCH, POOL)


def test_exact_scan_switch_differs_only_inside_the_tie_band(tmp_path):
    """RR_SCAN_EXACT=1 serves a batch with the split-operand scans (fp32-rounding-equal scores, not the per-row chain):
    its answer may differ from the default's only where float64 cannot tell the rows apart either -- positions whose
    float64 scores lie within 4e-7 of each other (the parity bar's tie band), scores within 1e-6."""
    outs = []
    for name, env in (("default", {}), ("exact", {"RR_SCAN_EXACT": "1"})):
        f = str(tmp_path / (name + ".npz"))
        e = dict(os.environ)
        e.update(env)
        p = subprocess.run([sys.executable, "-c", TIE_CHILD, f], env=e, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(np.load(f))
    d, x = outs
    assert np.abs(d["sims"] - x["sims"]).max() < 1e-6
    swapped = 0
    for i in range(BATCH):
        diff = np.nonzero(d["rows"][i] != x["rows"][i])[0]
        swapped += len(diff)
        for j in diff:                                        # the two rows at a differing position are a float64 near-tie
            assert abs(d["s64"][i][j] - x["s64"][i][j]) < 4e-7, (i, j, d["s64"][i][j], x["s64"][i][j])
        # and both lists are ordered by float64 score up to that band
        for o in (d, x):
            assert np.all(np.diff(o["s64"][i]) < 4e-7), i
    print(f"RR_SCAN_EXACT: {swapped} of {BATCH * POOL} positions differ, all inside the 4e-7 band")


FLAGGED_CHILD = r"""
import sys, json, hashlib
import numpy as np
sys.path.insert(0, %r)
from review_recommender_amd import synth
from review_recommender_amd.index import ProductIndex
V = synth.unit_rows(200_000, 384, 91)
V[::10] = V[3]                                   # 20000 copies of row 3: a query equal to it overflows its candidate list
Q = synth.unit_rows(200, 384, 92)
flagged = [0, 1, 2, 3, 4, 5, 70, 71, 72, 73, 74, 75, 199]   # more than eight: all of them go to the split-operand pass;
Q[flagged] = V[3]                                # flagged queries in three of the four 64-query blocks
ix = ProductIndex(V)
rows, sims = ix.dense_topk(Q, 150)
for q in flagged:
    assert rows[q].tolist() == sorted({3} | set(range(0, 1500, 10)))[:150]
print("RESULT " + json.dumps([hashlib.sha256(rows.tobytes()).hexdigest(), hashlib.sha256(sims.tobytes()).hexdigest()]))
""" % ROOT


def test_one_slice_of_score_scratch_serves_the_flagged_queries_block_by_block():
    """An index too large for one score slice per 64 queries of a call keeps ONE slice and sends the flagged queries
    through the exact scan block by block (rr_ensure_scratch / rr_dense_x3w_fallback_all); RR_SCRATCH_SMALL=1 forces that
    on a small index.  Same rows and scores, bit for bit, as the sliced two-launch fallback."""
    got = []
    for env in ({}, {"RR_SCRATCH_SMALL": "1"}):
        p = subprocess.run([sys.executable, "-c", FLAGGED_CHILD], env=dict(os.environ, **env), capture_output=True, text=True,
                           timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        got.append([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1])
    assert got[0] == got[1]
