"""The register-ring contract of the shipped scan kernels, checked statically on the compiler's own assembly (VERDICT r2
item 5).  The batched scans load matrix rows into VGPRs with inline-asm `global_load_dwordx4` and wait with hand-counted
`s_waitcnt vmcnt(N)`; nothing may touch a destination register while its load is in flight -- the compiler cannot know.
tools/check_ring_hazards.py walks every path of a kernel's instructions with the hardware's in-order vector-memory queue.
Round 2's faulting ablation (`rr_scan_flt<4, true, 1>`, commit 603cbc7) is flagged by it: the compiler had put a ring
destination on top of its own address pair (`global_load_dwordx4 v[208:211], v[208:209]`) and recomputed the next row pointer
into v[208:209] (`v_mad_u64_u32 v[208:209], ...`) while that load was on its way -- the landing data overwrote the pointer
(profiles/r03_ring_hazard_of_the_r2_fault.txt).  Every kernel the product launches must come out clean."""
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT))
import check_ring_hazards as H                                     # noqa: E402
from review_recommender_amd import build as B                      # noqa: E402


def test_checker_flags_a_touched_ring_register_and_accepts_a_counted_wait():
    bad = """
k_bad:
	global_load_dwordx4 v[8:11], v[8:9], off
	global_load_dwordx4 v[12:15], v[2:3], off
	s_waitcnt vmcnt(1)
	v_add_f32_e32 v0, v8, v9
	v_mad_u64_u32 v[12:13], s[0:1], v4, s14, v[20:21]
	s_waitcnt vmcnt(0)
	s_endpgm
.Lfunc_end0:
"""
    name, loads, shown, total = H.check_kernel(bad, "k_bad")
    assert loads == 2 and total == 1 and shown[0][1].startswith("v_mad_u64_u32") and ("v", 12) in shown[0][4]
    good = bad.replace("v_mad_u64_u32 v[12:13]", "v_mad_u64_u32 v[16:17]")
    assert H.check_kernel(good, "k_bad")[3] == 0
    # a loop: the load of the body's end is covered by the wait at its top on the way round
    loop = """
k_loop:
	global_load_dwordx4 v[8:11], v[2:3], off
.LBB0_1:
	s_waitcnt vmcnt(0)
	v_add_f32_e32 v0, v8, v9
	global_load_dwordx4 v[8:11], v[2:3], off
	global_store_dword v[4:5], v0, off
	s_cbranch_scc1 .LBB0_1
	s_waitcnt vmcnt(0)
	s_endpgm
.Lfunc_end1:
"""
    assert H.check_kernel(loop, "k_loop")[3] == 0
    assert H.check_kernel(loop.replace("\ts_waitcnt vmcnt(0)\n\tv_add", "\ts_waitcnt vmcnt(2)\n\tv_add"), "k_loop")[3] > 0


@pytest.fixture(scope="module")
def scan_assembly(tmp_path_factory):
    """source -> device assembly, for the product build and (key "debug:...") for the ablation harness's build of the filter
    scans (-DRR_DEBUG_HARNESS + the generated timing ablations of the hand-scheduled loop): tools/ launch those on the GPU."""
    out = tmp_path_factory.mktemp("asm")
    jobs = [("rr_dense_flt.hip", False), ("rr_dense_x3w.hip", False), ("rr_dense_x3.hip", False), ("rr_dense_flt.hip", True)]
    B.check_generated(debug=True)

    def one(job):
        src, debug = job
        dst = out / (("dbg_" if debug else "") + src + ".s")
        flags = B.FLAGS + (B.DEBUG_FLAGS if debug else [])
        cmd = [B.hipcc_path(), *flags, "--cuda-device-only", "-S", str(B.CSRC / src), "-o", str(dst)]
        p = subprocess.run(cmd, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-2000:]
        return dst.read_text()
    with ThreadPoolExecutor(4) as ex:
        return dict(zip([("debug:" if d else "") + s for s, d in jobs], ex.map(one, jobs)))


def test_no_shipped_scan_touches_a_ring_register_before_its_counted_wait(scan_assembly):
    checked = 0
    for src, asm in scan_assembly.items():
        syms = sorted(set(re.findall(r"^(_Z\d+rr_scan_(?:flt|flt16|fltq|x3w|mfma_x3)I\w+):", asm, flags=re.M)))
        assert syms, src
        for sym in syms:
            name, loads, shown, total = H.check_kernel(asm, sym)
            assert total == 0, (name, shown[:3])
            checked += 1 if loads else 0
    assert checked >= 20          # every instantiation with register loads was walked
    # the harness build holds the product's instantiations plus the ablated ones: all of them were walked
    dbg = set(re.findall(r"^(_Z\d+rr_scan_(?:flt|flt16|fltq)I\w+):", scan_assembly["debug:rr_dense_flt.hip"], flags=re.M))
    prod = set(re.findall(r"^(_Z\d+rr_scan_(?:flt|flt16|fltq)I\w+):", scan_assembly["rr_dense_flt.hip"], flags=re.M))
    assert prod <= dbg and len(dbg) > len(prod)


def test_every_variant_a_tool_names_is_a_case_of_the_harness_switch():
    """tools/flt_ablate.py and tools/fltq_ablate.py pass their variant numbers to rr_debug_scan_flt: a number that is not a
    case of its switch stops the tool with "unknown ablation"  -- and a case whose kernel breaks the ring
    contract must not come back through a tool's default list."""
    src = (B.CSRC / "rr_dense_flt.hip").read_text()
    body = src[src.index('extern "C" int rr_debug_scan_flt('):]
    body = body[:body.index("#endif  // RR_DEBUG_HARNESS")]
    cases = {int(x) for x in re.findall(r"^\s*case (\d+):", body, flags=re.M)}
    cases |= {3000 + int(x) for x in re.findall(r"RR_FLTQA_CASE\((\d+)\)", body.split("#undef RR_FLTQA_CASE")[0].split("#define RR_FLTQA_CASE")[1])}
    assert {0, 64, 128, 3000, 3128} <= cases and not ({15, 31} & cases)
    assert {3000 + int(x) for x in B.FLTQ_ABLATIONS.split(",")} <= cases
    flt = (ROOT / "tools" / "flt_ablate.py").read_text()
    names = {int(x) for x in re.findall(r"[{ ,](\d+): \"", flt[flt.index("names = {"):flt.index("only = ")])}
    assert names and names <= cases, sorted(names - cases)
    fq = (ROOT / "tools" / "fltq_ablate.py").read_text()
    default = [int(x) for x in re.search(r"else \[([\d, ]+)\]", fq).group(1).split(",")]
    assert default and set(default) <= cases, sorted(set(default) - cases)
