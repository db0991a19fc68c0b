#!/usr/bin/env python3
"""Generates csrc/rr_fltq_loop.inc: the hand-scheduled steady-state loop of rr_scan_fltq (gfx950), as ONE asm statement.

Why a generator: the C++ form of the loop (one asm statement per instruction, lambdas, sched_barrier) let hipcc put
`s_nop`s between dependent asm statements, 64-bit address arithmetic and v_readfirstlane chains in front of every
LDS-DMA piece and branches around the epilogue stores -- 2 350 cycles per M-tile for 1 536 cycles of MFMA (r02 stamps).
Here every instruction of four M-tile bodies (ring of four LDS images x two accumulator sets) is placed by hand:

  per M-tile and wave: 96 v_mfma_f32_16x16x32_bf16 (the chip holds a higher clock on this shape than on 32x32x16 at the
  same cycles per FLOP: r03 ablations, MI355X_MICROARCH.md "DVFS give-back" (7)), 24 ds_read_b128 (A operands = 16 rows x 32
  dims, four operands ahead, counted lgkmcnt), 6 LDS-DMA pieces `buffer_load_dwordx4 ... lds` (scalar base, scalar piece
  offset and ONE per-lane offset register: no vector address arithmetic; the resource's num_records makes reads past the
  matrix return zeros), the epilogue of the PREVIOUS M-tile (35 vector instructions per 32-query fragment pair, the
  arithmetic of `piece()` in rr_scan_fltq, bit for bit) spread over the MFMA shadows at <= CAPS[n] instructions behind
  the MFMA of fragment n, 2 tile-word stores (buffer_store_dword: the upper lane half carries an out-of-range offset and
  is dropped by the bounds check).

Hazards kept by construction (wait states = instructions in between):
  VALU write -> v_permlane16/32_swap of that register: 2;  v_cmp (VALU) write of an SGPR pair -> VALU read of it: 2;
  s_add m0 -> LDS-DMA: 1;  MFMA write of an accumulator -> VALU read: the previous body's last MFMAs are > 60 cycles back.

Usage: python gen_fltq_loop.py > rr_fltq_loop.inc   (build.py checks that the committed .inc is up to date)
       python gen_fltq_loop.py --abl 1,2,3,... > rr_fltq_loop_abl.inc   (debug harness only: timing ablations of the loop,
       selected by `if constexpr (ABL == n)`; bits: 1 no LDS-DMA pieces, 2 no epilogue instructions / stores, 4 no vmcnt wait +
       barrier, 8 no A reads (stale operands), 64 pieces re-read the same M-tile (cache hits), 512 no epilogue instructions
       in the restart bubble at the top of an M-tile (all of them in MFMA shadows); 128 = nothing ablated)
"""
import sys

ABL = 0                # ablation bits of the body being generated (0 = the product's loop)

NB = 4                 # ring of LDS images (M-tiles); the loop is unrolled over it
AD = 4                 # A operands (16 rows x 32 dims: four MFMAs each) requested this many operands ahead
TILE_BYTES = 32 * 768  # one 32-row M-tile image
PIECES = 6             # LDS-DMA pieces per wave and M-tile (four waves: 24 pieces of 8 rows x 128 B)
HALF_BYTES = 16 * 768  # rows 16 .. 31 of an image sit this far behind rows 0 .. 15
BUBBLE = 0             # epilogue instructions placed at the top of an M-tile, in front of its first MFMA (measured r03 with 22:
                       # 1 816 -> 1 912 cycles per M-tile, +2 % wall -- the first A operand is back sooner than they issue; off)
CAPS = (2, 1, 1, 0)    # filler instructions behind the MFMA of 16-query fragment n = 0 .. 3 of an operand (a 16-cycle MFMA
                       # leaves the wave 8 issue cycles; the next operand's ds_read + s_waitcnt stand behind fragment 3)

# pinned registers: accumulator sets [P][f] and the two buffer resources
ACC = {(0, 0): 0, (0, 1): 16, (1, 0): 32, (1, 1): 48}     # v[...]: first register of the 16-tuple
LD_RSRC = 40           # s[40:43]  LDS-DMA resource (base advances one M-tile per body)
ST_RSRC = 44           # s[44:47]  tile-word resource of this wave's query set


def acc_tuple(P, f):
    b = ACC[(P, f)]
    return f"v[{b}:{b + 15}]"


def acc16(P, r, n):
    """Accumulator of (set P, row half r, 16-query fragment n): four registers inside the tuple of fragment pair n >> 1."""
    b = ACC[(P, n >> 1)] + 8 * r + 4 * (n & 1)
    return f"v[{b}:{b + 3}]"


def acc_reg(P, f, i):
    return f"v{ACC[(P, f)] + i}"


class Instr:
    def __init__(self, text, reads=(), writes=(), kind="valu", cond_w=None, cond_r=None):
        self.text, self.reads, self.writes, self.kind = text, set(reads), set(writes), kind
        self.cond_w, self.cond_r = cond_w, cond_r      # SGPR pair written by a v_cmp / read as a condition


def epilogue_program(P, t):
    """The epilogue of fragment t of accumulator set P: list of Instr.  Temps x0..x9 of the fragment, three condition
    pairs c0..c2 and a junk carry pair cj (named operands)."""
    x = [f"%[x{t}_{i}]" for i in range(10)]
    c = [f"%[c{t}_0]"]
    gm = f"%[gm{t}]"
    R = [acc_reg(P, t, i) for i in range(16)]
    e0, e1, e2, e3, eu, ew, gu, gw, cu, cw = x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], x[8], x[9]
    mh, t1, em, w0, mine = x[0], x[1], x[2], x[3], x[0]
    p = []
    V = lambda text, r, w, **k: p.append(Instr(text, r, w, **k))
    # registers 4 k .. 4 k + 3 of the pair's tuple, k = 2 r + (n & 1): rows 16 r + 4 (l >> 4) .. + 3, query 16 n + (l & 15)
    for k, e in enumerate((e0, e1, e2, e3)):                                   # lane-local maxima of those four rows
        V(f"v_max_f32 {e}, {R[4 * k + 2]}, {R[4 * k + 3]}", [], [e])
        V(f"v_max3_f32 {e}, {R[4 * k]}, {R[4 * k + 1]}, {e}", [e], [e])
    # v_permlane16_swap(x, y) = {x.row0, y.row0, x.row2, y.row2}, {x.row1, y.row1, x.row3, y.row3} (rows of 16 lanes): the
    # maximum of the two is, for query 32 t + (l & 31), the 8-row M-tile 2 r in lanes < 32 and 2 r + 1 above
    V(f"v_permlane16_swap_b32 {e0}, {e1}", [e0, e1], [e0, e1], kind="swap")
    V(f"v_max_f32 {eu}, {e0}, {e1}", [e0, e1], [eu])                           # M-tiles 0 / 1 by lane half
    V(f"v_permlane16_swap_b32 {e2}, {e3}", [e2, e3], [e2, e3], kind="swap")
    V(f"v_max_f32 {ew}, {e2}, {e3}", [e2, e3], [ew])                           # M-tiles 2 / 3
    V(f"v_max_f32 {mh}, {eu}, {ew}", [eu, ew], [mh])
    V(f"v_mov_b32 {t1}, {mh}", [mh], [t1])
    V(f"v_permlane32_swap_b32 {mh}, {t1}", [mh, t1], [mh, t1], kind="swap")
    V(f"v_max_f32 {em}, {mh}, {t1}", [mh, t1], [em])                           # the tile maximum, in both halves
    V(f"v_max_f32 {gm}, {gm}, {em}", [gm, em], [gm])
    # bf16 of the tile maximum, rounded up: (b >> 31) ? (b >> 16) : ((b + 0xFFFF) >> 16)
    V(f"v_cmp_gt_i32_e64 {c[0]}, 0, {em}", [em], [], cond_w=c[0])
    V(f"v_add_u32 {w0}, 0xffff, {em}", [em], [w0])
    V(f"v_sub_f32 {gu}, {em}, {eu}", [em, eu], [gu])
    V(f"v_cndmask_b32_e64 {w0}, {w0}, {em}, {c[0]}", [w0, em], [w0], cond_r=c[0])
    V(f"v_lshrrev_b32 {w0}, 16, {w0}", [w0], [w0])
    # the two gap codes of this lane (rr_flt_gap_code, rr_x3.h): bits 20..23 of min(8 + gap / step, 31.99)
    V(f"v_fma_f32 {gu}, {gu}, %[inv], %[k8]", [gu], [gu])
    V(f"v_sub_f32 {gw}, {em}, {ew}", [em, ew], [gw])
    V(f"v_fma_f32 {gw}, {gw}, %[inv], %[k8]", [gw], [gw])
    V(f"v_min_f32 {gu}, 0x41ffeb85, {gu}", [gu], [gu])
    V(f"v_min_f32 {gw}, 0x41ffeb85, {gw}", [gw], [gw])
    V(f"v_bfe_u32 {cu}, {gu}, 20, 4", [gu], [cu])
    V(f"v_bfe_u32 {cw}, {gw}, 20, 4", [gw], [cw])
    V(f"v_lshl_or_b32 {mine}, {cw}, 8, {cu}", [cw, cu], [mine])
    V(f"v_lshlrev_b32 {mine}, %[cshift], {mine}", [mine], [mine])
    V(f"v_mov_b32 {t1}, {mine}", [mine], [t1])
    V(f"v_permlane32_swap_b32 {mine}, {t1}", [mine, t1], [mine, t1], kind="swap")
    V(f"v_or3_b32 {w0}, {mine}, {t1}, {w0}", [mine, t1, w0], [w0])
    off = " offset:128" if t else ""
    V(f"buffer_store_dword {w0}, %[stvoff], s[{ST_RSRC}:{ST_RSRC + 3}], %[stsoff] offen{off}", [w0], [], kind="vmem")
    return p


class Stream:
    """Emitted instructions of the whole loop, with the positions hazards are measured in."""

    def __init__(self):
        self.lines = []
        self.pos = 0
        self.valu_w = {}      # register -> position of its last VALU write
        self.cond_w = {}      # SGPR pair -> position of the v_cmp that wrote it
        self.cond_busy = {}   # SGPR pair -> True while a written condition has not been consumed

    def raw(self, text, n=1):
        self.lines.append(text)
        self.pos += n

    def ok(self, ins):
        if ins.kind == "swap":
            for r in ins.reads:
                if r in self.valu_w and self.pos - self.valu_w[r] < 3:
                    return False
        if ins.cond_r is not None and self.pos - self.cond_w.get(ins.cond_r, -99) < 3:
            return False
        if ins.cond_w is not None and self.cond_busy.get(ins.cond_w):
            return False
        return True

    def emit(self, ins):
        assert self.ok(ins), ins.text
        if ins.kind in ("valu", "swap"):
            for r in ins.writes:
                self.valu_w[r] = self.pos
        if ins.cond_w is not None:
            self.cond_w[ins.cond_w] = self.pos
            self.cond_busy[ins.cond_w] = True
        if ins.cond_r is not None:
            # the last reader of a condition frees it: a pair is read exactly once per write
            self.cond_busy[ins.cond_r] = False
        self.raw(ins.text)


def gen_body(S, k):
    """Body k of the unrolled loop: M-tile it = 4 n + 1 + k."""
    P = (1 + k) & 1
    buf = (1 + k) % NB
    dbuf = k % NB                                   # image of M-tile it + 3
    S.raw(f"; ---- body {k}: accumulator set {P}, image {buf}, LDS-DMA into image {dbuf}", 0)
    if not (ABL & 4):
        S.raw("s_waitcnt vmcnt(16)")
        S.raw("s_barrier")
    S.raw("s_add_u32 %[stsoff], %[stsoff], 512")     # -> the tile words of M-tile it - 1
    # A operand u = 2 ks + r: rows 16 r .. 16 r + 15 of the M-tile, dims 32 ks .. 32 ks + 31 (lane l: row l & 15, 16-byte
    # piece 4 ks + (l >> 4)); per-lane addresses by the parity of ks, everything else in the offset field
    abase = lambda u: (f"%[ah{(u >> 1) & 1}]" if buf >= 2 else f"%[al{(u >> 1) & 1}]")
    aoff = lambda u: (buf & 1) * TILE_BYTES + (u & 1) * HALF_BYTES + 1024 * (u >> 2)
    for i in range(AD):
        if not (ABL & 8):
            S.raw(f"ds_read_b128 %[A{i}], {abase(i)} offset:{aoff(i)}")
    progs = [epilogue_program(1 - P, 0), epilogue_program(1 - P, 1)]
    if ABL & 2:
        progs = [[], []]
    nxt = [0, 0]
    # (experiment, BUBBLE > 0: epilogue instructions between the first A reads and the first MFMA of the M-tile)
    for _ in range(0 if (ABL & 512) else BUBBLE):
        # fragment 0 first; fragment 1 (last written by the previous body's LAST MFMA) only behind >= 18 issued instructions
        t = 0 if (nxt[0] < len(progs[0]) and S.ok(progs[0][nxt[0]])) else 1
        if t == 1 and (nxt[0] < 12 or nxt[1] >= len(progs[1]) or not S.ok(progs[1][nxt[1]])):
            break
        S.emit(progs[t][nxt[t]])
        nxt[t] += 1
    start_gap = [2, 10]                             # first gap a fragment's epilogue may use
    dma_gaps = {4 * (4 * j + 3): j for j in range(PIECES)}          # behind the first MFMA of operands 3, 7, ... 23
    for idx in range(96):
        u, n = idx >> 2, idx & 3                    # operand (K-step u >> 1, row half u & 1), 16-query fragment
        ks, r = u >> 1, u & 1
        if n == 0 and not (ABL & 8):
            if u + AD < 24:
                S.raw(f"ds_read_b128 %[A{(u + AD) % (AD + 1)}], {abase(u + AD)} offset:{aoff(u + AD)}")
            S.raw(f"s_waitcnt lgkmcnt({min(AD, 23 - u)})")
        acc = acc16(P, r, n)
        S.raw(f"v_mfma_f32_16x16x32_bf16 {acc}, %[A{u % (AD + 1)}], %[b{ks}_{n}], {'0' if ks == 0 else acc}")
        if idx in dma_gaps and not (ABL & 1):
            j = dma_gaps[idx]
            # M0 = LDS address of the piece; its 128 j bytes into the rows go through the SCALAR offset (memory side only:
            # an instruction offset would move the LDS address as well)
            S.raw(f"s_add_u32 m0, %[ldsw], {dbuf * TILE_BYTES + j * 1024}")
            S.raw("s_nop 0")                                            # (one wait state between M0 and the piece)
            soff = "0" if j == 0 else f"%[o{128 * j}]"
            S.raw(f"buffer_load_dwordx4 %[dvoff], s[{LD_RSRC}:{LD_RSRC + 3}], {soff} offen lds")
            if j == PIECES - 1 and not (ABL & 64):                      # the resource moves on one M-tile
                S.raw(f"s_add_u32 s{LD_RSRC}, s{LD_RSRC}, {TILE_BYTES}")
                S.raw(f"s_addc_u32 s{LD_RSRC + 1}, s{LD_RSRC + 1}, 0")
                S.raw(f"s_sub_u32 s{LD_RSRC + 2}, s{LD_RSRC + 2}, {TILE_BYTES}")
                S.raw(f"s_cselect_b32 s{LD_RSRC + 2}, 0, s{LD_RSRC + 2}")
            continue
        cap = CAPS[n]
        m = 0
        while m < cap:
            cand = [t for t in (0, 1) if nxt[t] < len(progs[t]) and idx >= start_gap[t]]
            # the fragment that is further behind goes first
            cand.sort(key=lambda t: nxt[t])
            done = False
            for t in cand:
                ins = progs[t][nxt[t]]
                if S.ok(ins):
                    S.emit(ins)
                    nxt[t] += 1
                    m += 1
                    done = True
                    break
            if not done:
                break                              # nothing may issue here: leave the rest of the shadow empty
    # whatever is left (should be nothing) runs behind the last MFMA
    for t in (0, 1):
        while nxt[t] < len(progs[t]):
            ins = progs[t][nxt[t]]
            if not S.ok(ins):
                S.raw("s_nop 1", 2)
                continue
            S.emit(ins)
            nxt[t] += 1
            S.raw("; (epilogue instruction behind the last MFMA)", 0)


def emit_loop(abl=0):
    """The asm statement of the loop (a C++ block), as text; abl = ablation bits (0: the product's loop)."""
    global ABL, CAPS, AD
    ABL = abl
    # 1024 + v: schedule variants for same-box A/Bs (nothing ablated: correct results)
    CAPS, AD = {1025: ((1, 1, 1, 1), 4), 1026: ((2, 2, 0, 0), 4), 1027: ((2, 1, 1, 0), 6), 1028: ((2, 1, 1, 0), 3),
                1029: ((3, 1, 0, 0), 4), 1030: ((2, 2, 0, 0), 6)}.get(abl, ((2, 1, 1, 0), 4))
    if abl >= 1024:
        ABL = 0
    S = Stream()
    # m0 is rewritten for every LDS-DMA piece below.  The compiler ignores a clobber of that reserved register ("may not be
    # preserved"), and its own LDS-DMA code in the C++ bodies around this statement may keep a value in it: saved and restored
    S.raw("s_mov_b32 %[m0s], m0", 0)
    S.raw("L_fltq_loop_%=:", 0)
    for k in range(4):
        gen_body(S, k)
    S.raw("s_sub_u32 %[loops], %[loops], 1")
    S.raw("s_cmp_lg_u32 %[loops], 0")
    S.raw("s_cbranch_scc1 L_fltq_loop_%=")
    # the compiler copies the accumulators out of their pinned registers right behind this statement, and does not know
    # that an MFMA wrote them 32 cycles ago: let the matrix pipe drain here
    S.raw("s_nop 15", 16)
    S.raw("s_nop 7", 8)
    S.raw("s_mov_b32 m0, %[m0s]", 0)
    out = []
    out.append("{")
    out.append("    u32x4 " + ", ".join(f"tA{i}" for i in range(AD + 1)) + ";")
    out.append("    uint32_t " + ", ".join(f"tx{t}_{i}" for t in (0, 1) for i in range(10)) + ";")
    out.append("    uint64_t tc0_0, tc1_0;")
    out.append("    uint32_t tm0;")
    out.append("    asm volatile(")
    for ln in S.lines:
        out.append('        "' + ln + '\\n\\t"')
    outs = []
    for (P, f), b in ACC.items():
        outs.append(f'"+{{v[{b}:{b + 15}]}}"(acc[{P}][{f}])')
    outs += [f'[gm{t}] "+v"(gm[{t}])' for t in (0, 1)]
    outs += [f'"+{{s[{LD_RSRC}:{LD_RSRC + 3}]}}"(ld_rsrc)', '[stsoff] "+s"(st_soff)', '[loops] "+s"(loops)']
    outs += [f'[A{i}] "=&v"(tA{i})' for i in range(AD + 1)]
    outs += [f'[x{t}_{i}] "=&v"(tx{t}_{i})' for t in (0, 1) for i in range(10)]
    outs += [f'[c{t}_0] "=&s"(tc{t}_0)' for t in (0, 1)]
    outs += ['[m0s] "=&s"(tm0)']
    ins = [f'[b{ks}_{n}] "a"(bq[{ks}][{n}])' for ks in range(12) for n in range(4)]
    ins += [f'[al{m}] "v"(al[{m}])' for m in range(2)] + [f'[ah{m}] "v"(ah[{m}])' for m in range(2)]
    ins += ['[dvoff] "v"(dma_voff)', '[stvoff] "v"(st_voff)', '[cshift] "v"(code_shift)', '[inv] "s"(inv_step_s)',
            '[ldsw] "s"(lds_w)', f'"{{s[{ST_RSRC}:{ST_RSRC + 3}]}}"(st_rsrc)', '[k8] "v"(8.0f)']
    ins += [f'[o{128 * j}] "s"({128 * j}u)' for j in range(1, PIECES)]
    out.append("        : " + ",\n          ".join(outs))
    out.append("        : " + ",\n          ".join(ins))
    out.append('        : "vcc", "scc", "memory");')
    out.append("}")
    # schedule statistics on stderr
    n_nop = sum(1 for ln in S.lines if ln.startswith("s_nop"))
    late = sum(1 for ln in S.lines if "behind the last MFMA" in ln)
    sys.stderr.write(f"abl {abl}: {len(S.lines)} lines, {n_nop} s_nop, {late} epilogue instructions behind the last MFMA\n")
    return "\n".join(out) + "\n"


HEADER = """// GENERATED by gen_fltq_loop.py -- do not edit.  The steady-state loop of rr_scan_fltq: four M-tile bodies per
// iteration, every instruction placed by hand (see the generator's header).  Expects in scope: acc[2][2] (f32x16),
// gm[2], bq[12][4], al[2], ah[2] (A-read addresses of images 0/1 and 2/3), dma_voff, st_voff, code_shift (VGPR),
// inv_step_s, lds_w, st_soff, loops (SGPR), ld_rsrc, st_rsrc (u32x4 SGPR).
"""


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--abl":
        sys.stdout.write("// GENERATED by gen_fltq_loop.py --abl -- timing ablations of the loop (debug harness only; wrong results).\n")
        for i, a in enumerate(int(x) for x in sys.argv[2].split(",")):
            sys.stdout.write(("else " if i else "") + f"if constexpr (ABL == {a})\n" + emit_loop(a))
        return
    sys.stdout.write(HEADER + emit_loop(0))


if __name__ == "__main__":
    main()
