#!/usr/bin/env python3
"""Generates csrc/rr_fltq_loop.inc: the hand-scheduled steady-state loop of rr_scan_fltq (gfx950), as ONE asm statement.

Why a generator: the C++ form of the loop (one asm statement per instruction, lambdas, sched_barrier) let hipcc put
`s_nop`s between dependent asm statements, 64-bit address arithmetic and v_readfirstlane chains in front of every
LDS-DMA piece and branches around the epilogue stores -- 2 350 cycles per M-tile for 1 536 cycles of MFMA (r02 stamps).
Here every instruction of four M-tile bodies (ring of four LDS images x two accumulator sets) is placed by hand:

  per M-tile and wave: 48 v_mfma_f32_32x32x16_bf16, 24 ds_read_b128 (A operands, four K-steps ahead, counted lgkmcnt),
  6 LDS-DMA pieces `buffer_load_dwordx4 ... lds` (scalar base, scalar piece offset and ONE per-lane offset register: no
  vector address arithmetic; the resource's num_records makes reads past the matrix return zeros), the epilogue of the PREVIOUS M-tile
  (46 vector instructions per 32-query fragment, the arithmetic of `piece()` in rr_scan_fltq, bit for bit) spread over
  the MFMA shadows at <= 4 (behind an even MFMA) / <= 2 (behind an odd one) instructions, 2 tile-word stores
  (buffer_store_dword: the upper lane half carries an out-of-range offset and is dropped by the bounds check).

Hazards kept by construction (wait states = instructions in between):
  VALU write -> v_permlane32_swap of that register: 2;  v_cmp (VALU) write of an SGPR pair -> VALU read of it: 2;
  s_add m0 -> LDS-DMA: 1;  MFMA write of an accumulator -> VALU read: the previous body's last MFMAs are > 60 cycles back.

Usage: python gen_fltq_loop.py > rr_fltq_loop.inc   (build.py checks that the committed .inc is up to date)
"""
import sys

NB = 4                 # ring of LDS images (M-tiles); the loop is unrolled over it
AD = 4                 # A operands requested this many K-steps ahead
TILE_BYTES = 32 * 768  # one 32-row M-tile image
PIECES = 6             # LDS-DMA pieces per wave and M-tile (four waves: 24 pieces of 8 rows x 128 B)
E_CAP, O_CAP = 4, 2    # filler instructions behind an even / odd MFMA

# pinned registers: accumulator sets [P][f] and the two buffer resources
ACC = {(0, 0): 0, (0, 1): 16, (1, 0): 32, (1, 1): 48}     # v[...]: first register of the 16-tuple
LD_RSRC = 40           # s[40:43]  LDS-DMA resource (base advances one M-tile per body)
ST_RSRC = 44           # s[44:47]  tile-word resource of this wave's query set


def acc_tuple(P, f):
    b = ACC[(P, f)]
    return f"v[{b}:{b + 15}]"


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
    c = [f"%[c{t}_{i}]" for i in range(3)]
    cj = f"%[cj{t}]"
    gm = f"%[gm{t}]"
    R = [acc_reg(P, t, i) for i in range(16)]
    e0, e1, e2, e3, eu, ew, gu, gw, cu, cw = x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], x[8], x[9]
    mh, t1, em, w0, mine = x[0], x[1], x[2], x[3], x[0]
    p = []
    V = lambda text, r, w, **k: p.append(Instr(text, r, w, **k))
    for k, e in enumerate((e0, e1, e2, e3)):                                   # lane-local maxima of the 8-row M-tiles
        V(f"v_max_f32 {e}, {R[4 * k + 2]}, {R[4 * k + 3]}", [], [e])
        V(f"v_max3_f32 {e}, {R[4 * k]}, {R[4 * k + 1]}, {e}", [e], [e])
    V(f"v_permlane32_swap_b32 {e0}, {e1}", [e0, e1], [e0, e1], kind="swap")
    V(f"v_max_f32 {eu}, {e0}, {e1}", [e0, e1], [eu])                           # M-tiles 0 / 1 by lane half
    V(f"v_permlane32_swap_b32 {e2}, {e3}", [e2, e3], [e2, e3], kind="swap")
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
    V(f"v_mul_f32 {gu}, %[inv], {gu}", [gu], [gu])
    V(f"v_sub_f32 {gw}, {em}, {ew}", [em, ew], [gw])
    V(f"v_mul_f32 {gw}, %[inv], {gw}", [gw], [gw])
    V(f"v_cvt_u32_f32 {cu}, {gu}", [gu], [cu])
    V(f"v_min_u32 {cu}, 12, {cu}", [cu], [cu])
    V(f"v_cvt_u32_f32 {cw}, {gw}", [gw], [cw])
    V(f"v_min_u32 {cw}, 12, {cw}", [cw], [cw])
    for g, cc in ((gu, cu), (gw, cw)):                                         # coarse steps: + (g >= 16) + (g >= 24) + (g >= 40)
        for i, k in enumerate(("%[k16]", "%[k24]", "%[k40]")):
            V(f"v_cmp_le_f32_e64 {c[i]}, {k}, {g}", [g], [], cond_w=c[i])
        for i in range(3):
            V(f"v_addc_co_u32_e64 {cc}, {cj}, 0, {cc}, {c[i]}", [cc], [cc], cond_r=c[i])
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
    S.raw("s_waitcnt vmcnt(16)")
    S.raw("s_barrier")
    S.raw("s_add_u32 %[stsoff], %[stsoff], 512")     # -> the tile words of M-tile it - 1
    abase = lambda ks: (f"%[ah{ks & 3}]" if buf >= 2 else f"%[al{ks & 3}]")
    aoff = lambda ks: (buf & 1) * TILE_BYTES + 1024 * (ks >> 2)
    for i in range(AD):
        S.raw(f"ds_read_b128 %[A{i}], {abase(i)} offset:{aoff(i)}")
    progs = [epilogue_program(1 - P, 0), epilogue_program(1 - P, 1)]
    nxt = [0, 0]
    start_gap = [1, 5]                              # first gap a fragment's epilogue may use
    dma_gaps = {2 * (4 * j + 3): j for j in range(PIECES)}          # even gaps E(3), E(7), ... E(23)
    for idx in range(48):
        ks, f = idx >> 1, idx & 1
        if f == 0:
            if ks + AD < 24:
                S.raw(f"ds_read_b128 %[A{(ks + AD) % (AD + 1)}], {abase(ks + AD)} offset:{aoff(ks + AD)}")
            S.raw(f"s_waitcnt lgkmcnt({min(AD, 23 - ks)})")
        c_in = "0" if ks == 0 else acc_tuple(P, f)
        S.raw(f"v_mfma_f32_32x32x16_bf16 {acc_tuple(P, f)}, %[A{ks % (AD + 1)}], %[b{ks}_{f}], {c_in}")
        if idx in dma_gaps:
            j = dma_gaps[idx]
            # M0 = LDS address of the piece; its 128 j bytes into the rows go through the SCALAR offset (memory side only:
            # an instruction offset would move the LDS address as well)
            S.raw(f"s_add_u32 m0, %[ldsw], {dbuf * TILE_BYTES + j * 1024}")
            S.raw("s_nop 0")                                            # (one wait state between M0 and the piece)
            soff = "0" if j == 0 else f"%[o{128 * j}]"
            S.raw(f"buffer_load_dwordx4 %[dvoff], s[{LD_RSRC}:{LD_RSRC + 3}], {soff} offen lds")
            if j == PIECES - 1:                                         # the resource moves on one M-tile
                S.raw(f"s_add_u32 s{LD_RSRC}, s{LD_RSRC}, {TILE_BYTES}")
                S.raw(f"s_addc_u32 s{LD_RSRC + 1}, s{LD_RSRC + 1}, 0")
                S.raw(f"s_sub_u32 s{LD_RSRC + 2}, s{LD_RSRC + 2}, {TILE_BYTES}")
                S.raw(f"s_cselect_b32 s{LD_RSRC + 2}, 0, s{LD_RSRC + 2}")
            continue
        cap = E_CAP if f == 0 else O_CAP
        n = 0
        while n < cap:
            cand = [t for t in (0, 1) if nxt[t] < len(progs[t]) and idx >= start_gap[t]]
            # the fragment that is further behind goes first
            cand.sort(key=lambda t: nxt[t])
            done = False
            for t in cand:
                ins = progs[t][nxt[t]]
                if S.ok(ins):
                    S.emit(ins)
                    nxt[t] += 1
                    n += 1
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


def main():
    S = Stream()
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
    out = []
    out.append("// GENERATED by gen_fltq_loop.py -- do not edit.  The steady-state loop of rr_scan_fltq: four M-tile bodies per")
    out.append("// iteration, every instruction placed by hand (see the generator's header).  Expects in scope: acc[2][2] (f32x16),")
    out.append("// gm[2], bq[24][2], al[4], ah[4] (A-read addresses of images 0/1 and 2/3), dma_voff, st_voff, code_shift (VGPR),")
    out.append("// inv_step_s, lds_w, st_soff, loops (SGPR), ld_rsrc, st_rsrc (u32x4 SGPR).")
    out.append("{")
    out.append("    u32x4 tA0, tA1, tA2, tA3, tA4;")
    out.append("    uint32_t " + ", ".join(f"tx{t}_{i}" for t in (0, 1) for i in range(10)) + ";")
    out.append("    uint64_t " + ", ".join([f"tc{t}_{i}" for t in (0, 1) for i in range(3)] + ["tcj0", "tcj1"]) + ";")
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
    outs += [f'[c{t}_{i}] "=&s"(tc{t}_{i})' for t in (0, 1) for i in range(3)]
    outs += [f'[cj{t}] "=&s"(tcj{t})' for t in (0, 1)]
    ins = [f'[b{ks}_{f}] "a"(bq[{ks}][{f}])' for ks in range(24) for f in (0, 1)]
    ins += [f'[al{m}] "v"(al[{m}])' for m in range(4)] + [f'[ah{m}] "v"(ah[{m}])' for m in range(4)]
    ins += ['[dvoff] "v"(dma_voff)', '[stvoff] "v"(st_voff)', '[cshift] "v"(code_shift)', '[inv] "s"(inv_step_s)',
            '[ldsw] "s"(lds_w)', f'"{{s[{ST_RSRC}:{ST_RSRC + 3}]}}"(st_rsrc)',
            '[k16] "s"(0x41800000u)', '[k24] "s"(0x41c00000u)', '[k40] "s"(0x42200000u)']
    ins += [f'[o{128 * j}] "s"({128 * j}u)' for j in range(1, PIECES)]
    out.append("        : " + ",\n          ".join(outs))
    out.append("        : " + ",\n          ".join(ins))
    out.append('        : "vcc", "scc", "m0", "memory");')
    out.append("}")
    sys.stdout.write("\n".join(out) + "\n")
    # schedule statistics on stderr
    n_nop = sum(1 for ln in S.lines if ln.startswith("s_nop"))
    late = sum(1 for ln in S.lines if "behind the last MFMA" in ln)
    sys.stderr.write(f"{len(S.lines)} lines, {n_nop} s_nop, {late} epilogue instructions behind the last MFMA\n")


if __name__ == "__main__":
    main()
