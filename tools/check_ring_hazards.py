"""Static check of the register-ring contract of the scan kernels, over the compiler's own assembly (hipcc -S).

The batched scans load matrix rows straight into VGPRs with inline-asm `global_load_dwordx4` and wait with hand-counted
`s_waitcnt vmcnt(N)`: between a load's issue and the wait that covers it NOTHING may read or write its destination
registers -- the compiler knows nothing about the data being in flight (to it the asm statement "wrote" them at once), so
an ablation that leaves such a register unread lets it hand the register to something else while the load is on its way
(round 2: rr_scan_flt<4, true, 1> faulted that way).  This walks a kernel's instructions in program order, keeps the
vector-memory queue the way the hardware counts it (loads, stores, atomics and LDS-DMA retire in order; vmcnt(N) = all but
the N youngest are done) and reports every instruction that touches a destination register of a load still in flight
(a later LOAD into the same register is fine -- returns are in order --, its address operands are not).  Every path of the
control-flow graph is walked (conditional branches fork, loops are walked around more than once).

    python tools/check_ring_hazards.py <file.s> <kernel-symbol-substring> [...]     exit status 1 on a violation
"""
import re
import sys

VM_PREFIX = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "flat_load",
             "flat_store", "flat_atomic", "scratch_load", "scratch_store")
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def kernel_body(asm, symbol):
    """Instruction lines (label or instruction, with their line numbers) of the first kernel whose symbol holds `symbol`."""
    lines = asm.splitlines()
    start = next((i for i, ln in enumerate(lines) if re.match(r"^\S*%s\S*:" % re.escape(symbol), ln) and not ln.startswith(".")), None)
    if start is None:
        raise ValueError(f"no kernel symbol containing {symbol!r}")
    body = []
    for i in range(start + 1, len(lines)):
        ln = lines[i].split(";")[0].rstrip()
        if not ln.strip():
            continue
        s = ln.strip()
        if s.startswith(".") and not s.endswith(":"):
            if s.startswith(".Lfunc_end"):
                break
            continue
        body.append((i + 1, s))
    return lines[start].split(":")[0], body


def check_kernel(asm, symbol, max_report=8, visits_per_pc=6):
    """Walks every path of the kernel's control-flow graph (conditional branches fork; a (pc, queue state) pair is walked
    once, a pc at most `visits_per_pc` times with different queue states: loops are walked around at least twice)."""
    name, body = kernel_body(asm, symbol)
    labels = {s[:-1]: k for k, (_, s) in enumerate(body) if s.endswith(":")}
    violations = {}
    loads_seen = set()
    visits = {}
    seen_states = set()
    # state: tuple of (age, regs) for the loads in flight, age = vector-memory operations issued since (oldest first)
    work = [(0, ())]
    while work:
        k, state = work.pop()
        inflight = list(state)
        while k < len(body):
            sig = (k, tuple((a, tuple(sorted(r))) for a, r, _, _ in inflight))
            if sig in seen_states or visits.get(k, 0) >= visits_per_pc:
                break
            seen_states.add(sig)
            visits[k] = visits.get(k, 0) + 1
            line, s = body[k]
            if s.endswith(":"):
                k += 1
                continue
            mnem = s.split()[0]
            ops = s[len(mnem):]
            if mnem == "s_endpgm":
                break
            if mnem == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", ops)
                if m:
                    n = int(m.group(1))
                    inflight = [e for e in inflight if e[0] < n]      # all but the n youngest operations are done
                k += 1
                continue
            is_vm = mnem.startswith(VM_PREFIX)
            is_load = is_vm and "load" in mnem and " lds" not in (" " + ops + " ") and "_lds_" not in mnem
            first, _, rest = ops.partition(",")
            # a load may re-use the destination of an older load (returns are in order): only its ADDRESS operands count
            touched = regs_of(rest) if is_load else regs_of(ops)
            for age, regs, l0, t0 in inflight:
                hit = touched & regs
                if hit:
                    violations.setdefault((line, l0), (line, s, l0, t0, sorted(hit)[:4]))
            if is_vm:
                inflight = [(a + 1, r, l0, t0) for a, r, l0, t0 in inflight]
                if is_load:
                    inflight.append((0, frozenset(regs_of(first)), line, s))
                    loads_seen.add(line)
            m = re.match(r"s_cbranch_(\w+)\s+(\S+)", s)
            if m and m.group(1) == "execz":
                m = None          # the skip around a lane-masked region: the region is issued (the kernels' counted waits rely
                                  # on a constant number of vector-memory operations per tile: some lane is always active)
            if m and m.group(2) in labels:
                work.append((labels[m.group(2)], tuple(inflight)))          # taken; fall through below
            m = re.match(r"s_branch\s+(\S+)", s)
            if m and m.group(1) in labels:
                k = labels[m.group(1)]
                continue
            k += 1
    uniq = sorted(violations.values())
    return name, len(loads_seen), uniq[:max_report], len(uniq)


def main():
    asm = open(sys.argv[1]).read()
    bad = 0
    for sym in sys.argv[2:]:
        name, n_loads, shown, total = check_kernel(asm, sym)
        print(f"{name}: {n_loads} register loads walked, {total} violation(s)")
        for line, s, l0, t0, regs in shown:
            print(f"  line {line}: `{s}` touches {regs} while the load of line {l0} is in flight: `{t0}`")
        bad += total
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
