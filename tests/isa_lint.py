"""isa_lint.py -- a linear scan of a kernel's disassembly for the hazard hand-placed waits can have: an instruction that reads (or
overwrites) a vector register while a load into it is still in flight.

step_hot_kernel (hydrochrono_amd/csrc/hc_kernels.hip) issues its global loads as asm statements and waits for them by hand
(s_waitcnt vmcnt(N)); the compiler believes such a register is written where the request is made.  Anything it places between the
request and the wait that touches the register -- a copy for a phi, a spill to an accumulation register -- reads what the register
held BEFORE.  (One revision's release build did: EXPERIMENTS.md, round 6.)  The scan follows the instructions in address order, counts
vmcnt the way the hardware does on gfx9 (loads and stores, retired in order), and reports every such touch.  It is exact for code
without a branch between a request and its wait -- which is what step_hot_kernel promises; for compiler-managed loads in branchy
code it over-reports (two exclusive paths look like one), so it is applied to that kernel only (tests/test_isa_lint.py).
"""
import re, subprocess, sys
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

def regs(tok):
    """VGPR indices named by one operand token: v12, v[2:5]; accumulation registers are not tracked."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()

def kernel_lines(co, symbol):
    txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--no-leading-addr", co], capture_output=True, text=True, check=True).stdout
    out, on = [], False
    for ln in txt.splitlines():
        if re.match(r"^\S*<.*%s.*>:" % re.escape(symbol), ln) or (symbol in ln and ln.rstrip().endswith(">:")):
            on = True
            continue
        if on:
            if ln.rstrip().endswith(">:") and symbol not in ln:
                break
            ins = ln.split("//")[0].strip()
            if ins:
                out.append(ins)
    return out

def lint(lines):
    pending = []  # oldest first: (mnemonic, dest registers)
    bad = []
    for n, ins in enumerate(lines):
        parts = ins.replace(",", " ").split()
        mn, ops = parts[0], parts[1:]
        if mn == "s_endpgm":
            pending = []
            continue
        if mn == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if m:
                keep = int(m.group(1))
                pending = pending[len(pending) - keep:] if keep < len(pending) else pending
            continue
        inflight = set().union(*[d for _, d in pending]) if pending else set()
        is_load = mn.startswith("global_load") or mn.startswith("flat_load") or mn.startswith("buffer_load")
        is_store = mn.startswith("global_store") or mn.startswith("flat_store") or mn.startswith("buffer_store") or mn.startswith("global_atomic")
        if is_load:
            dest = regs(ops[0]) if ops else set()
            srcs = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
            if srcs & inflight:
                bad.append((n, ins, "address from a register whose load is in flight", sorted(srcs & inflight)))
            if dest & inflight:
                bad.append((n, ins, "second load into a register whose load is in flight", sorted(dest & inflight)))
            pending.append((mn, dest))
            continue
        used = set().union(*[regs(o) for o in ops]) if ops else set()
        if used & inflight:
            bad.append((n, ins, "touches a register whose load is in flight", sorted(used & inflight)))
        if is_store:
            pending.append((mn, set()))
    return bad

if __name__ == "__main__":
    co = sys.argv[1]
    for sym in sys.argv[2:]:
        L = kernel_lines(co, sym)
        b = lint(L)
        print(sym, len(L), "instructions,", len(b), "findings")
        for x in b[:12]:
            print("   ", x)
