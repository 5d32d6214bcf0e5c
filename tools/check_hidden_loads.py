"""Build-time check of the hand-placed waits behind inline-asm loads.

k_hjoin.hip's partition kernel issues its batch loads from inline assembly (ld_hidden_* in hark_internal.h) so that the
compiler's s_waitcnt bookkeeping does not see them, and waits for them by hand (wait_vm<N>: `s_waitcnt vmcnt(N)`).  The
contract the compiler cannot check: between such a load and the wait that covers it NO instruction may read or write the
load's destination VGPRs (a v_mov the register allocator slipped in, a spill to scratch, a use hoisted above the wait).
This script checks it on the code the compiler actually produced:

    hipcc -S --cuda-device-only <unit>  ->  per kernel: a control-flow graph of the instruction text, a forward dataflow whose
    state is the set of hidden loads still in flight {destination registers: vector-memory operations issued since}, every
    instruction tested against it.  A load is complete after `s_waitcnt vmcnt(N)` when at least N vector-memory operations
    (loads, stores, atomics -- gfx950 counts them in ONE in-order counter) were issued after it; paths merge pessimistically
    (the fewest operations since, the union of the loads).

Usage: python tools/check_hidden_loads.py <unit.hip> [more units]   (run by harkdb_amd/csrc/Makefile; exit 1 on a violation)
"""
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "--cuda-device-only", "-S"]
VM_PREFIX = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "flat_load", "flat_store",
             "flat_atomic", "scratch_load", "scratch_store")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse_functions(asm):
    """{kernel name: [(mnemonic, operand text, in_inline_asm, source line number)]} with labels as ('LABEL', name, ...)."""
    funcs, cur, name, in_asm = {}, None, None, False
    for ln, raw in enumerate(asm.split("\n"), 1):
        line = raw.strip()
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.startswith(";") or line.startswith("//"):
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", raw)
        if m:
            lab = m.group(1)
            if not lab.startswith(".L") and not lab.startswith("."):
                name, cur = lab, []
                funcs[name] = cur
            elif cur is not None:
                cur.append(("LABEL", lab, False, ln))
            continue
        if line.startswith(".") or cur is None:
            if line.startswith(".Lfunc_end") or line.startswith(".size"):
                pass
            continue
        code = line.split(";")[0].strip()
        if not code:
            continue
        parts = code.split(None, 1)
        cur.append((parts[0], parts[1] if len(parts) > 1 else "", in_asm, ln))
    return funcs


def blocks_of(instrs):
    """Basic blocks [(start, end)] and successor lists."""
    leaders = {0}
    label_at = {}
    for i, (mn, ops, _, _) in enumerate(instrs):
        if mn == "LABEL":
            label_at[ops] = i
            leaders.add(i)
        elif mn.startswith("s_cbranch") or mn in ("s_branch", "s_endpgm", "s_setpc_b64"):
            leaders.add(i + 1)
    starts = sorted(x for x in leaders if x < len(instrs))
    blocks = [(s, starts[j + 1] if j + 1 < len(starts) else len(instrs)) for j, s in enumerate(starts)]
    index_of = {s: j for j, (s, _) in enumerate(blocks)}
    succ = []
    for (s, e) in blocks:
        last = instrs[e - 1]
        out = []
        if last[0].startswith("s_cbranch"):
            out.append(index_of[label_at[last[1].strip()]])
            if e < len(instrs):
                out.append(index_of[e])
        elif last[0] == "s_branch":
            out.append(index_of[label_at[last[1].strip()]])
        elif last[0] in ("s_endpgm", "s_setpc_b64"):
            pass
        elif e < len(instrs):
            out.append(index_of[e])
        succ.append(out)
    return blocks, succ


def vmcnt_of(ops):
    m = re.search(r"vmcnt\((\d+)\)", ops)
    if m:
        return int(m.group(1))
    m = re.match(r"^\s*(0x[0-9a-fA-F]+|\d+)\s*$", ops)         # a raw immediate: decode the gfx9 vmcnt field
    if m:
        imm = int(m.group(1), 0)
        return (imm & 0xF) | ((imm >> 14) & 0x3) << 4
    return None


SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")


def sregs_of(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_kernel(name, instrs):
    """Violations {(source line, instruction): registers of an in-flight hidden load it touches}.

    The walk is PATH-SENSITIVE in one respect: the AMDGPU structurizer funnels every `break` of a loop and its back-edge through
    one latch block, telling them apart by a flag (`s_mov_b64 s[a:b], -1` on the break paths, `s_and_b64 vcc, exec, s[a:b]`,
    `s_cbranch_vccz <header>` in the latch).  A path-insensitive merge would send the state of the break paths -- a prefetch
    issued, nothing waited for -- around the back-edge, where no execution goes.  So a state carries the known constants
    (-1 / 0) of SGPR pairs and of vcc, states with different constants are not merged, and a conditional branch on a known
    vcc follows one side only."""
    blocks, succ_idx = blocks_of(instrs)
    label_block = {}
    for j, (s0, _) in enumerate(blocks):
        if instrs[s0][0] == "LABEL":
            label_block[instrs[s0][1]] = j
    seen = [set() for _ in blocks]
    start = (frozenset(), frozenset())                          # (hidden loads in flight {(dest regs, ops since)}, constants {(name, value)})
    work = [(0, start)]
    seen[0].add(start)
    violations, hidden_sites = {}, set()
    steps = 0
    while work:
        b, (pend_f, const_f) = work.pop()
        steps += 1
        if steps > 2_000_000:
            raise SystemExit(f"{name}: state space too large for the path-sensitive walk")
        pend, const = dict(pend_f), dict(const_f)
        s, e = blocks[b]
        nxt = None                                              # successor blocks; None = by the block's last instruction, both sides
        for i in range(s, e):
            mn, ops, in_asm, ln = instrs[i]
            if mn == "LABEL":
                continue
            if mn == "s_waitcnt":
                n = vmcnt_of(ops)
                if n is not None:
                    pend = {d: c for d, c in pend.items() if c < n}   # complete: at least n operations were issued after the load
                continue
            opl = [o.strip() for o in ops.split(",")]
            # ---- constants of scalar pairs / vcc
            if mn == "s_mov_b64" and len(opl) == 2 and opl[1] in ("-1", "0"):
                const[opl[0]] = -1 if opl[1] == "-1" else 0
            elif mn == "s_and_b64" and len(opl) == 3 and opl[0] == "vcc" and opl[1] == "exec" and const.get(opl[2]) is not None:
                const["vcc"] = const[opl[2]]                       # (exec is not empty where a wave executes)
            elif not mn.startswith(("s_cbranch", "s_branch")):
                written = set(opl[:2])                             # destination, and the carry-out / mask of the *_co_* and VOPC forms
                wr = sregs_of(" ".join(opl[:2]))
                for k_ in list(const):
                    if k_ in written or (k_ != "vcc" and sregs_of(k_) & wr) or (k_ == "vcc" and "vcc" in opl[:2]):
                        del const[k_]
                if mn.startswith("v_cmp") and mn.endswith("_e32"):
                    const.pop("vcc", None)
            # ---- the contract: nobody touches the destination of a load that is still in flight
            touched = regs_of(ops)
            for d in pend:
                if touched & d:
                    violations[(ln, mn + " " + ops)] = sorted(touched & d)
            if mn.startswith(VM_PREFIX):
                pend = {d: min(c + 1, 16) for d, c in pend.items()}
                if in_asm and "load" in mn:
                    pend[frozenset(regs_of(opl[0]))] = 0
                    hidden_sites.add(ln)
            if i == e - 1 and mn in ("s_cbranch_vccz", "s_cbranch_vccnz") and const.get("vcc") is not None:
                taken = (const["vcc"] == 0) == (mn == "s_cbranch_vccz")
                nxt = [label_block[ops.strip()]] if taken else ([b + 1] if e < len(instrs) else [])
        state = (frozenset(pend.items()), frozenset(const.items()))
        for nb in (succ_idx[b] if nxt is None else nxt):
            if state not in seen[nb]:
                seen[nb].add(state)
                work.append((nb, state))
    return len(hidden_sites), violations


def main():
    bad = 0
    for unit in sys.argv[1:]:
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "unit.s")
            subprocess.check_call([HIPCC] + FLAGS + ["-o", out, unit], stderr=subprocess.DEVNULL)
            funcs = parse_functions(open(out).read())
        total_hidden = 0
        for name, instrs in funcs.items():
            if not any(a and "load" in mn for mn, _, a, _ in instrs):
                continue
            hidden, viol = check_kernel(name, instrs)
            total_hidden += hidden
            for (ln, text), regs in sorted(viol.items()):
                bad += 1
                print(f"{unit}: {name}: asm line {ln}: `{text}` touches v{regs} while an inline-asm load into them is in flight", file=sys.stderr)
        print(f"{os.path.basename(unit)}: {total_hidden} inline-asm load sites followed through the control-flow graph, {bad} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
