#!/usr/bin/env python3
"""Static check of one device-function call in a gfx950 code object (no GPU): which registers are LIVE ACROSS the call in the caller
(backward dataflow over the caller's control-flow graph, built from llvm-objdump's disassembly) and also WRITTEN by the callee.
AMDGPU compiles local functions with interprocedural register allocation: the callee saves nothing, the caller must keep every
live value in registers the callee never writes.  A non-empty intersection is a miscompiled call.

    python3 scripts/call_liveness.py CODE_OBJECT_OR_FAT_OBJECT CALLER_SUBSTRING CALLEE_SUBSTRING

Approximations (all on the safe side for SGPRs, which are written whole): an instruction's first operand is its destination unless
the opcode has none (stores, branches, compares to SCC, waits); v_writelane / DPP / SDWA-preserve destinations are read as well as
written; a VALU write kills its register although lanes switched off by EXEC keep their old value -- so a VGPR that is live only in
lanes that were inactive at a later write is missed (reported separately: VGPRs written by the callee while EXEC may be partial)."""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_meta as km

NODEST = ("scratch_store", "flat_store", "global_store", "buffer_store", "ds_write", "s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_barrier",
          "s_endpgm", "s_setpc", "s_cmp", "s_bitcmp", "s_sleep", "s_setprio", "v_cmpx", "s_set_gpr_idx", "s_sendmsg", "s_trap", "s_code_end",
          "buffer_wbl2", "buffer_inv", "s_dcache", "s_icache", "s_setreg")


def regs(tok):
    out = set()
    tok = tok.strip()
    m = re.fullmatch(r"-?\|?([vsa])(\d+)\|?", tok)
    if m:
        out.add((m.group(1), int(m.group(2))))
    m = re.fullmatch(r"([vsa])\[(\d+):(\d+)\]", tok)
    if m:
        out |= {(m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1)}
    if tok == "vcc":
        out |= {("vcc", 0)}
    if tok == "exec":
        out |= {("exec", 0)}
    return out


def parse(line):
    m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*[0-9A-Fa-f ]+(?:<(.*)>)?\s*$", line)
    if not m:
        return None
    op, rest, addr, tgt = m.group(1), m.group(2), int(m.group(3), 16), m.group(4)
    ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", rest)] if rest else []
    mods = " ".join(ops)
    ops = [o.split()[0] if o else o for o in ops]
    d, s = set(), set()
    no_dest = op.startswith(NODEST) or ("atomic" in op and " sc0" not in mods and " glc" not in mods)
    if no_dest:
        for o in ops:
            s |= regs(o)
    else:
        if ops:
            d |= regs(ops[0])
        two = op.startswith(("v_mad_u64", "v_mad_i64", "v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_subrev_co", "v_subbrev_co", "v_div_scale")) and len(ops) > 1
        if two:
            d |= regs(ops[1])                                  # carry-out / scale flag: a second destination
        for o in ops[(2 if two else 1):]:
            s |= regs(o)
        if op.startswith("v_writelane") or "dpp" in op or "UNUSED_PRESERVE" in mods or "row_" in mods or "quad_perm" in mods or op.startswith(("v_mac", "v_fmac", "v_pk_fmac", "v_dot")):
            s |= {r for r in d if r[0] == "v"}                 # partial / accumulating write of a VGPR
    if "saveexec" in op or op.startswith("v_cmpx"):
        d |= {("exec", 0)}; s |= {("exec", 0)}
    if op.startswith("v_") and not op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        s |= {("exec", 0)}
    return dict(op=op, addr=addr, d=d, s=s, tgt=tgt, text=f"{op} {rest}")


def function(lines, name):
    starts = [(i, l) for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <", l)]
    for k, (i, l) in enumerate(starts):
        if name in l:
            j = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
            base = int(l.split()[0], 16)
            ins = [p for p in (parse(x) for x in lines[i + 1:j]) if p]
            return base, ins
    raise SystemExit(f"no function matching {name}")


def main():
    obj, caller_n, callee_n = sys.argv[1:4]
    cos = km.code_objects(obj) or [obj]
    txt = subprocess.run([os.path.join(km.LLVM, "llvm-objdump"), "-d", cos[0]], capture_output=True, text=True).stdout.split("\n")
    cbase, caller = function(txt, caller_n)
    _, callee = function(txt, callee_n)
    clob = set()
    for p in callee:
        clob |= p["d"]
    clob |= {("s", 30), ("s", 31)}
    print(f"callee {callee_n}: {len(callee)} instructions; writes v{sorted(k for t, k in clob if t == 'v')}\n  s{sorted(k for t, k in clob if t == 's')}")
    idx = {p["addr"]: i for i, p in enumerate(caller)}
    succ = []
    for i, p in enumerate(caller):
        nx = []
        op = p["op"]
        if op.startswith(("s_cbranch", "s_branch")) and p["tgt"]:
            m = re.search(r"\+0x([0-9a-fA-F]+)$", p["tgt"])
            t = cbase + (int(m.group(1), 16) if m else 0)
            if t in idx:
                nx.append(idx[t])
        if not op.startswith(("s_branch", "s_endpgm", "s_setpc")) and i + 1 < len(caller):
            nx.append(i + 1)
        succ.append(nx)
    live_in = [set() for _ in caller]
    changed = True
    while changed:
        changed = False
        for i in range(len(caller) - 1, -1, -1):
            p = caller[i]
            out = set()
            for j in succ[i]:
                out |= live_in[j]
            d, s = p["d"], p["s"]
            if p["op"].startswith("s_swappc"):
                d = set()                                       # the question is what is live across it: do not let it kill
            new = (out - d) | s
            if new != live_in[i]:
                live_in[i] = new; changed = True
    bad_total = 0
    for i, p in enumerate(caller):
        if p["op"].startswith("s_swappc"):
            out = set()
            for j in succ[i]:
                out |= live_in[j]
            bad = sorted(r for r in out & clob if r[0] in "vs")
            bad_total += len(bad)
            print(f"call at +0x{p['addr'] - cbase:x}: {len(out)} registers live across it; live AND written by the callee: "
                  f"{[t + str(k) for t, k in bad] or 'none'}")
            for r in bad[:12]:                                   # where the caller reads it next (first reader in address order reached)
                seen, todo = set(), list(succ[i])
                while todo:
                    j = todo.pop()
                    if j in seen:
                        continue
                    seen.add(j)
                    q = caller[j]
                    if r in q["s"]:
                        print(f"    {r[0]}{r[1]} read at +0x{q['addr'] - cbase:x}: {q['text']}")
                        break
                    if r in q["d"]:
                        continue
                    todo += succ[j]
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
