"""Lint of the shading kernels' ISA for the two hazards the compiler does not see inside asm statements (shade.hip):
  (a) a scalar load issued by an asm statement (the light pairs: s_load_dwordx4 between ;;#ASMSTART / ;;#ASMEND) whose destination
      SGPRs are read or overwritten before the next s_waitcnt lgkmcnt(0);
  (b) a transcendental (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos) whose result is read by the VALU instruction that
      immediately follows it (gfx950 needs one wait state there; the compiler inserts it for its own instructions only).
The scan is per BASIC BLOCK: a label resets both checks.  A scalar load still pending at the end of its block is not followed further -- across
a merge "pending" is only a maybe (the light loop's last trip issues no load and leaves through the same block as the trips that do), and the
hazards this lint exists for are the certain ones: the compiler touching an asm load's destination right behind it (round 4 met exactly that:
under an SGPR budget the register allocator SPILLED the freshly loaded light pair, v_writelane of registers whose data had not arrived).
usage: python tools/isa_lint.py <shade .s from `make -C arctic-renderer_amd/csrc asm`>   (exit code 1 when something is found)"""
import re, sys

def regs(text, kind):
    out = set()
    for a, b in re.findall(r"\b%s\[(\d+):(\d+)\]" % kind, text):
        out.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\b%s(\d+)\b" % kind, text):
        out.add(int(a))
    return out

def split_ops(line):
    parts = line.split(None, 1)
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return parts[0], ops

TRANS = re.compile(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_")

def lint(path):
    problems, kernel, in_asm = [], None, False
    pending = {}            # SGPR -> line number of the asm scalar load that writes it, until the next s_waitcnt lgkmcnt(0)
    prev_trans = None       # (dest VGPRs, line number) of the instruction just before, when it was a transcendental
    for n, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
        label = raw.split(";")[0].rstrip()        # (the assembler prints `name: ; @name`)
        if label.startswith("_Z") and label.endswith(":") and "k_material" in label:
            kernel, pending, prev_trans = label[:-1], {}, None
            continue
        if kernel is None:
            continue
        if line.startswith(".end_amdhsa_kernel") or raw.startswith(".Lfunc_end"):
            kernel = None
            continue
        if line.startswith(";;#ASMSTART"): in_asm = True; continue
        if line.startswith(";;#ASMEND"): in_asm = False; continue
        if not line or line.startswith((".", ";")) or line.endswith(":"):
            if line.endswith(":"):   # a label: another path may enter here, so nothing is CERTAIN to be pending or to be the instruction before any more
                prev_trans = None
                pending = {}
            continue
        op, ops = split_ops(line)
        if op == "s_waitcnt" and (re.search(r"lgkmcnt\(\s*0\s*\)", line) or (ops and re.fullmatch(r"0(x0+)?", ops[0]))):
            pending = {}      # any form of the instruction whose lgkmcnt field is 0: with other counters beside it, in any order, or the plain immediate 0
        elif op.startswith("s_load") and in_asm:
            for r in regs(ops[0], "s"): pending[r] = n
        elif pending:
            touched = regs(" ".join(ops), "s") & set(pending)
            if touched:
                problems.append(f"{path}:{n}: {kernel}: `{line}` touches s{sorted(touched)} loaded at line {pending[min(touched)]} before s_waitcnt lgkmcnt(0)")
        if op.startswith("v_") and prev_trans is not None:
            srcs = regs(" ".join(ops[1:]), "v")
            if srcs & prev_trans[0]:
                problems.append(f"{path}:{n}: {kernel}: `{line}` reads v{sorted(srcs & prev_trans[0])} right behind the transcendental at line {prev_trans[1]}")
        prev_trans = (regs(ops[0], "v"), n) if TRANS.match(op) else None
        if op == "s_nop": prev_trans = None
    return problems

if __name__ == "__main__":
    found = lint(sys.argv[1])
    for p in found: print(p)
    print(f"isa_lint: {len(found)} problem(s)")
    sys.exit(1 if found else 0)
