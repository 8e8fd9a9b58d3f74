"""Lint of the shading kernels' ISA for the two hazards the compiler does not see inside asm statements (shade.hip):
  (a) a scalar load issued by an asm statement (the light pairs, the split material-descriptor load: s_load_* between ;;#ASMSTART / ;;#ASMEND)
      whose destination SGPRs are read or overwritten before an s_waitcnt lgkmcnt(0) -- on ANY path that leaves the load: the walk follows
      fall-through edges and branch targets (s_branch, s_cbranch_*) from the load until every path has met the wait, the end of the program,
      or an instruction it has visited before.  It starts AT a load, so what it reports is certain for that path: "nothing touches the
      destination between issue and wait" is checked, not assumed, however many basic blocks lie between the two (the descriptor load of the
      fast tile is waited for some 90 instructions and 7 blocks later);
  (b) a transcendental (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos) whose result is read by the VALU instruction that
      immediately follows it (gfx950 needs one wait state there; the compiler inserts it for its own instructions only).  Per straight line:
      a label or an s_nop in between clears it.
Round 4 met (a) for real: under an SGPR budget the register allocator SPILLED the freshly loaded light pair (v_writelane of registers whose
data had not arrived) -- and this lint had matched no kernel at the time.  So lint() reports what it scanned as well, and the test asserts it:
kernels, asm scalar loads, loads whose walk crossed a label, instructions visited.
usage: python tools/isa_lint.py <shade .s from `make -C arctic-renderer_amd/csrc asm`>   (exit code 1 when something is found or nothing was scanned)"""
import re, sys
from collections import namedtuple

Report = namedtuple("Report", "problems kernels asm_loads loads_across_labels visited")

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
Ins = namedtuple("Ins", "n op ops text in_asm")

def is_full_lgkm_wait(ins):
    return ins.op == "s_waitcnt" and (re.search(r"lgkmcnt\(\s*0\s*\)", ins.text) is not None or (ins.ops and re.fullmatch(r"0(x0+)?", ins.ops[0]) is not None))

def parse_kernels(path, match="k_material"):
    """[(kernel symbol, [Ins ...], {label: index of the first instruction behind it})] for every function whose symbol contains `match`"""
    kernels, cur, in_asm = [], None, False
    for n, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip() if not raw.strip().startswith(";;#") else raw.strip()
        label = raw.split(";")[0].rstrip()        # (the assembler prints `name: ; @name`)
        if label.startswith("_Z") and label.endswith(":") and match in label:
            cur = (label[:-1], [], {})
            kernels.append(cur)
            in_asm = False
            continue
        if cur is None:
            continue
        if line.startswith(".end_amdhsa_kernel") or raw.startswith(".Lfunc_end"):
            cur = None
            continue
        if line.startswith(";;#ASMSTART"): in_asm = True; continue
        if line.startswith(";;#ASMEND"): in_asm = False; continue
        if not line or line.startswith(";"):
            continue
        if line.endswith(":"):
            cur[2][line[:-1]] = len(cur[1])
            continue
        if line.startswith("."):
            continue
        op, ops = split_ops(line)
        cur[1].append(Ins(n, op, ops, line, in_asm))
    return kernels

def successors(ins_list, labels, i):
    ins = ins_list[i]
    if ins.op in ("s_endpgm", "s_setpc_b64"):
        return []
    if ins.op == "s_branch":
        return [labels[ins.ops[0]]] if ins.ops and ins.ops[0] in labels else []
    out = [i + 1] if i + 1 < len(ins_list) else []
    if ins.op.startswith("s_cbranch") and ins.ops and ins.ops[-1] in labels:
        out.append(labels[ins.ops[-1]])
    return out

def lint(path, match="k_material"):
    problems, n_loads, n_across, n_visited = [], 0, 0, 0
    kernels = parse_kernels(path, match)
    for name, ins_list, labels in kernels:
        label_at = set(labels.values())
        # (a) every asm scalar load: walk every path that leaves it until lgkmcnt(0)
        for i, ins in enumerate(ins_list):
            if not (ins.in_asm and ins.op.startswith("s_load")):
                continue
            n_loads += 1
            dest = regs(ins.ops[0], "s")
            seen, stack, crossed = set(), list(successors(ins_list, labels, i)), False
            while stack:
                j = stack.pop()
                if j in seen or j >= len(ins_list):
                    continue
                seen.add(j)
                if j in label_at: crossed = True
                nxt = ins_list[j]
                if is_full_lgkm_wait(nxt):
                    continue
                touched = regs(" ".join(nxt.ops), "s") & dest
                if touched:
                    problems.append(f"{path}:{nxt.n}: {name}: `{nxt.text}` touches s{sorted(touched)} loaded at line {ins.n} before s_waitcnt lgkmcnt(0)")
                    continue
                stack.extend(successors(ins_list, labels, j))
            n_visited += len(seen)
            n_across += 1 if crossed else 0
        # (b) transcendental -> the very next VALU instruction
        prev_trans = None
        for i, ins in enumerate(ins_list):
            if i in label_at: prev_trans = None
            if ins.op.startswith("v_") and prev_trans is not None:
                srcs = regs(" ".join(ins.ops[1:]), "v")
                if srcs & prev_trans[0]:
                    problems.append(f"{path}:{ins.n}: {name}: `{ins.text}` reads v{sorted(srcs & prev_trans[0])} right behind the transcendental at line {prev_trans[1]}")
            prev_trans = (regs(ins.ops[0], "v"), ins.n) if TRANS.match(ins.op) else None
    return Report(problems, len(kernels), n_loads, n_across, n_visited)

if __name__ == "__main__":
    rep = lint(sys.argv[1])
    for p in rep.problems: print(p)
    print(f"isa_lint: {len(rep.problems)} problem(s) in {rep.kernels} kernel(s), {rep.asm_loads} asm scalar load(s) followed to their wait "
          f"({rep.loads_across_labels} across labels, {rep.visited} instructions visited)")
    sys.exit(1 if rep.problems or rep.kernels == 0 or rep.asm_loads == 0 else 0)
