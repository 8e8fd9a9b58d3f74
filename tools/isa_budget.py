"""Per-basic-block instruction census of one kernel from hipcc's -save-temps ISA (make -C arctic-renderer_amd/csrc asm).
usage: python tools/isa_budget.py /tmp/shade-hip-amdgcn-amd-amdhsa-gfx950.s k_materialILi2E [--dump]
       python tools/isa_budget.py <.s> <kernel> --census-json   (instruction classes of the light loop and of the rest, for bench.py)
Classes: valu (v_* except the next three), pk (v_pk_*), trans (v_rcp/rsq/sqrt/exp/log/sin/cos: 8-cycle issue),
sel (v_cndmask), salu (s_*), vmem (global_/buffer_/flat_), lds (ds_*).  Cycles = issue cost from
tools/experiments/valu_rates.hip (4 per VALU, 8 per transcendental)."""
import re, sys, collections

def kernel_text(path, key):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if start is None and re.match(r"^_Z\w*%s\w*:" % re.escape(key), l):
            start = i
        elif start is not None and (l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end")):
            return lines[start:i]
    raise SystemExit("kernel not found")

def classify(op):
    if op.startswith("v_pk_"): return "pk"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", op): return "trans"
    if op.startswith("v_cndmask"): return "sel"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    return "other"

def max_vgpr_per_block(path, key, kind="v"):
    """highest VGPR (kind "s": SGPR) index touched per basic block: where the kernel's register count comes from"""
    txt = kernel_text(path, key)
    cur, out = "entry", []
    mx = -1
    for l in txt[1:]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            out.append((cur, mx)); cur = m.group(1); mx = -1; continue
        s = l.strip()
        if not s or s.startswith((";", ".")): continue
        for a, b in re.findall(kind + r"\[(\d+):(\d+)\]", s): mx = max(mx, int(b))
        for a in re.findall(r"\b" + kind + r"(\d+)\b", s): mx = max(mx, int(a))
    out.append((cur, mx))
    return out

FAST = re.compile(r"v_(add|sub|subrev|mul|fma|fmac|fmaak|fmamk)_f32|v_(mov|and|or|xor|lshrrev|lshlrev|add_u|sub_u|subrev_u)")   # ~2.6 cycles per wave64 (profiles/r2_valu_rates.txt)
def census(path, key):
    """{class: count} per basic block + the light loop's bodies (blocks with >= 40 v_pk_* and 6 transcendentals) + the mix of the rest:
    what bench.py prices the counters with (valu_issue_frac, valu_flop_frac)."""
    txt = kernel_text(path, key)
    blocks, cur = [], ["entry", collections.Counter()]
    for l in txt[1:]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur); cur = [m.group(1), collections.Counter()]; continue
        st = l.strip()
        if not st or st.startswith((";", ".", "//")): continue
        op = st.split()[0]
        c = classify(op)
        if c in ("valu", "sel"):
            c = "fast" if FAST.match(op) else "slow"
            if re.match(r"v_(fma|fmac|fmaak|fmamk)_f32", op): cur[1]["flops2"] += 1
            elif re.match(r"v_(add|sub|subrev|mul)_f32", op): cur[1]["flops1"] += 1
        if c == "pk":
            c = "pk_fma" if op.startswith("v_pk_fma") else "pk_other"
        cur[1][c] += 1
    blocks.append(cur)
    loops = [c for _, c in blocks if c["pk_fma"] + c["pk_other"] >= 40 and c["trans"] == 6]
    rest = collections.Counter()
    for _, c in blocks:
        if not (c["pk_fma"] + c["pk_other"] >= 40 and c["trans"] == 6): rest.update(c)
    n = max(len(loops), 1)
    body = {k: sum(c[k] for c in loops) / n for k in ("pk_fma", "pk_other", "trans", "salu")}
    nv = rest["fast"] + rest["slow"] + rest["pk_fma"] + rest["pk_other"]
    return {"kernel": key, "light_loop_bodies": len(loops), "per_pair_trip": body,
            "rest_static": {k: rest[k] for k in ("fast", "slow", "pk_fma", "pk_other", "trans", "flops1", "flops2")},
            "rest_mix": {"fast": rest["fast"] / nv, "slow": rest["slow"] / nv, "pk": (rest["pk_fma"] + rest["pk_other"]) / nv,
                         "flops_per_inst_lane": (rest["flops1"] + 2 * rest["flops2"] + 4 * rest["pk_fma"] + 2 * rest["pk_other"]) / nv}}

def main():
    path, key = sys.argv[1], sys.argv[2]
    if "--census-json" in sys.argv:
        import json
        print(json.dumps(census(path, key), indent=1))
        return
    if "--vgpr" in sys.argv or "--sgpr" in sys.argv:
        kind = "v" if "--vgpr" in sys.argv else "s"
        for name, mx in max_vgpr_per_block(path, key, kind):
            if mx >= 0: print(f"{name:12s} max {kind}{mx}")
        return
    dump = "--dump" in sys.argv
    txt = kernel_text(path, key)
    blocks, cur = [], ["entry", collections.Counter(), []]
    for l in txt[1:]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur); cur = [m.group(1), collections.Counter(), []]; continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")): continue
        op = s.split()[0]
        c = classify(op)
        cur[1][c] += 1
        cur[2].append(s)
    blocks.append(cur)
    tot = collections.Counter()
    print(f"{'block':12s} {'valu':>5s} {'pk':>5s} {'trans':>5s} {'sel':>5s} {'salu':>5s} {'vmem':>5s} {'lds':>5s}  vector-issue cycles")
    for name, c, ins in blocks:
        if not sum(c.values()): continue
        cyc = 4 * (c['valu'] + c['pk'] + c['sel']) + 8 * c['trans']
        print(f"{name:12s} {c['valu']:5d} {c['pk']:5d} {c['trans']:5d} {c['sel']:5d} {c['salu']:5d} {c['vmem']:5d} {c['lds']:5d}  {cyc}")
        tot.update(c)
        if dump:
            for s in ins: print("      " + s)
    print(f"{'total':12s} {tot['valu']:5d} {tot['pk']:5d} {tot['trans']:5d} {tot['sel']:5d} {tot['salu']:5d} {tot['vmem']:5d} {tot['lds']:5d}")

if __name__ == "__main__":
    main()

