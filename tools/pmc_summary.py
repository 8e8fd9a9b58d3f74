"""Summarise the rocprofv3 passes of tools/profile_round.sh into profiles/<tag>_pmc_counters.json + profiles/pmc_latest.json.

usage: python tools/pmc_summary.py <dir with one sub-directory per rocprofv3 pass> <round tag>
Per counter: the rows of one dispatch of k_material are summed (rocprofv3 emits one row per counter instance), then averaged over
the dispatches.  HBM traffic per shading pass = 2 x FETCH_SIZE + WRITE_SIZE (KB): on gfx950 FETCH_SIZE counts a 128-byte request as
64 B (MI355X_MICROARCH.md, HBM/rocprofv3 section), writes as reported.  Clock = GRBM_GUI_ACTIVE per instance / kernel duration.
bench.py reads pmc_latest.json for the STATIC inputs of roofline.valu_issue_frac / hbm_frac / traffic."""
import csv, glob, json, os, re, sys
from collections import defaultdict

root, tag = sys.argv[1], sys.argv[2]
KERNEL = "k_material"
acc = defaultdict(lambda: defaultdict(float))      # counter -> dispatch -> summed value
rows = defaultdict(lambda: defaultdict(int))       # counter -> dispatch -> instances
dur = defaultdict(dict)                            # pass -> dispatch -> ns
for path in glob.glob(os.path.join(root, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if KERNEL in row["Kernel_Name"]:
            key = (path, row["Dispatch_Id"])
            acc[row["Counter_Name"]][key] += float(row["Counter_Value"])
            rows[row["Counter_Name"]][key] += 1
out = {c: sum(v.values()) / len(v) for c, v in sorted(acc.items())}
inst = {c: sum(v.values()) / len(v) for c, v in rows.items()}
stats = {}
for path in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if KERNEL in row["Name"]:
            which = os.path.relpath(path, root).split(os.sep)[0]
            full = re.search(r"(k_\w+(<[^>]*>)?)\(", row["Name"])
            stats[f"{which}:{full.group(1) if full else KERNEL}"] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"])}
res = {"round": tag, "counters": out, "instances_per_dispatch": inst, "kernel_stats": stats}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
json.dump(res, open(os.path.join(dst, f"{tag}_pmc_counters.json"), "w"), indent=1)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as entry   # (no GPU call: only the source hash / commit of the build that was profiled)
latest = {"round": tag, "build": entry.source_id(),
          "command": "tools/profile_round.sh: rocprofv3 --pmc <set> --kernel-trace -- python3 tools/prof_shade.py full (one set per run)",
          "config": "config 3: 3840x2160, 64 point lights, shadow 4000^2, ACES; one shading pass = one launch of k_material<2>",
          "algorithmic_bytes_per_launch": 3840 * 2160 * 80}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    latest.update({"FETCH_SIZE_KB": out["FETCH_SIZE"], "WRITE_SIZE_KB": out["WRITE_SIZE"],
                   "correction": "reads doubled (gfx950 FETCH_SIZE counts 128-B requests as 64 B); writes as reported",
                   "hbm_bytes_per_launch": int(round((2 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024))})
if "SQ_INSTS_VALU" in out:
    latest["SQ_INSTS_VALU"] = out["SQ_INSTS_VALU"]
    latest["trans_insts"] = out.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
    pm = next((v for k, v in stats.items() if k.startswith("pmc3")), None)
    N_XCD, N_SIMD = 8, 1024   # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (one GRBM each): 4.0 M "cycles" for a 0.235 ms launch
    if pm and "GRBM_GUI_ACTIVE" in out:
        latest["GRBM_GUI_ACTIVE_per_XCD"] = out["GRBM_GUI_ACTIVE"] / N_XCD
        latest["measured_clock_GHz"] = round(out["GRBM_GUI_ACTIVE"] / N_XCD / pm["avg_ns"], 3)
        latest["kernel_ns_in_that_pass"] = pm["avg_ns"]
    if "SQ_ACTIVE_INST_VALU" in out and "GRBM_GUI_ACTIVE" in out:
        latest["SQ_ACTIVE_INST_VALU"] = out["SQ_ACTIVE_INST_VALU"]
        # rocprof's VALUBusy: SQ_ACTIVE_INST_VALU counts in units of 4 cycles, summed over all SIMDs
        latest["valu_busy_frac"] = round(out["SQ_ACTIVE_INST_VALU"] * 4 / N_SIMD / (out["GRBM_GUI_ACTIVE"] / N_XCD), 4)
json.dump(latest, open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
