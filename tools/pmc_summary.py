"""Summarise rocprofv3 counter passes into profiles/<round>_pmc_counters.json + profiles/pmc_latest.json.

usage: python tools/pmc_summary.py <dir with one sub-directory per `rocprofv3 --pmc ... --kernel-trace --output-format csv` pass> <round tag>
Per kernel (k_material, k_light) and counter: sum over the rows of one dispatch (rocprofv3 may emit one row per instance),
averaged over the dispatches.  HBM traffic per shading pass = 2 x FETCH_SIZE + WRITE_SIZE (KB): on gfx950 FETCH_SIZE counts a
128-byte request as 64 B (MI355X_MICROARCH.md, HBM/rocprofv3 section; checked here against two known byte counts), writes as reported.
"""
import csv, glob, json, os, sys
from collections import defaultdict

root, tag = sys.argv[1], sys.argv[2]
KERNELS = ("k_material", "k_light", "k_shade_fused")
acc = defaultdict(lambda: defaultdict(float))      # (kernel, counter) -> dispatch -> value
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = next((k for k in KERNELS if k + "(" in row["Kernel_Name"] or k + "<" in row["Kernel_Name"]), None)
        if k:
            acc[(k, row["Counter_Name"])][(path, row["Dispatch_Id"])] += float(row["Counter_Value"])
out = {f"{k}.{c}": sum(v.values()) / len(v) for (k, c), v in sorted(acc.items())}
for path in glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = next((k for k in KERNELS if k + "(" in row["Name"] or k + "<" in row["Name"]), None)
        if k:
            import re
            full = re.search(r"(k_\w+(<\d+>)?)\(", row["Name"]).group(1)   # with its template argument: k_material<2>, k_light<4>
            out[f"{full}.avg_ns_under_kernel_trace[{os.path.basename(os.path.dirname(os.path.dirname(path)))}] ({row['Calls']} calls)"] = float(row["AverageNs"])
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_counters.json"), "w"), indent=1)
f = {k: out[f"{k}.FETCH_SIZE"] for k in KERNELS if f"{k}.FETCH_SIZE" in out}
w = {k: out[f"{k}.WRITE_SIZE"] for k in KERNELS if f"{k}.WRITE_SIZE" in out}
if any(f.values()):
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 tools/prof_shade.py full",
               "config": "config 3: 3840x2160, 64 point lights, shadow 4000^2, ACES; one shading pass = the kernels listed (k_material<2> alone by default)",
               "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
               "correction": "reads doubled (gfx950 FETCH_SIZE counts 128-B requests as 64 B); writes as reported",
               "hbm_bytes_per_launch": int(round((2 * sum(f.values()) + sum(w.values())) * 1024)),
               "algorithmic_bytes_per_launch": 3840 * 2160 * 80, "round": tag},
              open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
