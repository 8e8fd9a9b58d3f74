"""per-kernel averages of the counters collected by raster_counters.sh.  usage: python tools/experiments/raster_counters_summary.py gpurun_out/prof_raster_pmc"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Kernel_Name"].replace("arctic::(anonymous namespace)::", "").split("(")[0]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n in sorted(acc):
    if not any(k in n for k in ("k_raster", "k_setup", "k_vertex", "k_material_vis")): continue
    print(n)
    for c, v in sorted(acc[n].items()):
        print(f"    {c:28s} {sum(v)/len(v)/1e6:12.3f} M per launch ({len(v)} launches)")
