"""per-kernel averages from a rocprofv3 --kernel-trace --stats output directory.  usage: python tools/kernel_times.py <dir>"""
import csv, glob, sys
for p in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        n = r["Name"].replace("arctic::(anonymous namespace)::", "").split("(")[0]
        print("%-40s calls %4s avg %9.1f us  min %8.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
