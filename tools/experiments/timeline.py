"""a rocprofv3 kernel_trace.csv as a timeline: per kernel its queue, start and end relative to the first of a window of frames.
usage: python tools/experiments/timeline.py kernel_trace.csv [first dispatch] [count]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[first:first + count]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    n = r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0]
    if n.startswith("SetupTables"): n = "k_setup*"
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"q{r['Queue_Id']} {n:34s} {s:9.1f} -> {e:9.1f} us  ({e - s:6.1f})")
