"""per-kernel averages from a rocprofv3 --kernel-trace --stats output directory.
usage: python tools/kernel_times.py <dir>                       every kernel: calls, average, minimum (the profiler's own stats)
       python tools/kernel_times.py <dir> --timed <bench.json>  the shading pass cut to the TIMED launches of that bench.py run: the per-launch
                                                                trace rows of k_material, in start order, [roofline.timed_launches) of the line
                                                                bench.py printed -- the average that has to agree with its ms_per_step"""
import csv, glob, json, re, sys


def short(name):
    return name.replace("arctic::(anonymous namespace)::", "").split("(")[0]


def timed_cut(directory, bench_json):
    line = json.loads([l for l in open(bench_json).read().splitlines() if l.strip().startswith("{")][-1])
    first, last = line["roofline"]["timed_launches"]
    rows = []
    for p in glob.glob(directory + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            n = short(r["Kernel_Name"])
            if re.match(r"(void )?k_material(_many_lights|_few_lights|<)", n):   # the pass over a G-buffer (not k_material_vis)
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
    rows.sort()
    cut = rows[first:last]
    if len(cut) != last - first:
        raise SystemExit(f"kernel trace holds {len(rows)} launches of k_material; the line asks for [{first}, {last})")
    d = [(e - s) / 1e3 for s, e, _ in cut]
    span = (cut[-1][1] - cut[0][0]) / 1e3 / len(cut)
    return {"kernel": cut[0][2], "launches_in_trace": len(rows), "timed_launches": [first, last], "avg_us": round(sum(d) / len(d), 2), "min_us": round(min(d), 2),
            "max_us": round(max(d), 2), "span_per_launch_us": round(span, 2), "bench_ms_per_step": line["ms_per_step"], "bench_kernel_ms": line["roofline"]["kernel_ms"],
            "all_launches_avg_us": round(sum((e - s) / 1e3 for s, e, _ in rows) / len(rows), 2)}


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[2] == "--timed":
        print(json.dumps(timed_cut(sys.argv[1], sys.argv[3]), indent=1))
        sys.exit(0)
    for p in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            print("%-40s calls %4s avg %9.1f us  min %8.1f" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
