"""print a rocprofv3 kernel_stats.csv compactly: kernel, calls, average / min / max in microseconds"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(anonymous namespace)::')[-1].split('(')[0]
    if n.startswith('SetupTables'): n = 'k_setup' + ('_clipped' if 'const*)' in r['Name'][-40:] else '')
    print(f"{n:40s} calls {r['Calls']:>5s}  avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}")
