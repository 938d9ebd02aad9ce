"""Print the top rows of a rocprofv3 kernel_stats.csv with shortened kernel names: python tools/lab/stats_top.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in list(csv.DictReader(open(f)))[:n]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(7), f"{int(r['TotalDurationNs'])/1e6:9.2f} ms", f"{float(r['AverageNs'])/1e3:8.2f} us", r["Percentage"])
