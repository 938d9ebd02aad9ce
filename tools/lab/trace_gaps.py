"""Lab: from a rocprofv3 kernel trace, how busy is the GPU inside the timed steps?  python tools/lab/trace_gaps.py <dir>
Prints, over thirteen steady-state steps near the end of the trace: span, union of kernel intervals (busy), sum of durations
(> busy when kernels of two streams overlap), and where the idle time sits (gap histogram)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]))
rows.sort()
per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 5400
rows = rows[-15 * per_step:-2 * per_step]          # thirteen steady-state steps out of the timed region
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in rows)
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy(union) {busy / 1e6:.2f} ms ({busy / span * 100:.1f}%)  sum of durations {tot / 1e6:.2f} ms")
edges = [0, 1000, 2000, 3000, 5000, 10000, 20000, 50000, 100000, 10 ** 12]
for a, b in zip(edges[:-1], edges[1:]):
    g = [x for x, _ in gaps if a <= x < b]
    print(f"  gaps {a / 1e3:6.0f}-{b / 1e3:<9.0f} us: {len(g):6d}  total {sum(g) / 1e6:8.2f} ms")
big = sorted(gaps, reverse=True)[:12]
print("largest gaps (us, next kernel):", [(round(x / 1e3), n) for x, n in big])
