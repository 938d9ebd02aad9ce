"""Lab: where does the wall time of one replayed step go?  Reads a rocprofv3 kernel trace (the CSV `tools/prof.sh` leaves under
/tmp/prof_raw_<tag>/trace), takes the dispatches between two consecutive adam_kernel launches and prints: the step's span, the
time at least one kernel was running (union over the queues), per-queue busy time and launch counts, the idle time between
consecutive kernels of the busiest queue (sum, histogram) and the kernels that are followed by the longest gaps.
    python tools/lab/timeline.py /tmp/prof_raw_<tag>"""
import collections, csv, glob, os, sys
raw = sys.argv[1]
f = glob.glob(os.path.join(raw, "trace/**/*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
tr = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get(qkey, "0")) for r in rows)
marks = [i for i, t in enumerate(tr) if "adam_kernel" in t[2]]
step = min(zip(marks, marks[1:]), key=lambda ab: ab[1] - ab[0])
ks = tr[step[0] + 1: step[1] + 1]
t0, t1 = ks[0][0], max(k[1] for k in ks)
print(f"{len(ks)} dispatches, span {(t1 - t0) / 1e6:.3f} ms, kernel time {sum(k[1] - k[0] for k in ks) / 1e6:.3f} ms")
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in sorted(ks):
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"at least one kernel running: {busy / 1e6:.3f} ms; nothing running: {(t1 - t0 - busy) / 1e6:.3f} ms")
perq = collections.defaultdict(list)
for k in ks:
    perq[k[3]].append(k)
for q, v in sorted(perq.items(), key=lambda kv: -len(kv[1])):
    print(f"  queue {q}: {len(v)} dispatches, busy {sum(k[1] - k[0] for k in v) / 1e6:.3f} ms")


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:70]


# idle between consecutive kernels (all queues together: gaps where NOTHING runs)
gaps, cur_e, last = [], None, None
for s, e, n, q in sorted(ks):
    if cur_e is not None and s > cur_e:
        gaps.append((s - cur_e, last, short(n)))
    if cur_e is None or e > cur_e:
        cur_e, last = e, short(n)
hist = collections.Counter()
for g, _, _ in gaps:
    hist["<2us" if g < 2000 else "2-5us" if g < 5000 else "5-10us" if g < 10000 else "10-30us" if g < 30000 else ">30us"] += 1
print("idle gaps (nothing running):", dict(hist), f"total {sum(g for g, _, _ in gaps) / 1e6:.3f} ms in {len(gaps)} gaps")
by = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    by[(a, b)][0] += g
    by[(a, b)][1] += 1
print("largest idle totals by (kernel before -> kernel after):")
for (a, b), (g, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {g / 1e3:8.1f} us in {n:3d} gaps   {a}  ->  {b}")
